"""`nn.Module` front ends that keep the reference's construction and call signatures
(`CDAN()`, `CBAM(gate_channels, ...)`, `forward(x)`) and its checkpoint
keys, while the arithmetic runs in libmdie_hip.so.

The modules hold parameters only.  Parameter containers are generated from the
layout table in `arch.py`, so `state_dict()` / `load_state_dict()` interoperate with
checkpoints written by the reference (`torch.save(network.state_dict())`,
/root/reference/models/base.py:52-55) in both directions.
"""
import math
import os

import torch
import torch.nn as nn

from . import arch
from . import engine as E
from . import lib as L


class _Node(nn.Module):
    """Bare container: a named position in the checkpoint key tree."""


def _attach(root, spec):
    fan_in = 1
    for name, (shape, kind) in spec.items():
        *path, leaf = name.split(".")
        node = root
        for part in path:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        if kind == "counter":
            node.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
        elif kind == "buffer":
            node.register_buffer(leaf, torch.ones(shape) if leaf == "running_var" else torch.zeros(shape))
        else:
            is_norm = (name.rsplit(".", 1)[0] + ".running_mean") in spec
            t = torch.empty(shape)
            if is_norm:
                t.fill_(1.0 if leaf == "weight" else 0.0)
            else:
                # torch's default for Conv2d / ConvTranspose2d / Linear: U(-1/sqrt(fan_in), +1/sqrt(fan_in))
                # with fan_in = size(1) * receptive field, for weight and bias alike
                if leaf == "weight":
                    fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
                bound = 1.0 / math.sqrt(fan_in)
                t.uniform_(-bound, bound)
            node.register_parameter(leaf, nn.Parameter(t))


def _fingerprint(module):
    """Changes whenever a parameter or buffer may have changed: tensor versions (eager updates bump them) plus the
    module's own counter, which everything that rewrites parameters WITHOUT bumping versions advances -- a replayed
    hipGraph with the optimizer step inside (train.CapturedStep) -- and which every train() -> eval() transition
    advances too, so an evaluation never runs on weights packed before the last training phase.

    The tensors are found once: walking the module tree through parameters() / buffers() cost 0.5 ms per forward -- half the
    GPU time of a batch-32 step, which made `net(x)` 7 % slower than the engine it wraps.  The cached list is re-validated by
    identity against the dicts the modules hold (a re-registered parameter or buffer rebuilds it)."""
    c = module.__dict__.get("_mdie_fp_cache")
    if c is not None:
        for d, n, t in c:
            if d.get(n) is not t:
                c = None
                break
    if c is None:
        c = []
        for m in module.modules():
            c.extend((m._parameters, n, t) for n, t in m._parameters.items() if t is not None)
            c.extend((m._buffers, n, t) for n, t in m._buffers.items() if t is not None)
        module.__dict__["_mdie_fp_cache"] = c
        module.__dict__["_mdie_fp_gen"] = module.__dict__.get("_mdie_fp_gen", 0) + 1      # a different SET of tensors is a change too
    v = getattr(module, "_mdie_epoch", 0) + (module.__dict__.get("_mdie_fp_gen", 0) << 40)
    for _, _, t in c:
        v += t._version
    return v


MAX_ENGINES = 4     # engines (workspace + side streams each) kept per module: least recently used beyond that are released


class CDAN(nn.Module):
    """Drop-in for `models.cdan.CDAN` (/root/reference/models/cdan.py:164-176).

    `precision` ("fp32" default, "bf16") can also be set through the environment
    variable MDIE_PRECISION so that the reference's config files stay unchanged.
    """

    def __init__(self, precision=None):
        super().__init__()
        _attach(self, arch.cdan_param_spec())
        self.precision = precision or os.environ.get("MDIE_PRECISION", "fp32")
        self.dropout_p = 0.2      # nn.Dropout(0.2), models/cdan.py:68 (training mode only)
        self._engines = {}
        self._packed = {}
        self._mdie_epoch = 0

    def train(self, mode=True):
        if self.training and not mode:
            self._mdie_epoch += 1      # weights may have been rewritten by graph replays (no version bump): repack at the next eval forward
        return super().train(mode)

    def _engine(self, device):
        # one engine (workspace + side streams) per (device, precision, STREAM): forwards issued on different streams may
        # overlap on the GPU and must not share a workspace; forwards on one stream are ordered by the stream
        key = (str(device), self.precision, torch.cuda.current_stream(device).cuda_stream)
        eng = self._engines.get(key)
        fp = (_fingerprint(self), id(next(self.parameters())))
        if eng is None:
            # transient streams must not pin a workspace each forever: the least recently used engine goes.  Its last forward may
            # still be running: wait for THAT engine's stream (the key holds its handle), not for the device -- a device-wide
            # synchronize per forward is what five round-robin streams would pay -- and never under capture, where any synchronize
            # is illegal (there nothing is evicted: the cache grows by one engine and is trimmed by the next eager forward).
            while len(self._engines) >= MAX_ENGINES and not torch.cuda.is_current_stream_capturing():
                old = next(iter(self._engines))
                try:
                    torch.cuda.ExternalStream(old[2], device=device).synchronize()
                except Exception:                               # (a stream its owner has destroyed since)
                    torch.cuda.synchronize(device)
                del self._engines[old], self._packed[old]
            eng = self._engines[key] = E.CdanEngine(device, self.precision)
            self._packed[key] = None
        else:
            self._engines[key] = self._engines.pop(key)         # most recently used last
        if self._packed[key] != fp:
            eng.load(self.state_dict())
            self._packed[key] = fp
        return eng

    def forward(self, x):
        if not x.is_cuda:
            raise L.MdieError(f"CDAN.forward: input is on {x.device}; this engine runs on the GPU only (no CPU fallback)")
        if self.training:
            from . import train as T   # batch-statistic BatchNorm, dropout, autograd through the HIP convolutions
            return T.forward_train(self, x, self.precision, self.dropout_p)
        return self._engine(x.device).forward(x)

    def forward_with_taps(self, x):
        y, extras = self._engine(x.device).forward(x, want_taps=True)
        return y, extras["taps"]


class CBAM(nn.Module):
    """Drop-in for `models.cbam.CBAM` (/root/reference/models/cbam.py:84-95), eval mode."""

    def __init__(self, gate_channels, reduction_ratio=16, pool_types=("avg", "max"), no_spatial=False, precision=None):
        super().__init__()
        if reduction_ratio != arch.REDUCTION or tuple(pool_types) != ("avg", "max"):
            raise NotImplementedError("the HIP CBAM implements reduction_ratio=16 with pool_types ['avg','max'] "
                                      "(the only configuration the CDAN path instantiates, cdan.py:104-112,168)")
        spec = arch.cbam_param_spec(gate_channels)
        if no_spatial:
            spec = type(spec)((k, v) for k, v in spec.items() if k.startswith("ChannelGate"))
        _attach(self, spec)
        self.gate_channels, self.no_spatial = gate_channels, no_spatial
        self.precision = precision or os.environ.get("MDIE_PRECISION", "fp32")

    def forward(self, x):
        if not x.is_cuda:
            raise L.MdieError(f"CBAM.forward: input is on {x.device}; GPU only (no CPU fallback)")
        dt = E.dtype_id(self.precision)
        if self.training:
            if self.no_spatial:
                raise NotImplementedError("CBAM(no_spatial=True) has no training path (the network never builds one)")
            from . import train as T
            self.SpatialGate.spatial.bn.num_batches_tracked += 1
            y = T.cbam(dt, self, x.to(E.TORCH_DTYPE[dt]).contiguous(memory_format=torch.channels_last))
            return y.float().contiguous()
        from . import ops  # noqa: F401  (registers torch.ops.mdie.*)
        return torch.ops.mdie.cbam_forward(x, *self._eval_constants(x.device), dt, self.no_spatial)

    def _eval_constants(self, dev):
        """(w1, b1, w2, b2, w7, folded BatchNorm(1)) as fp32 device tensors, rebuilt only when a parameter or buffer has changed
        (`_fingerprint`): a forward used to walk the state_dict, copy eight tensors and fold the BatchNorm on every call."""
        fp = (_fingerprint(self), str(dev))
        c = self.__dict__.get("_mdie_eval_consts")
        if c is not None and c[0] == fp:
            return c[1]
        sd = self.state_dict()
        f = lambda k: sd[k].detach().to(dev, torch.float32).contiguous()
        if self.no_spatial:
            w7 = torch.zeros(98, device=dev)
            bn = torch.tensor([1.0, 0.0], device=dev)
        else:
            w7 = f("SpatialGate.spatial.conv.weight").reshape(-1)
            p = "SpatialGate.spatial.bn."
            s = f(p + "weight") / torch.sqrt(f(p + "running_var") + 1e-5)
            bn = torch.cat((s, f(p + "bias") - f(p + "running_mean") * s))
        consts = (f("ChannelGate.mlp.1.weight"), f("ChannelGate.mlp.1.bias"), f("ChannelGate.mlp.3.weight"), f("ChannelGate.mlp.3.bias"), w7, bn)
        self.__dict__["_mdie_eval_consts"] = (fp, consts)
        return consts
