"""Device-side pieces around the network: input feed, post-processing,
uint8 output, PSNR/SSIM.  Thin ctypes wrappers; the arithmetic is in csrc/post.hip."""
import ctypes as C

import torch

from . import lib as L
from .engine import _require_gpu, _stream_ptr

DEFAULT_ARGS = {"enhance_contrast": ("contrast_factor", 1.1), "enhance_color": ("saturation_factor", 1.1),
                "sharpen": ("strength", 0.5), "soft_denoise": ("sigma", 0.2)}


def feed_uint8(images_u8_hwc):
    """uint8 [B,H,W,3] (GPU) -> float32 NCHW in [0,1]."""
    _require_gpu(images_u8_hwc, "feed_uint8")
    x = images_u8_hwc.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
    L.check(L.lib.mdie_u8hwc_to_f32nchw(B, H, W, x.data_ptr(), out.data_ptr(), _stream_ptr(x.device)), "mdie_u8hwc_to_f32nchw")
    return out


def to_uint8_hwc(images):
    """float32 NCHW -> uint8 [B,H,W,3] exactly like (img*255).clip(0,255).astype(uint8)."""
    _require_gpu(images, "to_uint8_hwc")
    x = images.contiguous()
    B, _, H, W = x.shape
    out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=x.device)
    L.check(L.lib.mdie_f32nchw_to_u8hwc(B, H, W, x.data_ptr(), out.data_ptr(), _stream_ptr(x.device)), "mdie_f32nchw_to_u8hwc")
    return out


def _ops_array(pp_cfg):
    ops = []
    for op in (pp_cfg or {}).get("ops", []):
        name = op["name"]
        if name not in L.PP_KINDS:
            raise ValueError(f"Unknown post-processing op: {name}")
        key, default = DEFAULT_ARGS[name]
        ops.append(L.PpOp(L.PP_KINDS[name], float((op.get("args") or {}).get(key, default))))
    return (L.PpOp * max(len(ops), 1))(*ops), len(ops)


def apply_postprocessing(images, pp_cfg, want_uint8=False):
    """Same contract as utils.postprocessing_factory.apply_postprocessing (reference), on the GPU.
    pp_cfg: {"enabled": bool, "ops": [{"name": ..., "args": {...}}, ...]}"""
    if not pp_cfg or not pp_cfg.get("enabled", False):
        return (images, to_uint8_hwc(images)) if want_uint8 else images
    _require_gpu(images, "apply_postprocessing")
    x = images.contiguous()
    B, _, H, W = x.shape
    arr, n = _ops_array(pp_cfg)
    nws = L.lib.mdie_postprocess_workspace_bytes(B, H, W)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    out = torch.empty_like(x)
    u8 = torch.empty(B, H, W, 3, dtype=torch.uint8, device=x.device) if want_uint8 else None
    L.check(L.lib.mdie_postprocess(B, H, W, x.data_ptr(), arr, n, out.data_ptr(), u8.data_ptr() if u8 is not None else None,
                                   ws.data_ptr(), nws, _stream_ptr(x.device)), "mdie_postprocess")
    return (out, u8) if want_uint8 else out


def psnr_ssim(pred, target):
    """Batch PSNR and SSIM (torchmetrics defaults) -> device tensor [2]."""
    _require_gpu(pred, "psnr_ssim")
    p, t = pred.contiguous().float(), target.contiguous().float()
    B, _, H, W = p.shape
    nws = L.lib.mdie_metrics_workspace_bytes(B, H, W)
    ws = torch.empty(nws, dtype=torch.uint8, device=p.device)
    out = torch.empty(2, dtype=torch.float32, device=p.device)
    L.check(L.lib.mdie_psnr_ssim(B, H, W, p.data_ptr(), t.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, _stream_ptr(p.device)),
            "mdie_psnr_ssim")
    return out


# ---- training loss (utils/loss_factory.py:146-230), value + gradient in one HIP call -------------------------------------
class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, terms):
        p, t = pred.detach().contiguous().float(), target.detach().contiguous().float()
        B, ch, H, W = p.shape
        if ch != 3:
            raise L.MdieError(f"fused_loss: 3-channel images only, got {ch}")
        arr = (L.LossTerm * len(terms))(*[L.LossTerm(L.LOSS_KINDS[n], float(w), float(a)) for n, w, a in terms])
        nws = L.lib.mdie_loss_workspace_bytes(B, H, W)
        from .train import _raw, _raw_like       # (uninitialised allocations of the training step: NaN-filled under MDIE_TRAIN_POISON=1)
        ws = _raw(nws, dtype=torch.uint8, device=p.device)
        values = _raw(len(terms) + 1, dtype=torch.float32, device=p.device)
        grad = _raw_like(p) if pred.requires_grad else None
        L.check(L.lib.mdie_loss_fwd_bwd(B, H, W, p.data_ptr(), t.data_ptr(), arr, len(terms), values.data_ptr(),
                                        grad.data_ptr() if grad is not None else None, ws.data_ptr(), nws, _stream_ptr(p.device)),
                "mdie_loss_fwd_bwd")
        ctx.grad = grad
        ctx.mark_non_differentiable(values)
        return values[len(terms)].clone(), values

    @staticmethod
    def backward(ctx, g_total, _g_values):
        g = ctx.grad
        ctx.grad = None
        return (g * g_total if g is not None else None), None, None


def fused_loss(pred, target, terms):
    """terms: [(name, weight, param)], name in lib.LOSS_KINDS; param = eps (charbonnier) / to_gray (gradient_l1).
    Returns (total, values): total is differentiable w.r.t. pred; values[k] is term k unweighted, values[-1] = total."""
    _require_gpu(pred, "fused_loss")
    return _LossFn.apply(pred, target, tuple(terms))


# ---- serving loop: uint8 batches in, uint8 batches out, several in flight -------------------------------------------------
class ServingLoop:
    """The test phase as a stream of batches (models/model.py:96-131 runs it one batch at a time, float tensors over PCIe):
    pinned uint8 HWC batch on the host -> async H2D -> normalise on the GPU (feed_uint8) -> network -> post-processing (optional) -> uint8
    HWC on the GPU -> async D2H into pinned memory, with `depth` batches in flight on `depth` streams (each stream has its own engine,
    workspace and side streams: modules.CDAN._engine is per stream), so one batch's transfers and HBM-bound tail overlap the next one's
    MFMA-bound head.  Four times fewer PCIe bytes than float32 tensors, and on one MI355X 25.6 k images/s at 256x256, batch 32, bf16
    (tools/bench_e2e.py) against ~31 k with the inputs resident in HBM.

        loop = ServingLoop(net, depth=3)
        for out_u8 in loop.run(batches):      # batches: an iterable of uint8 [B,H,W,3] host tensors (any B, H, W: buffers follow)
            ...                               # out_u8: uint8 [B,H,W,3] in the slot's pinned host buffer: valid until the next item is requested

    Results arrive in submission order and are bit-identical to the one-batch-at-a-time path (tests/test_gpu_parity.py).  The process
    should sit on its GPU's NUMA node before the pinned buffers are first touched (host.bind_to_gpu_numa: done here)."""

    def __init__(self, net, depth=3, postprocessing=None, device=None):
        from . import host as H
        self.net = net.eval()
        self.device = torch.device(device) if device is not None else next(net.parameters()).device
        if self.device.type != "cuda":
            raise L.MdieError(f"ServingLoop needs the network on a GPU, got {self.device} (no CPU fallback)")
        H.bind_to_gpu_numa(self.device.index if self.device.index is not None else torch.cuda.current_device())
        H.cap_cpu_threads()
        self.depth = int(depth)
        self.pp = postprocessing
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.depth)]
        self._in = [None] * self.depth
        self._out = [None] * self.depth
        self._done = [None] * self.depth

    def _slot_buffers(self, k, shape):
        if self._in[k] is None or tuple(self._in[k].shape) != tuple(shape):
            self._in[k] = torch.empty(shape, dtype=torch.uint8).pin_memory()
            self._out[k] = torch.empty(shape, dtype=torch.uint8).pin_memory()
        return self._in[k], self._out[k]

    def submit(self, k, batch_u8):
        """enqueue one batch on slot k (its previous batch must have been collected)"""
        if batch_u8.dtype != torch.uint8 or batch_u8.dim() != 4 or batch_u8.shape[-1] != 3:
            raise L.MdieError(f"ServingLoop: uint8 [B,H,W,3] batches, got {batch_u8.dtype} {tuple(batch_u8.shape)}")
        hin, hout = self._slot_buffers(k, batch_u8.shape)
        if batch_u8.is_pinned() and batch_u8.is_contiguous():
            hin = batch_u8                                   # the caller decodes into pinned memory of its own: no staging copy
        else:
            import numpy as np                               # one plain memcpy on this thread: torch's copy_ fans 6 MB out over every core the
            np.copyto(hin.numpy(), batch_u8.contiguous().numpy())   # host REPORTS (256 on a box that grants 16) and took 60 ms for what takes 1
        with torch.cuda.stream(self.streams[k]), torch.no_grad():
            x = feed_uint8(hin.to(self.device, non_blocking=True))
            y = self.net(x)
            u8 = apply_postprocessing(y, self.pp, want_uint8=True)[1] if (self.pp and self.pp.get("enabled", False)) else to_uint8_hwc(y)
            hout.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.streams[k])
        self._done[k] = ev

    def collect(self, k):
        self._done[k].synchronize()
        self._done[k] = None
        return self._out[k]

    def run(self, batches):
        """generator over the outputs, in order; keeps `depth` batches in flight"""
        pending = []                                         # slots in submission order
        i = 0
        for b in batches:
            k = i % self.depth
            if len(pending) == self.depth:
                yield self.collect(pending.pop(0))
            self.submit(k, b)
            pending.append(k)
            i += 1
        while pending:
            yield self.collect(pending.pop(0))
