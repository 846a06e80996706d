"""Deterministic parameter sets for the CDAN oracle and the parity tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).

`cdan_spec()` restates the checkpoint layout of the reference network
(`/root/reference/models/cdan.py:55-176`, `/root/reference/models/cbam.py:6-95`):
236 entries = 140 parameters + 64 BatchNorm running statistics + 32
`num_batches_tracked` counters.  `tests/golden/make_golden.py` asserts that the
names, shapes and order equal `CDAN().state_dict()` of the imported reference,
and commits them as `tests/golden/state_dict_spec.json`.

`make_state_dict(seed)` fills that layout from numpy's PCG64 stream (stable
across platforms and numpy versions), so the 14 MB of weights never have to be
committed: the GPU box regenerates bit-identical values from the seed, and
`tests/golden/params_checksum.json` guards against drift.  BatchNorm affine
terms and running statistics are randomised because a freshly initialised
eval-mode BatchNorm is the identity and would hide folding mistakes.
"""
from collections import OrderedDict

import numpy as np
import torch

GROWTH = 16
DENSE_LAYERS = 4
REDUCTION = 16


def _bn(spec, prefix, c):
    spec[prefix + ".weight"] = (c,)
    spec[prefix + ".bias"] = (c,)
    spec[prefix + ".running_mean"] = (c,)
    spec[prefix + ".running_var"] = (c,)
    spec[prefix + ".num_batches_tracked"] = ()


def _conv_block(spec, prefix, cin, cout):
    spec[prefix + ".conv.weight"] = (cout, cin, 3, 3)
    spec[prefix + ".conv.bias"] = (cout,)
    _bn(spec, prefix + ".bn", cout)


def _dense_block(spec, prefix, cin, cout):
    c = cin
    for i in range(DENSE_LAYERS):
        _bn(spec, f"{prefix}.layers.{i}.0", c)
        spec[f"{prefix}.layers.{i}.2.weight"] = (GROWTH, c, 3, 3)
        spec[f"{prefix}.layers.{i}.2.bias"] = (GROWTH,)
        c += GROWTH
    _bn(spec, prefix + ".transition_layer.0", c)
    spec[prefix + ".transition_layer.2.weight"] = (cout, c, 1, 1)
    spec[prefix + ".transition_layer.2.bias"] = (cout,)


def _cbam(spec, prefix, c):
    h = c // REDUCTION
    spec[prefix + ".ChannelGate.mlp.1.weight"] = (h, c)
    spec[prefix + ".ChannelGate.mlp.1.bias"] = (h,)
    spec[prefix + ".ChannelGate.mlp.3.weight"] = (c, h)
    spec[prefix + ".ChannelGate.mlp.3.bias"] = (c,)
    spec[prefix + ".SpatialGate.spatial.conv.weight"] = (1, 2, 7, 7)
    _bn(spec, prefix + ".SpatialGate.spatial.bn", 1)


def cdan_spec():
    """name -> shape, in the reference's `state_dict()` order."""
    spec = OrderedDict()
    widths = [3, 64, 128, 256, 512]
    for i in range(4):
        _conv_block(spec, f"encoder.conv{i + 1}", widths[i], widths[i + 1])
    for i in range(3):
        _dense_block(spec, f"encoder.dense{i + 1}", widths[i + 1], widths[i + 1])
    _cbam(spec, "bottleneck", 512)
    dec = [512, 256, 128, 64, 3]
    for i in range(4):
        # ConvTranspose2d stores its weight as [Cin, Cout, kh, kw] (cdan.py:103,107,111,115)
        spec[f"decoder.conv{i + 1}.weight"] = (dec[i], dec[i + 1], 3, 3)
        spec[f"decoder.conv{i + 1}.bias"] = (dec[i + 1],)
        if i < 3:
            _cbam(spec, f"decoder.cbam{i + 1}", dec[i + 1])
        _bn(spec, f"decoder.bn{i + 1}", dec[i + 1])
    _dense_block(spec, "decoder.final_dense", 3, 3)
    return spec


def _fan_in(name, shape):
    # torch's default init uses weight.size(1) * receptive field, also for
    # ConvTranspose2d (whose dim 1 is Cout).
    if len(shape) == 4:
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    raise ValueError(name)


def fill_spec(spec, seed, randomize_bn=True, weight_gain=1.0):
    """Fill any name->shape layout that follows torch's naming conventions."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    last_fan = 1
    for name, shape in spec.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[name] = torch.tensor(0, dtype=torch.int64)
            continue
        if leaf == "running_mean":
            v = rng.normal(0.0, 0.2, shape) if randomize_bn else np.zeros(shape)
        elif leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape) if randomize_bn else np.ones(shape)
        elif len(shape) == 1 and (name.endswith(".bn.weight") or _is_bn_affine(name, spec)):
            if leaf == "weight":
                v = rng.uniform(0.6, 1.4, shape) if randomize_bn else np.ones(shape)
            else:
                v = rng.normal(0.0, 0.15, shape) if randomize_bn else np.zeros(shape)
        elif leaf == "weight":
            last_fan = _fan_in(name, shape)
            bound = weight_gain * np.sqrt(3.0 / last_fan)  # unit-gain uniform
            v = rng.uniform(-bound, bound, shape)
        else:  # conv / linear bias
            bound = 1.0 / np.sqrt(last_fan)
            v = rng.uniform(-bound, bound, shape)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def _is_bn_affine(name, spec):
    stem = name.rsplit(".", 1)[0]
    return (stem + ".running_mean") in spec


def make_state_dict(seed=42, randomize_bn=True):
    return fill_spec(cdan_spec(), seed, randomize_bn=randomize_bn)


def checksum(sd):
    """Order-dependent float64 checksum of a state dict (drift guard)."""
    acc = 0.0
    for i, (k, v) in enumerate(sd.items()):
        a = v.detach().double().reshape(-1)
        if a.numel():
            w = torch.arange(1, a.numel() + 1, dtype=torch.float64) % 97 + 1.0
            acc += float((a * w).sum()) * (1 + (i % 13))
    return acc


def lowlight_batch(seed, b, h, w):
    """Synthetic degraded/clean pair in the recipe SURVEY.md 8(d) fixes:
    clean = 5x5 box-blurred uniform noise, degraded = clean * U(0.05, 0.4) per
    image, quantised to 8 bits (cf. the low-light synthesis at
    /root/reference/datasets_generation/generate_paired_degradation_dataset.py:119-122)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    clean = rng.random((b, 3, h + 4, w + 4), dtype=np.float32)
    c = np.cumsum(np.cumsum(clean, axis=2), axis=3)
    c = np.pad(c, ((0, 0), (0, 0), (1, 0), (1, 0)))
    box = (c[:, :, 5:, 5:] - c[:, :, :-5, 5:] - c[:, :, 5:, :-5] + c[:, :, :-5, :-5]) / 25.0
    box = (box - box.min()) / (box.max() - box.min())
    f = rng.uniform(0.05, 0.4, (b, 1, 1, 1)).astype(np.float32)
    deg = np.round(box * f * 255.0) / 255.0
    cl = np.round(box * 255.0) / 255.0
    return torch.from_numpy(deg.astype(np.float32)), torch.from_numpy(cl.astype(np.float32))
