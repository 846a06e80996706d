"""Synthetic workload for benchmarks and demos: seeded random-initialised checkpoints and low-light image batches.

There are no trained weights or datasets offline, so `bench.py` and `tools/` run the engine on
  * `make_state_dict(seed)`   any `name -> shape` layout (default: the CDAN checkpoint layout of `arch.py`) filled
                              from numpy's PCG64 stream -- torch-default-like uniform weights/biases, BatchNorm affine
                              terms and running statistics RANDOMISED (a fresh eval BatchNorm is the identity and would
                              make the folded path trivially right);
  * `lowlight_batch(...)`     degraded/clean pairs in the recipe SURVEY.md 8(d) fixes: clean = 5x5 box-blurred uniform
                              noise, degraded = clean * U(0.05, 0.4) per image, both quantised to 8 bits.
The CPU oracle's own generator (`oracle/params.py`, test infrastructure) produces bit-identical values from the same
seed -- `tests/test_host_cpu.py` holds the two to each other -- so the engine and its checker always see one dataset.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import arch


def _fan_in(shape):
    # torch's default init: weight.size(1) * receptive field (also for ConvTranspose2d, whose dim 1 is Cout)
    return shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)


def fill_layout(layout, seed, randomize_bn=True):
    """layout: OrderedDict name -> shape (torch naming: `<bn>.running_mean` marks a BatchNorm)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    is_bn = lambda name: name.rsplit(".", 1)[0] + ".running_mean" in layout
    out, fan = OrderedDict(), 1
    for name, shape in layout.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.tensor(0, dtype=torch.int64)
            continue
        if leaf == "running_mean":
            v = rng.normal(0.0, 0.2, shape) if randomize_bn else np.zeros(shape)
        elif leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape) if randomize_bn else np.ones(shape)
        elif len(shape) == 1 and is_bn(name):
            if leaf == "weight":
                v = rng.uniform(0.6, 1.4, shape) if randomize_bn else np.ones(shape)
            else:
                v = rng.normal(0.0, 0.15, shape) if randomize_bn else np.zeros(shape)
        elif leaf == "weight":
            fan = _fan_in(shape)
            bound = np.sqrt(3.0 / fan)
            v = rng.uniform(-bound, bound, shape)
        else:                                   # bias of the convolution / linear layer just filled
            bound = 1.0 / np.sqrt(fan)
            v = rng.uniform(-bound, bound, shape)
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def make_state_dict(seed=42, layout=None, randomize_bn=True):
    if layout is None:
        layout = OrderedDict((k, shape) for k, (shape, _) in arch.cdan_param_spec().items())
    return fill_layout(layout, seed, randomize_bn)


def lowlight_batch(seed, b, h, w):
    """-> (degraded, clean), float32 [b,3,h,w] in [0,1] on the CPU."""
    rng = np.random.Generator(np.random.PCG64(seed))
    noise = rng.random((b, 3, h + 4, w + 4), dtype=np.float32)
    c = np.pad(np.cumsum(np.cumsum(noise, axis=2), axis=3), ((0, 0), (0, 0), (1, 0), (1, 0)))   # summed-area table
    box = (c[:, :, 5:, 5:] - c[:, :, :-5, 5:] - c[:, :, 5:, :-5] + c[:, :, :-5, :-5]) / 25.0
    box = (box - box.min()) / (box.max() - box.min())
    gain = rng.uniform(0.05, 0.4, (b, 1, 1, 1)).astype(np.float32)
    degraded = np.round(box * gain * 255.0) / 255.0
    clean = np.round(box * 255.0) / 255.0
    return torch.from_numpy(degraded.astype(np.float32)), torch.from_numpy(clean.astype(np.float32))
