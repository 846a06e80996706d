"""The oracle (oracle/cdan_oracle.py) against the vectors the reference itself produced
(tests/golden/*.npz, written by tests/golden/make_golden.py in the build container).

This is what pins the oracle: every parity test on the GPU compares the HIP path with
either these vectors directly or with the oracle on fresh seeded inputs.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cdan_oracle as O
from oracle import params as P

TOL = 2e-6  # same ATen CPU kernels underneath; only op-ordering differences remain


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("p:") and z[k].dtype == np.float32}
    params = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p:")}
    return arrays, params


def _close(a, b, tol=TOL):
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-6)
    assert err / scale <= tol, f"max err {err:.3e} rel-to-max {err / scale:.3e}"


def test_state_dict_spec_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_spec.json")) as f:
        ref = json.load(f)
    spec = P.cdan_spec()
    assert len(ref) == 236 and len(spec) == 236
    assert [r[0] for r in ref] == list(spec.keys())
    for (k, shape, dtype), (k2, shape2) in zip(ref, spec.items()):
        assert tuple(shape) == tuple(shape2), k
        assert dtype == ("int64" if k.endswith("num_batches_tracked") else "float32")
    n_params = sum(int(np.prod(s)) for k, s in spec.items()
                   if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 3585663


def test_seeded_params_are_stable(golden_dir):
    with open(os.path.join(golden_dir, "params_checksum.json")) as f:
        ref = json.load(f)
    sd = P.make_state_dict(ref["seed"])
    assert sum(v.numel() for v in sd.values()) == ref["numel"]
    assert P.checksum(sd) == pytest.approx(ref["checksum"], rel=1e-12)


@pytest.mark.parametrize("tag", ["1x32x32", "1x40x56", "2x64x64_lowlight", "4x64x64_noise"])
def test_e2e_eval(golden_dir, tag):
    g, _ = _load(golden_dir, f"e2e_eval_{tag}.npz")
    sd = P.make_state_dict(42)
    taps = {}
    with torch.no_grad():
        y = O.cdan_forward(sd, g["x"], taps=taps)
    _close(y, g["y"])
    for k in ("enc", "bott", "skip0", "skip1", "skip2", "dense0", "dense1", "dense2"):
        if k in g:
            _close(taps[k], g[k])


@pytest.mark.parametrize("cin,cout", [(16, 32), (3, 16), (32, 64)])
def test_conv_block(golden_dir, cin, cout):
    g, p = _load(golden_dir, f"op_convblock_{cin}_{cout}.npz")
    sd = {"m." + k: v for k, v in p.items()}
    y = O.conv_block(sd, "m", g["x"])
    _close(y, g["y"])
    _close(torch.nn.functional.max_pool2d(y, 2, 2), g["y_pool"])


@pytest.mark.parametrize("cin", [16, 32, 3])
def test_dense_block(golden_dir, cin):
    g, p = _load(golden_dir, f"op_denseblock_{cin}.npz")
    sd = {"m." + k: v for k, v in p.items()}
    _close(O.dense_block(sd, "m", g["x"]), g["y"])


@pytest.mark.parametrize("c", [32, 64, 256])
def test_cbam(golden_dir, c):
    g, p = _load(golden_dir, f"op_cbam_{c}.npz")
    sd = {"m." + k: v for k, v in p.items()}
    _close(O.channel_gate(sd, "m.ChannelGate", g["x"]), g["y_channel"])
    _close(O.cbam(sd, "m", g["x"]), g["y"])


def test_up2_add(golden_dir):
    g, _ = _load(golden_dir, "op_up2_add.npz")
    _close(O.up2(g["lo"]) + g["skip"], g["y"])


def test_train_step(golden_dir):
    z = np.load(os.path.join(golden_dir, "train_step_32.npz"))
    sd = {k: v.clone().requires_grad_(v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")))
          for k, v in P.make_state_dict(42).items()}
    stats = {}
    y = O.cdan_forward(sd, torch.from_numpy(z["x"]), bn_mode="batch", stats_out=stats)
    loss = O.charbonnier(y, torch.from_numpy(z["t"]))
    loss.backward()
    _close(y.detach(), torch.from_numpy(z["y"]), 1e-5)
    assert loss.item() == pytest.approx(float(z["loss"]), rel=1e-6)
    for k in z.files:
        if k.startswith("g:"):
            _close(sd[k[2:]].grad, torch.from_numpy(z[k]), 2e-4)
        if k.startswith("s:"):
            _close(stats[k[2:]], torch.from_numpy(z[k]), 1e-5)
    norms = json.loads(str(z["grad_norms"]))
    for k, n in norms.items():
        assert float(sd[k].grad.double().norm()) == pytest.approx(n, rel=2e-3, abs=1e-6), k  # biases feeding a batch-stat BN have ~0 gradient


def test_ddp_two_shard_gradients(golden_dir):
    """the per-shard training steps of an N=2 data-parallel run (each rank: forward, charbonnier
    mean over ITS shard, backward -- models/model.py:159-164).  The oracle must reproduce each shard's gradients; their
    mean is what the gradient exchange delivers (tests/test_bench_sharding_cpu.py, tests/test_gpu_parity.py)."""
    z = np.load(os.path.join(golden_dir, "ddp_2shard_32.npz"))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    for r, sl in enumerate((slice(0, 2), slice(2, 4))):
        sd = {k: v.clone().requires_grad_(v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")))
              for k, v in P.make_state_dict(42).items()}
        y = O.cdan_forward(sd, x[sl], bn_mode="batch", stats_out={})
        loss = O.charbonnier(y, t[sl])
        loss.backward()
        _close(y.detach(), torch.from_numpy(z[f"y{r}"]), 1e-5)
        assert loss.item() == pytest.approx(float(z[f"loss{r}"]), rel=1e-6)
        for k in z.files:
            if k.startswith(f"g{r}:"):
                _close(sd[k[3:]].grad, torch.from_numpy(z[k]), 2e-4)
