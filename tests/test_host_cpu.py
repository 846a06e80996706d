"""CPU: the boundary caller's host logic (config parsing, factory, datasets, device-tail detection)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from mdie_amd import host as H


def _write_pairs(root, n=5, size=(40, 56)):
    rng = np.random.default_rng(0)
    for sub in ("degraded", "clean"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in range(n):
        a = rng.integers(0, 256, (*size, 3), dtype=np.uint8)
        Image.fromarray(a).save(os.path.join(root, "clean", f"im{i}.png"))
        Image.fromarray((a * 0.5).astype(np.uint8)).save(os.path.join(root, "degraded", f"im{i}.png"))
    Image.fromarray(a).save(os.path.join(root, "clean", "orphan.png"))


def test_config_comments_missing_keys_and_phase(tmp_path):
    p = tmp_path / "c.json"
    p.write_text('{\n  // a comment\n  "name": "x", // trailing\n  "model": {"networks": [{"name": ["models.cdan", "CDAN"], "args": {}}]}\n}\n')
    from utils.parser import parse
    cfg = parse(argparse.Namespace(config=str(p), phase="test"))
    assert cfg["phase"] == "test" and cfg["name"] == "x"
    assert cfg["nope"] is None and cfg["model"]["absent"] is None
    assert cfg["model"]["networks"][0]["also_absent"] is None


def test_factory_builds_the_hip_network_and_reports_failures():
    from utils.parser import define_network
    net = define_network({"name": ["models.cdan", "CDAN"], "args": {}})
    assert type(net).__name__ == "CDAN" and net.__name__ == "CDAN" and len(net.state_dict()) == 236
    with pytest.raises(NotImplementedError, match="not recognized"):
        define_network({"name": ["models.cdan", "NoSuchNet"], "args": {}})


def test_paired_dataset_modes_and_device_tail(tmp_path):
    _write_pairs(str(tmp_path))
    tf = {"backend": "albumentations", "ops": [{"name": "Resize", "args": {"height": 32, "width": 48}},
                                               {"name": "Normalize", "args": {"mean": [0, 0, 0], "std": [1, 1, 1]}},
                                               {"name": "ToTensorV2", "args": {}}]}
    ds = H.ImageFolderPairs(str(tmp_path / "degraded"), str(tmp_path / "clean"), "filename", tf)
    assert len(ds) == 5                                    # the orphan target is dropped
    x, t = ds[0]
    assert x.dtype == torch.uint8 and tuple(x.shape) == (32, 48, 3)   # Normalize(0,1)+ToTensorV2 deferred to the GPU
    tf2 = {"backend": "albumentations", "ops": [{"name": "Normalize", "args": {"mean": [0.5, 0.5, 0.5], "std": [0.5, 0.5, 0.5]}},
                                                {"name": "ToTensorV2", "args": {}}]}
    x2, _ = H.ImageFolderPairs(str(tmp_path / "degraded"), str(tmp_path / "clean"), "stem", tf2)[1]
    assert x2.dtype == torch.float32 and tuple(x2.shape) == (3, 40, 56) and x2.min() >= -1 and x2.max() <= 1
    with pytest.raises(ValueError):
        H.ImageFolderPairs(str(tmp_path / "degraded"), str(tmp_path / "clean"), "nearest")
    with pytest.raises(ValueError, match="not supported"):
        H.ImageFolderPairs(str(tmp_path / "degraded"), str(tmp_path / "clean"), "filename",
                           {"ops": [{"name": "CLAHE", "args": {}}]})[0]


def test_example_config_parses():
    cfg = H.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "example_noise_64.json"), "test")
    assert cfg["model"]["networks"][0]["name"] == ["models.cdan", "CDAN"]
    assert cfg["test"]["dataloader"]["args"]["batch_size"] == 4


def test_package_synthetic_workload_equals_the_oracle_generators():
    """bench.py / tools use mdie_amd.synthetic; the oracle and the tests use oracle.params: one dataset, bit for bit"""
    import torch
    from mdie_amd import router as R
    from mdie_amd import synthetic as S
    from oracle import params as P
    a, b = S.make_state_dict(42), P.make_state_dict(42)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    (d1, c1), (d2, c2) = S.lowlight_batch(1000, 2, 32, 40), P.lowlight_batch(1000, 2, 32, 40)
    assert torch.equal(d1, d2) and torch.equal(c1, c2)
    r1, r2 = S.make_state_dict(7, R.router_param_spec()), P.fill_spec(R.router_param_spec(), 7, randomize_bn=True)
    assert all(torch.equal(r1[k], r2[k]) for k in r1)
