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


def _reference_configs():
    """every task config of the reference: read from /root/reference where it exists (the build container), else from
    the digest tests/golden/make_golden.py wrote from it"""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    paths = sorted(glob.glob("/root/reference/config/*.json"))
    if paths:
        return {os.path.basename(p): {ph: H.load_config(p, ph) for ph in ("train", "test")} for p in paths}
    with open(os.path.join(here, "golden", "reference_config_digest.json")) as f:
        digest = json.load(f)
    out = {}
    for name, e in digest.items():
        out[name] = {}
        for ph in ("train", "test"):
            out[name][ph] = H._wrap({"phase": ph, "model": e["model"], "loss": e["loss"], "metrics": e["metrics"],
                                     "post_processing": e["post_processing"],
                                     ph: {"dataset": {"name": e[ph]["dataset_name"], "args": {"transform": e[ph]["transform"]}},
                                          "dataloader": e[ph]["dataloader"]}})
    return out


def test_every_reference_config_drives_both_phases(tmp_path):
    """`run.py -c config/<task>.json -p train|test` for all 11 task configs: the transform list of each phase is accepted
    and runs on a sample pair (RandomGamma / RandomBrightnessContrast included, config/low_light.json:102-103), the loss
    section builds (network-bound terms skipped with a warning), the factory names resolve to this repository's classes."""
    import warnings
    _write_pairs(str(tmp_path), n=2)
    cfgs = _reference_configs()
    assert len(cfgs) == 11
    seen_ops = set()
    for name, phases in cfgs.items():
        for ph, cfg in phases.items():
            ds = cfg[ph]["dataset"]
            tf = ds["args"]["transform"]
            seen_ops |= {o["name"] for o in tf["ops"]}
            pairs = H.instantiate({"name": ds["name"], "args": {"input_root": str(tmp_path / "degraded"), "target_root": str(tmp_path / "clean"),
                                                                "pairing_mode": "filename", "transform": tf}}, default_module="data", kind="Dataset")
            for rep in range(8):                      # the random ops all get to fire
                x, t = pairs[rep % 2]
                assert x.dtype == torch.uint8 and tuple(x.shape) == (256, 384, 3) and tuple(t.shape) == (256, 384, 3), (name, ph)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                assert len(H.build_losses(cfg.get("loss"))) >= 1, name
            net_name = cfg["model"]["networks"][0]["name"]
            assert net_name == ["models.cdan", "CDAN"] and cfg["model"]["which_model"]["name"] == ["models.model", "Model"]
    assert {"RandomGamma", "RandomBrightnessContrast", "RandomRotate90", "VerticalFlip", "HorizontalFlip", "Resize"} <= seen_ops


def test_photometric_ops_share_parameters_between_input_and_target():
    """albumentations' additional_targets={"target": "image"} (utils/transforms_factory.py:85): one draw per sample"""
    tf = H._Transform({"ops": [{"name": "RandomGamma", "args": {"gamma_limit": [70, 130], "p": 1.0}},
                               {"name": "RandomBrightnessContrast", "args": {"brightness_limit": 0.1, "contrast_limit": 0.1, "p": 1.0}}]}, device_tail=False)
    img = Image.fromarray(np.tile(np.arange(256, dtype=np.uint8)[None, :, None], (4, 1, 3)))
    a, b = tf(img, img, rng=np.random.default_rng(3))
    assert torch.equal(a, b) and not torch.equal(a, torch.from_numpy(np.asarray(img)).permute(2, 0, 1).float())
    assert (a[0, 0, 1:] >= a[0, 0, :-1]).all()                 # both ops are monotonic look-up tables


def test_resize_is_plain_bilinear_without_antialiasing():
    """albumentations Resize = cv2.INTER_LINEAR: half-pixel centres, two taps per axis also when shrinking"""
    x = np.zeros((8, 8, 3), np.uint8)
    x[:, 4:] = 200
    y = H._resize_bilinear(x, 4, 4)               # 2x shrink: output column j samples between input columns 2j and 2j+1
    assert y.shape == (4, 4, 3) and (y[:, :2] == 0).all() and (y[:, 2:] == 200).all()
    z = H._resize_bilinear(x, 8, 16)
    assert z.shape == (8, 16, 3) and z[0, 0, 0] == 0 and z[0, -1, 0] == 200


def test_workers_and_epochs_draw_different_augmentations(tmp_path):
    _write_pairs(str(tmp_path), n=4)
    tf = {"ops": [{"name": "HorizontalFlip", "args": {"p": 0.5}}, {"name": "VerticalFlip", "args": {"p": 0.5}},
                  {"name": "Normalize", "args": {"mean": [0, 0, 0], "std": [1, 1, 1]}}, {"name": "ToTensorV2", "args": {}}]}
    ds = H.ImageFolderPairs(str(tmp_path / "degraded"), str(tmp_path / "clean"), "filename", tf)
    torch.manual_seed(1)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, num_workers=2)
    epochs = [torch.cat([x for x, _ in loader]) for _ in range(3)]
    assert not (torch.equal(epochs[0], epochs[1]) and torch.equal(epochs[1], epochs[2]))   # (was: every epoch identical)


def test_module_fingerprint_sees_every_kind_of_change():
    """modules._fingerprint (decides when the eval engine repacks its weights): in-place parameter and buffer updates, the
    module's own epoch counter (graph replays that rewrite weights without bumping versions), and a re-registered tensor --
    the tensor list is cached on the module (the tree walk cost 0.5 ms per forward) and must notice a changed SET too."""
    import torch
    from models.cdan import CDAN
    from mdie_amd import modules as M
    net = CDAN()
    a = M._fingerprint(net)
    assert M._fingerprint(net) == a
    with torch.no_grad():
        net.encoder.conv1.conv.weight.add_(1.0)
    b = M._fingerprint(net)
    assert b != a
    net.encoder.conv1.bn.running_mean.zero_()
    c = M._fingerprint(net)
    assert c not in (a, b)
    net._mdie_epoch += 1
    d = M._fingerprint(net)
    assert d not in (a, b, c)
    w = net.decoder.conv4.weight
    net.decoder.conv4.weight = torch.nn.Parameter(w.detach().clone())          # same values, same version count, other tensor
    e = M._fingerprint(net)
    assert e not in (a, b, c, d) and M._fingerprint(net) == e
    net.load_state_dict(CDAN().state_dict())                                    # copies in place: versions move
    assert M._fingerprint(net) != e


def test_timeline_tools_parse_a_rocprofv3_kernel_trace(tmp_path):
    """tools/train_timeline.py and tools/infer_timeline.py on a hand-made rocprofv3 kernel trace (the column set of rocprofv3 1.x):
    steps are cut at their marker kernels, launch order, queue and overlap come out."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path / "run" / "host"
    d.mkdir(parents=True)
    hdr = '"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"\n'
    rows, t, n = [], 1000, 0

    def k(name, q, start, dur):
        nonlocal n
        n += 1
        rows.append(f'"KERNEL_DISPATCH","Agent 2",{q},0,1,{n},1,"{name}",{n},{start},{start + dur},0,0,64,0,32,256,1,1,2048,1,1\n')

    for step in range(12):
        base = t + step * 100000
        k("void mdie::nchw3_to_nhwc16_kernel<bf16>(int)", 1, base, 2000)                       # training marker
        k("_ZN4mdie22conv_first_pool_kernelIDF16bLb1EEEvNS_9FirstArgsE", 1, base + 3000, 30000)      # inference marker
        k("_ZN4mdie16conv_wide_kernelIDF16bLi1ELb1ELb0ELi4EEEvNS_8WideArgsE", 1, base + 33000, 60000)
        k("_ZN4mdie11conv_kernelIDF16bLi3ELi16ELi16ELb0ELb0EEEvNS_8ConvArgsE", 2, base + 40000, 20000)   # a side-queue kernel under the wide one
    (d / "1_kernel_trace.csv").write_text(hdr + "".join(rows))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "train_timeline.py"), str(tmp_path / "run")], capture_output=True, text=True, check=True).stdout
    assert "4 kernels" in out and "conv_wide_kernel" in out and "nchw3_to_nhwc16_kernel" in out
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "infer_timeline.py"), str(tmp_path / "run")], capture_output=True, text=True, check=True).stdout
    assert "on 2 queues" in out
    wide = [l for l in out.splitlines() if "conv_wide_kernel" in l and "|" in l][0]
    assert wide.split("|")[1].strip().startswith("conv_kernel")              # the overlap column names the side-queue kernel
