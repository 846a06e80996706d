#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels
to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's `models.cdan` / `models.cbam` on CPU, loads the
deterministic parameter sets of `oracle.params`, pushes seeded inputs through
the reference modules and stores inputs + expected outputs as small .npz
files.  Only data is written: no reference source text is copied.

Files written
  state_dict_spec.json    names/shapes/dtypes of CDAN().state_dict() (236 entries)
  params_checksum.json    checksum of oracle.params.make_state_dict(42)
  e2e_eval_*.npz          whole-network eval outputs + stage taps
  op_*.npz                per-op vectors (ConvBlock, DenseBlock, CBAM, ConvTranspose2d,
                          bilinear x2 + add, maxpool)
  train_step_32.npz       batch-stat BN forward, charbonnier loss, selected gradients,
                          updated running statistics (dropout disabled)
  e2e_dec_taps.npz        decoder stage outputs (dec1..dec4) of two of the e2e_eval inputs, taken from the reference's
                          own forward with hooks            [python make_golden.py dectaps]
  ddp_2shard_32.npz       data-parallel semantics: one batch of 4 split into two shards,
                          per-shard gradients of the reference's training step (models/model.py:159-164) for 12
                          parameters -- their mean is what an N=2 gradient all-reduce + /world must reproduce
                                                                                 [python make_golden.py ddp]

  reference_config_digest.json   transform / loss / metric / post-processing sections of the reference's 11 configs
                                                                                 [python make_golden.py configs]

`python make_golden.py` regenerates everything; a section name regenerates only that file.
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from models.cbam import CBAM  # noqa: E402  (reference)
from models.cdan import CDAN, ConvBlock, DenseBlock  # noqa: E402  (reference)

from oracle import params as P  # noqa: E402

torch.set_num_threads(8)


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def module_spec(m):
    return OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())


def load_seeded(m, seed):
    sd = P.fill_spec(module_spec(m), seed)
    m.load_state_dict(sd, strict=True)
    m.eval()
    return sd


def pack_params(sd):
    return {"p:" + k: v.numpy() for k, v in sd.items()}


GRAD_KEYS = ("encoder.conv1.conv.weight", "encoder.conv1.bn.weight", "encoder.dense1.layers.1.2.weight",
             "encoder.dense2.transition_layer.2.weight", "bottleneck.ChannelGate.mlp.1.weight",
             "bottleneck.SpatialGate.spatial.conv.weight", "decoder.conv1.bias", "decoder.conv3.weight",
             "decoder.bn4.bias", "decoder.cbam2.ChannelGate.mlp.3.bias",
             "decoder.final_dense.layers.3.2.weight", "decoder.final_dense.transition_layer.2.weight")


def train_mode_reference(sd):
    """the reference network as Model.train_step runs it (models/model.py:144), Dropout modules left in eval (their
    RNG cannot be matched bit for bit)"""
    ref = CDAN()
    ref.load_state_dict(sd, strict=True)
    ref.train()
    for mod in ref.modules():
        if isinstance(mod, nn.Dropout):
            mod.eval()
    return ref


def dec_taps():
    """decoder taps through the reference's OWN forward: `out *= denses[k]` (models/cdan.py:133,141,149) multiplies the
    CBAM module's output tensor in place, so a forward hook that keeps that tensor (no clone) holds the stage output
    once the forward has finished; the input of final_dense (:154-155) is taken by a pre-hook."""
    sd = P.make_state_dict(42)
    ref = CDAN()
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    arrays = {}
    for tag in ("1x32x32", "1x40x56"):
        z = np.load(os.path.join(HERE, f"e2e_eval_{tag}.npz"))
        x = torch.from_numpy(z["x"])
        kept, hooks = {}, []
        for i, name in enumerate(("cbam1", "cbam2", "cbam3")):
            hooks.append(getattr(ref.decoder, name).register_forward_hook(lambda m, inp, out, k=f"dec{i + 1}": kept.__setitem__(k, out)))
        hooks.append(ref.decoder.final_dense.register_forward_pre_hook(lambda m, inp: kept.__setitem__("dec4", inp[0])))
        with torch.no_grad():
            y = ref(x)
        for h in hooks:
            h.remove()
        assert np.array_equal(np32(y), z["y"]), "the tapped forward must reproduce the committed output bit for bit"
        for k, v in kept.items():
            arrays[f"{tag}:{k}"] = np32(v)
        arrays[f"{tag}:x"] = z["x"]
    save("e2e_dec_taps.npz", **arrays)


def ddp_fixture():
    """N = 2 data parallel: each rank runs the reference's step (forward, charbonnier loss = mean over ITS shard, backward;
    models/model.py:159-164) on its half of one batch; the exchanged gradient is the mean over ranks."""
    sd = P.make_state_dict(42)
    x, t = P.lowlight_batch(33, 4, 32, 32)
    shards = [(x[:2], t[:2]), (x[2:], t[2:])]
    out = {"x": np32(x), "t": np32(t)}
    per = []
    for r, (xs, ts) in enumerate(shards):
        ref = train_mode_reference(sd)            # identical replicas at the start of the step
        y = ref(xs)
        loss = torch.sqrt((y - ts) ** 2 + 1e-6).mean()
        loss.backward()
        named = dict(ref.named_parameters())
        per.append({k: named[k].grad.detach().clone() for k in GRAD_KEYS})
        out[f"loss{r}"] = np.float64(loss.item())
        out[f"y{r}"] = np32(y)
        for k in GRAD_KEYS:
            out[f"g{r}:" + k] = np32(per[r][k])
        if r == 0:
            out["norms0"] = np.array(json.dumps({k: float(v.grad.double().norm()) for k, v in named.items()}))
    save("ddp_2shard_32.npz", **out)      # (the exchanged gradient, (g0 + g1) / 2, is derived by the tests)


def config_digest():
    """the parts of the reference's 11 task configs the boundary caller consumes, as data: per config and phase the
    transform op list, the loss terms, metric names, post-processing ops and the ["module", "Class"] names"""
    import glob
    out = {}
    for path in sorted(glob.glob("/root/reference/config/*.json")):
        with open(path) as f:
            cfg = json.loads("\n".join(line.split("//")[0] for line in f.read().splitlines()))
        entry = {"model": cfg["model"], "loss": cfg.get("loss"), "metrics": cfg.get("metrics"), "post_processing": cfg.get("post_processing")}
        for phase in ("train", "test"):
            ds = cfg[phase]["dataset"]
            entry[phase] = {"dataset_name": ds["name"], "transform": ds["args"].get("transform"), "dataloader": cfg[phase]["dataloader"]}
        out[os.path.basename(path)] = entry
    with open(os.path.join(HERE, "reference_config_digest.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"reference_config_digest.json: {len(out)} configs")


def main():
    # ---- checkpoint layout ------------------------------------------------------------
    ref = CDAN()
    ref_sd = ref.state_dict()
    spec = P.cdan_spec()
    assert list(spec.keys()) == list(ref_sd.keys()), "state_dict key order differs"
    for k, shp in spec.items():
        assert tuple(ref_sd[k].shape) == tuple(shp), (k, shp, ref_sd[k].shape)
    with open(os.path.join(HERE, "state_dict_spec.json"), "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()], f, indent=0)

    sd = P.make_state_dict(42)
    with open(os.path.join(HERE, "params_checksum.json"), "w") as f:
        json.dump({"seed": 42, "checksum": P.checksum(sd),
                   "numel": int(sum(v.numel() for v in sd.values()))}, f)
    ref.load_state_dict(sd, strict=True)
    ref.eval()

    # ---- whole network, eval ------------------------------------------------------------
    def run_e2e(tag, x, keep_taps):
        taps = {}
        with torch.no_grad():
            e, skips, denses = ref.encoder(x)
            b = ref.bottleneck(e)
            y = ref.decoder(x, b, skips, denses)
            y2 = ref(x)
        assert torch.equal(y, y2)
        arrays = {"x": np32(x), "y": np32(y)}
        if keep_taps:
            arrays["enc"] = np32(e)
            arrays["bott"] = np32(b)
            for i in range(3):
                arrays[f"skip{i}"] = np32(skips[i])
                arrays[f"dense{i}"] = np32(denses[i])
        save(f"e2e_eval_{tag}.npz", **arrays)
        print(f"  {tag}: y min {y.min():.4f} max {y.max():.4f} std {y.std():.4f}")

    g = torch.Generator().manual_seed(1234)
    run_e2e("1x32x32", torch.rand(1, 3, 32, 32, generator=g), True)
    run_e2e("1x40x56", torch.rand(1, 3, 40, 56, generator=g), True)
    xl, _ = P.lowlight_batch(7, 2, 64, 64)
    run_e2e("2x64x64_lowlight", xl, False)
    # BASELINE configs[0] shape: 4 x 3 x 64 x 64 (noise-like input: clean + N(0, sigma))
    xn, cl = P.lowlight_batch(11, 4, 64, 64)
    xn = (cl + 0.1 * torch.randn(cl.shape, generator=g)).clamp(0, 1)
    run_e2e("4x64x64_noise", xn, False)

    # ---- per-op vectors -----------------------------------------------------------------
    with torch.no_grad():
        for cin, cout, h, w in ((16, 32, 16, 16), (3, 16, 12, 20), (32, 64, 8, 8)):
            m = ConvBlock(cin, cout)
            p = load_seeded(m, 100 + cin)
            x = torch.randn(2, cin, h, w, generator=g)
            y = m(x)
            save(f"op_convblock_{cin}_{cout}.npz", x=np32(x), y=np32(y),
                 y_pool=np32(F.max_pool2d(y, 2, 2)), **pack_params(p))

        for cin, cout, h, w in ((16, 16, 16, 16), (32, 32, 8, 12), (3, 3, 16, 16)):
            m = DenseBlock(cin, cout, 16, 4)
            p = load_seeded(m, 200 + cin)
            x = torch.randn(2, cin, h, w, generator=g)
            save(f"op_denseblock_{cin}.npz", x=np32(x), y=np32(m(x)), **pack_params(p))

        for c in (32, 64, 256):
            m = CBAM(c)
            p = load_seeded(m, 300 + c)
            x = torch.randn(2, c, 8, 12, generator=g).relu() + 0.1 * torch.randn(2, c, 8, 12, generator=g)
            cg = m.ChannelGate(x)
            save(f"op_cbam_{c}.npz", x=np32(x), y_channel=np32(cg), y=np32(m(x)), **pack_params(p))

        for cin, cout in ((32, 16), (64, 3)):
            m = nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=1, padding=1)
            p = load_seeded(m, 400 + cin)
            x = torch.randn(2, cin, 8, 12, generator=g)
            save(f"op_convtranspose_{cin}_{cout}.npz", x=np32(x), y=np32(m(x)), **pack_params(p))

        lo = torch.randn(2, 16, 6, 10, generator=g)
        skip = torch.randn(2, 16, 12, 20, generator=g)
        up = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=False)
        save("op_up2_add.npz", lo=np32(lo), skip=np32(skip), y=np32(torch.add(up, skip)))

    # ---- post-processing (utils/post_processing.py is torch-only: imported from the reference) -------------------
    from utils import post_processing as RPP  # noqa: E402  (reference)
    from utils.postprocessing_factory import apply_postprocessing as ref_apply  # noqa: E402  (reference)
    yp = torch.sigmoid(torch.randn(2, 3, 24, 40, generator=g) * 1.5)
    chain = {"enabled": True, "ops": [{"name": "enhance_contrast", "args": {"contrast_factor": 1.03}},
                                      {"name": "enhance_color", "args": {"saturation_factor": 1.55}}]}  # config/low_light.json:42-48
    chain4 = {"enabled": True, "ops": [{"name": "soft_denoise", "args": {"sigma": 0.3}}, {"name": "sharpen", "args": {"strength": 0.7}},
                                       {"name": "enhance_contrast", "args": {"contrast_factor": 1.2}},
                                       {"name": "enhance_color", "args": {"saturation_factor": 0.8}}]}
    out_chain = ref_apply(yp, chain)
    save("op_postproc.npz", y=np32(yp), contrast=np32(RPP.enhance_contrast(yp, 1.03)), color=np32(RPP.enhance_color(yp, 1.55)),
         sharpen=np32(RPP.sharpen(yp, 0.5)), denoise=np32(RPP.soft_denoise(yp, 0.2)), chain_lowlight=np32(out_chain),
         chain4=np32(ref_apply(yp, chain4)),
         chain_lowlight_u8=(out_chain.permute(0, 2, 3, 1).numpy() * 255).clip(0, 255).astype(np.uint8))

    # ---- one training step (batch-stat BN, dropout off) ----------------------------------------
    ref.load_state_dict(sd, strict=True)
    ref.train()
    for mod in ref.modules():
        if isinstance(mod, nn.Dropout):
            mod.eval()
    xt, tt = P.lowlight_batch(21, 2, 32, 32)
    y = ref(xt)
    loss = torch.sqrt((y - tt) ** 2 + 1e-6).mean()  # charbonnier, utils/loss_factory.py:160-167
    loss.backward()
    named = dict(ref.named_parameters())
    grads = {}
    for k in GRAD_KEYS:
        grads["g:" + k] = np32(named[k].grad)
    norms = {k: float(v.grad.double().norm()) for k, v in named.items()}
    stats = {}
    new_sd = ref.state_dict()
    for k in ("encoder.conv1.bn.running_mean", "encoder.conv1.bn.running_var",
              "decoder.cbam3.SpatialGate.spatial.bn.running_mean",
              "decoder.cbam3.SpatialGate.spatial.bn.running_var",
              "decoder.final_dense.layers.2.0.running_var"):
        stats["s:" + k] = np32(new_sd[k])
    save("train_step_32.npz", x=np32(xt), t=np32(tt), y=np32(y), loss=np.float64(loss.item()),
         grad_norms=np.array(json.dumps(norms)), **grads, **stats)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "all":
        main()
    if what in ("all", "dectaps"):
        dec_taps()
    if what in ("all", "ddp"):
        ddp_fixture()
    if what in ("all", "configs"):
        config_digest()
