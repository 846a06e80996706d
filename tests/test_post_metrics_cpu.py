"""CPU: the post-processing oracle against vectors from the reference's own utils/post_processing.py,
and the PSNR/SSIM restatement against an independent float64 scipy.ndimage implementation
(torchmetrics is unavailable offline: "parity unpinned" at that boundary, see oracle/metrics_oracle.py)."""
import os

import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import metrics_oracle as M
from oracle import post_oracle as PO


@pytest.fixture(scope="module")
def g(golden_dir):
    z = np.load(os.path.join(golden_dir, "op_postproc.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_single_ops_match_reference(g):
    y = g["y"]
    for key, out in (("contrast", PO.enhance_contrast(y, 1.03)), ("color", PO.enhance_color(y, 1.55)),
                     ("sharpen", PO.sharpen(y, 0.5)), ("denoise", PO.soft_denoise(y, 0.2))):
        assert (out - g[key]).abs().max().item() <= 1e-6, key


def test_chains_and_uint8_match_reference(g):
    low = {"enabled": True, "ops": [{"name": "enhance_contrast", "args": {"contrast_factor": 1.03}},
                                    {"name": "enhance_color", "args": {"saturation_factor": 1.55}}]}
    c4 = {"enabled": True, "ops": [{"name": "soft_denoise", "args": {"sigma": 0.3}}, {"name": "sharpen", "args": {"strength": 0.7}},
                                   {"name": "enhance_contrast", "args": {"contrast_factor": 1.2}},
                                   {"name": "enhance_color", "args": {"saturation_factor": 0.8}}]}
    out = PO.apply_postprocessing(g["y"], low)
    assert (out - g["chain_lowlight"]).abs().max().item() <= 1e-6
    assert (PO.apply_postprocessing(g["y"], c4) - g["chain4"]).abs().max().item() <= 2e-6
    assert torch.equal(PO.to_uint8_hwc(g["chain_lowlight"]), g["chain_lowlight_u8"])
    assert PO.apply_postprocessing(g["y"], {"enabled": False, "ops": low["ops"]}) is g["y"]


def _ssim_scipy(p, t):
    p, t = p.double().numpy(), t.double().numpy()
    L = max(p.max() - p.min(), t.max() - t.min())
    c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    d = np.arange(-5, 6, dtype=np.float64)
    w = np.exp(-0.5 * (d / 1.5) ** 2)
    w /= w.sum()

    def f(z):  # separable Gaussian over H and W, then keep windows fully inside the image
        z = ndimage.correlate1d(ndimage.correlate1d(z, w, axis=2, mode="reflect"), w, axis=3, mode="reflect")
        return z[:, :, 5:-5, 5:-5]

    mp, mt = f(p), f(t)
    spp, stt, spt = f(p * p) - mp * mp, f(t * t) - mt * mt, f(p * t) - mp * mt
    m = ((2 * mp * mt + c1) * (2 * spt + c2)) / ((mp * mp + mt * mt + c1) * (spp + stt + c2))
    return float(m.mean())


@pytest.mark.parametrize("shape", [(2, 3, 32, 48), (1, 3, 64, 64), (3, 3, 17, 23)])
def test_ssim_restatement_vs_scipy(shape):
    gen = torch.Generator().manual_seed(shape[2])
    t = torch.rand(shape, generator=gen)
    p = (t + 0.1 * torch.randn(shape, generator=gen)).clamp(0, 1)
    assert M.ssim(p, t) == pytest.approx(_ssim_scipy(p, t), abs=1e-9)
    assert M.ssim(t, t) == pytest.approx(1.0, abs=1e-12)


def test_psnr_restatement():
    t = torch.full((2, 3, 16, 16), 0.5)
    t[0, 0, 0, 0] = 0.9  # data range = max(target) - min(target, 0) = 0.9
    p = t + 0.1
    assert M.psnr(p, t) == pytest.approx(10 * np.log10(0.81 / 0.01), rel=1e-6)


# ---- loss oracle (utils/loss_factory.py:146-230 restated) -----------------------------------------------------------
def _pair(seed, b=2, h=24, w=20):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(b, 3, h, w, generator=g, dtype=torch.float64)
    o = (t + 0.1 * torch.randn(b, 3, h, w, generator=g, dtype=torch.float64)).clamp(0, 1)
    return o, t


def test_loss_oracle_ssim_term_is_one_minus_the_metric():
    from oracle import loss_oracle as LO
    o, t = _pair(3)
    assert LO.ssim_loss(o, t).item() == pytest.approx(1.0 - M.ssim(o, t), abs=1e-12)


def test_loss_oracle_known_answers():
    """hand-computable cases of each term"""
    from oracle import loss_oracle as LO
    o = torch.zeros(1, 3, 12, 12, dtype=torch.float64)
    t = torch.full((1, 3, 12, 12), 0.5, dtype=torch.float64)
    assert LO.mse(o, t).item() == pytest.approx(0.25)
    assert LO.l1(o, t).item() == pytest.approx(0.5)
    assert LO.charbonnier(o, t, 1e-3).item() == pytest.approx((0.25 + 1e-6) ** 0.5)
    # Sobel of a constant difference is zero inside and +-(3|4)*0.5 at the zero-padded border
    d = LO._sobel(o[:, :1] - t[:, :1])
    assert d[0, 0, :, 1:-1, 1:-1].abs().max().item() == 0.0
    assert d[0, 0, 0, 5, 0].item() == pytest.approx(-(1 + 2 + 1) * 0.5)     # dx at the left edge: right column only
    # a horizontal unit ramp has dx = 8 everywhere inside
    ramp = torch.arange(12, dtype=torch.float64).reshape(1, 1, 1, 12).expand(1, 1, 12, 12)
    assert torch.all(LO._sobel(ramp)[0, 0, 0, 1:-1, 1:-1] == 8.0)
    # luminance weights
    rgb = torch.zeros(1, 3, 12, 12, dtype=torch.float64)
    rgb[:, 1] = ramp[0]
    assert LO.gradient_l1(rgb, torch.zeros_like(rgb), to_gray=True).item() > 0


def test_loss_oracle_gradients_by_finite_differences():
    from oracle import loss_oracle as LO
    terms = [("charbonnier", 1.0, 1e-3), ("ssim", 0.5, 0.0), ("mse", 0.3, 0.0)]
    o, t = _pair(5, b=1, h=14, w=13)
    o = o.requires_grad_(True)
    total, _ = LO.pipeline(o, t, terms)
    total.backward()
    g = o.grad.clone()
    idx = [(0, 0, 7, 6), (0, 2, 0, 0), (0, 1, 13, 12), (0, 1, 5, 9)]
    for i in idx:
        e = torch.zeros_like(o)
        e[i] = 1e-6
        with torch.no_grad():
            # interior points only move the data range when they are the extreme: central difference is robust to that
            fp, _ = LO.pipeline(o + e, t, terms)
            fm, _ = LO.pipeline(o - e, t, terms)
        assert ((fp - fm) / 2e-6).item() == pytest.approx(g[i].item(), rel=2e-4, abs=1e-9)
