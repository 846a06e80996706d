"""CPU restatement of the PSNR / SSIM the reference reports (utils/metrics_factory.py:74-94).

TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED at the torchmetrics boundary: the reference calls torchmetrics'
PeakSignalNoiseRatio() / StructuralSimilarityIndexMeasure() with default arguments; torchmetrics
is not installed in the build image (unpinned in requirements.txt:9) and the reference has no test
that pins a value, so this file restates torchmetrics' published defaults and is
cross-checked against an independent float64 scipy.ndimage implementation in tests/.
"""
import math

import torch
import torch.nn.functional as F


def psnr(pred, target):
    """10 log10(R^2 / MSE); data_range=None: R = max(target, 0) - min(target, 0) (state starts at 0)."""
    mse = ((pred.double() - target.double()) ** 2).mean()
    r = max(float(target.max()), 0.0) - min(float(target.min()), 0.0)
    return float(10.0 * torch.log10(torch.tensor(r * r, dtype=torch.float64) / mse))


def gaussian_1d(size=11, sigma=1.5, dtype=torch.float64):
    d = torch.arange((1 - size) / 2, (1 + size) / 2, 1, dtype=dtype)
    g = torch.exp(-((d / sigma) ** 2) / 2)
    return g / g.sum()


def ssim(pred, target, size=11, sigma=1.5, k1=0.01, k2=0.03):
    """Gaussian-window SSIM, windows fully inside the picture (reflect-pad, filter, crop = valid conv
    on the interior), data_range = max(range(pred), range(target)); mean over all pixels and images."""
    p, t = pred.double(), target.double()
    L = max(float(p.max() - p.min()), float(t.max() - t.min()))
    c1, c2 = (k1 * L) ** 2, (k2 * L) ** 2
    g = gaussian_1d(size, sigma)
    k = (g[:, None] * g[None, :]).reshape(1, 1, size, size).repeat(p.shape[1], 1, 1, 1)

    def filt(z):
        return F.conv2d(z, k, groups=z.shape[1])

    mu_p, mu_t = filt(p), filt(t)
    s_pp, s_tt, s_pt = filt(p * p) - mu_p ** 2, filt(t * t) - mu_t ** 2, filt(p * t) - mu_p * mu_t
    m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_pp + s_tt + c2))
    return float(m.reshape(m.shape[0], -1).mean(-1).mean())
