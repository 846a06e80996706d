"""CPU restatement of the reference's network-free loss terms (utils/loss_factory.py:90-103, 146-230).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): plain torch ops, differentiated by autograd, used by tests/ to
check csrc/loss.hip (value and gradient).  `utils.loss_factory` itself cannot be imported here (it imports
torchvision, absent from this image), so mse / l1 / charbonnier / gradient_l1 are restated from the file's text;
the ssim term wraps torchmetrics' StructuralSimilarityIndexMeasure() (absent, version unpinned): PARITY UNPINNED at
that boundary, restated from the published defaults exactly as oracle/metrics_oracle.py does.  The data range of
SSIM is treated as a constant in the gradient.
"""
import torch
import torch.nn.functional as F

from . import metrics_oracle as M


def mse(o, t):
    return torch.mean((o - t) ** 2)                      # loss_factory.py:146-151 (nn.MSELoss)


def l1(o, t):
    return torch.mean(torch.abs(o - t))                  # :153-158 (nn.L1Loss)


def charbonnier(o, t, eps=1e-3):
    d = o - t
    return torch.mean(torch.sqrt(d * d + eps * eps))     # :160-167


def ssim_loss(o, t):
    """1 - SSIM, :180-189.  Differentiable restatement of metrics_oracle.ssim."""
    rng = torch.maximum(o.max() - o.min(), t.max() - t.min()).detach()
    c1, c2 = (0.01 * rng) ** 2, (0.03 * rng) ** 2
    g = M.gaussian_1d(11, 1.5).to(o.dtype)
    k = (g[:, None] * g[None, :]).reshape(1, 1, 11, 11).repeat(o.shape[1], 1, 1, 1)
    f = lambda z: F.conv2d(z, k, groups=z.shape[1])      # valid windows == reflect-pad 5, filter, crop 5
    mo, mt = f(o), f(t)
    soo, stt, sot = f(o * o) - mo * mo, f(t * t) - mt * mt, f(o * t) - mo * mt
    m = ((2 * mo * mt + c1) * (2 * sot + c2)) / ((mo * mo + mt * mt + c1) * (soo + stt + c2))
    return 1.0 - m.reshape(m.shape[0], -1).mean(-1).mean()


def _sobel(x):
    """:90-103 for one-channel x (the only shape the reference's view() accepts) and, per channel, for more."""
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype)
    k = torch.stack((kx, kx.t()), 0).unsqueeze(1)        # [2,1,3,3]
    b, c, h, w = x.shape
    return F.conv2d(x.reshape(b * c, 1, h, w), k, padding=1).reshape(b, c, 2, h, w)


def gradient_l1(o, t, to_gray=False):
    if to_gray:                                          # :207-213
        lum = lambda z: 0.2989 * z[:, 0:1] + 0.5870 * z[:, 1:2] + 0.1140 * z[:, 2:3]
        o, t = lum(o), lum(t)
    return torch.mean(torch.abs(_sobel(o) - _sobel(t)))  # :224-228


TERMS = {"mse": lambda o, t, p: mse(o, t), "l1": lambda o, t, p: l1(o, t), "charbonnier": lambda o, t, p: charbonnier(o, t, p),
         "ssim": lambda o, t, p: ssim_loss(o, t), "gradient_l1": lambda o, t, p: gradient_l1(o, t, bool(p))}


def pipeline(o, t, terms):
    """terms: [(name, weight, param)] -> (total, [values])  (LossPipeline.__call__, :24-55)"""
    vals = [TERMS[n](o, t, p) for n, _, p in terms]
    return sum(w * v for (_, w, _), v in zip(terms, vals)), vals
