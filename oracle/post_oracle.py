"""CPU restatement of the reference's post-processing ops and uint8 conversion.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned: tests/golden/op_postproc.npz holds outputs of
the reference's own utils/post_processing.py (torch-only, importable in the build container).
"""
import torch
import torch.nn.functional as F


def enhance_contrast(x, contrast_factor=1.1):
    """utils/post_processing.py:5-15."""
    m = x.mean(dim=(2, 3), keepdim=True)
    return ((x - m) * contrast_factor + m).clamp(0.0, 1.0)


def enhance_color(x, saturation_factor=1.1):
    """utils/post_processing.py:18-30."""
    gray = (0.2989 * x[:, 0] + 0.5870 * x[:, 1] + 0.1140 * x[:, 2]).unsqueeze(1)
    return (gray + saturation_factor * (x - gray)).clamp(0.0, 1.0)


def _depthwise3(x, k):
    k = k.to(x.dtype).reshape(1, 1, 3, 3).repeat(x.shape[1], 1, 1, 1)
    return F.conv2d(x, k, padding=1, groups=x.shape[1])


def sharpen(x, strength=0.5):
    """utils/post_processing.py:33-54 (the identity MATRIX eye(3) is added to the scaled Laplacian-like kernel)."""
    k = torch.tensor([[0., -1., 0.], [-1., 5., -1.], [0., -1., 0.]]) * strength + torch.eye(3)
    return _depthwise3(x, k / k.sum()).clamp(0.0, 1.0)


def soft_denoise(x, sigma=0.2):
    """utils/post_processing.py:57-77."""
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]])
    return ((1 - sigma) * x + sigma * _depthwise3(x, k / k.sum())).clamp(0.0, 1.0)


OPS = {"enhance_contrast": enhance_contrast, "enhance_color": enhance_color, "sharpen": sharpen, "soft_denoise": soft_denoise}


def apply_postprocessing(x, cfg):
    """utils/postprocessing_factory.py:19-41."""
    if not cfg or not cfg.get("enabled", False):
        return x
    for op in cfg.get("ops", []):
        x = OPS[op["name"]](x, **(op.get("args") or {}))
    return x


def to_uint8_hwc(x):
    """models/model.py:80-84: CHW float -> HWC, (img*255).clip(0,255).astype(uint8) (truncation)."""
    a = (x.permute(0, 2, 3, 1).contiguous().numpy() * 255.0).clip(0, 255).astype("uint8")
    return torch.from_numpy(a)
