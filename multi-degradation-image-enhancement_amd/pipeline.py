"""Device-side pieces around the network: input feed, post-processing,
uint8 output, PSNR/SSIM.  Thin ctypes wrappers; the arithmetic is in csrc/post.hip."""
import ctypes as C

import torch

from . import lib as L
from .engine import _require_gpu, _stream_ptr

DEFAULT_ARGS = {"enhance_contrast": ("contrast_factor", 1.1), "enhance_color": ("saturation_factor", 1.1),
                "sharpen": ("strength", 0.5), "soft_denoise": ("sigma", 0.2)}


def feed_uint8(images_u8_hwc):
    """uint8 [B,H,W,3] (GPU) -> float32 NCHW in [0,1]."""
    _require_gpu(images_u8_hwc, "feed_uint8")
    x = images_u8_hwc.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
    L.check(L.lib.mdie_u8hwc_to_f32nchw(B, H, W, x.data_ptr(), out.data_ptr(), _stream_ptr(x.device)), "mdie_u8hwc_to_f32nchw")
    return out


def to_uint8_hwc(images):
    """float32 NCHW -> uint8 [B,H,W,3] exactly like (img*255).clip(0,255).astype(uint8)."""
    _require_gpu(images, "to_uint8_hwc")
    x = images.contiguous()
    B, _, H, W = x.shape
    out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=x.device)
    L.check(L.lib.mdie_f32nchw_to_u8hwc(B, H, W, x.data_ptr(), out.data_ptr(), _stream_ptr(x.device)), "mdie_f32nchw_to_u8hwc")
    return out


def _ops_array(pp_cfg):
    ops = []
    for op in (pp_cfg or {}).get("ops", []):
        name = op["name"]
        if name not in L.PP_KINDS:
            raise ValueError(f"Unknown post-processing op: {name}")
        key, default = DEFAULT_ARGS[name]
        ops.append(L.PpOp(L.PP_KINDS[name], float((op.get("args") or {}).get(key, default))))
    return (L.PpOp * max(len(ops), 1))(*ops), len(ops)


def apply_postprocessing(images, pp_cfg, want_uint8=False):
    """Same contract as utils.postprocessing_factory.apply_postprocessing (reference), on the GPU.
    pp_cfg: {"enabled": bool, "ops": [{"name": ..., "args": {...}}, ...]}"""
    if not pp_cfg or not pp_cfg.get("enabled", False):
        return (images, to_uint8_hwc(images)) if want_uint8 else images
    _require_gpu(images, "apply_postprocessing")
    x = images.contiguous()
    B, _, H, W = x.shape
    arr, n = _ops_array(pp_cfg)
    nws = L.lib.mdie_postprocess_workspace_bytes(B, H, W)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    out = torch.empty_like(x)
    u8 = torch.empty(B, H, W, 3, dtype=torch.uint8, device=x.device) if want_uint8 else None
    L.check(L.lib.mdie_postprocess(B, H, W, x.data_ptr(), arr, n, out.data_ptr(), u8.data_ptr() if u8 is not None else None,
                                   ws.data_ptr(), nws, _stream_ptr(x.device)), "mdie_postprocess")
    return (out, u8) if want_uint8 else out


def psnr_ssim(pred, target):
    """Batch PSNR and SSIM (torchmetrics defaults) -> device tensor [2]."""
    _require_gpu(pred, "psnr_ssim")
    p, t = pred.contiguous().float(), target.contiguous().float()
    B, _, H, W = p.shape
    nws = L.lib.mdie_metrics_workspace_bytes(B, H, W)
    ws = torch.empty(nws, dtype=torch.uint8, device=p.device)
    out = torch.empty(2, dtype=torch.float32, device=p.device)
    L.check(L.lib.mdie_psnr_ssim(B, H, W, p.data_ptr(), t.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, _stream_ptr(p.device)),
            "mdie_psnr_ssim")
    return out
