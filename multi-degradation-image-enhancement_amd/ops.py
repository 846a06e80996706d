"""Custom torch operators over the C ABI (north star: "Python host on PyTorch-ROCm registers custom torch ops through
a thin C-ABI").  `torch.ops.mdie.*` are the dispatcher-visible entry points `models.cdan.CDAN.forward` /
`models.cbam.CBAM.forward` go through in eval mode; only a CUDA(=HIP) kernel is registered, so a CPU tensor fails in
the dispatcher ("no kernel for CPU backend") -- there is no fallback.  Fake (meta) implementations make the ops
traceable by torch.export / torch.compile without running them.

  mdie::cdan_forward(x, params, workspace, dtype, aux, flags) -> y     CDAN.forward, eval (models/cdan.py:171-176)
  mdie::cbam_forward(x, w1, b1, w2, b2, w7, bn, dtype, channel_only) -> y   CBAM.forward, eval (models/cbam.py:91-95), NCHW fp32
  mdie::psnr_ssim(pred, target) -> float[2]                              utils/metrics_factory.py:76,87
"""
import ctypes as C

import torch

from . import lib as L

_LIB = torch.library.Library("mdie", "DEF")
_LIB.define("cdan_forward(Tensor x, Tensor params, Tensor(a!) workspace, int dtype, int aux, int flags) -> Tensor")
_LIB.define("cbam_forward(Tensor x, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor w7, Tensor bn, int dtype, bool channel_only) -> Tensor")
_LIB.define("psnr_ssim(Tensor pred, Tensor target) -> Tensor")


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _cdan_forward(x, params, workspace, dtype, aux, flags):
    if x.dim() != 4 or x.shape[1] != 3:
        raise L.MdieError(f"mdie::cdan_forward: expected input [B,3,H,W], got {tuple(x.shape)}")
    x = x.to(torch.float32).contiguous()
    B, _, H, W = x.shape
    y = torch.empty_like(x)
    d = L.CdanFwdDesc()
    d.dtype, d.B, d.H, d.W = dtype, B, H, W
    d.params, d.x, d.y = params.data_ptr(), x.data_ptr(), y.data_ptr()
    d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel()
    d.flags = flags
    d.aux = C.c_void_p(aux) if aux else None
    with torch.cuda.device(x.device):
        L.check(L.lib.mdie_cdan_forward(C.byref(d), _stream(x.device)), "mdie_cdan_forward")
    return y


def _cbam_forward(x, w1, b1, w2, b2, w7, bn, dtype, channel_only):
    from . import engine as E
    y = E.cbam_fwd(E.to_nhwc(x.float(), dtype), w1, b1, w2, b2, w7, bn, dtype=dtype, channel_only=channel_only)
    return E.to_nchw(y, dtype)


def _psnr_ssim(pred, target):
    from . import pipeline as PL
    return PL.psnr_ssim(pred, target)


_LIB.impl("cdan_forward", _cdan_forward, "CUDA")
_LIB.impl("cbam_forward", _cbam_forward, "CUDA")
_LIB.impl("psnr_ssim", _psnr_ssim, "CUDA")


@torch.library.register_fake("mdie::cdan_forward")
def _(x, params, workspace, dtype, aux, flags):
    return torch.empty_like(x, dtype=torch.float32)


@torch.library.register_fake("mdie::cbam_forward")
def _(x, w1, b1, w2, b2, w7, bn, dtype, channel_only):
    return torch.empty_like(x, dtype=torch.float32)


@torch.library.register_fake("mdie::psnr_ssim")
def _(pred, target):
    return pred.new_empty(2, dtype=torch.float32)
