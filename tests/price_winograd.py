#!/usr/bin/env python3
"""Pricing of Winograd F(2x2, 3x3) for the four matrix-bound convolutions (enc.conv2-4, dec.conv1: models/cdan.py:59-61,103) --
the NUMERICS half of the question, emulated on the CPU (not a test: pytest does not collect this file; it lives under tests/
because it drives the oracle).  The engine's direct kernels round the activations and weights to the storage type and accumulate
in fp32; a Winograd kernel would round the TRANSFORMED operands (V = B^T d B of the rounded input tile, U = G g G^T of the fp32
weights) to the storage type before the MFMA, accumulate in fp32 and apply A^T . A in fp32.

  python tests/price_winograd.py            -> layer error of enc.conv4 (direct vs Winograd, vs the fp32 convolution) and the
                                               whole-network error vs the fp32 oracle with the four layers replaced

The bandwidth half is arithmetic (profiles/LEDGER.md, "Winograd"): per 32-channel K chunk a workgroup holding T tiles x N outputs
x 16 positions in registers (T N <= 4096 accumulators) must take in 1 KB x N of transformed weights + the raw patch for T N / 16
MFMAs -- 84 KB per 1024 matrix cycles at T = N = 64, against 77 KB per 4600 for the direct 32x16 x 64 stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import cdan_oracle as O
from oracle import params as P

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
_conv2d, _convT = F.conv2d, F.conv_transpose2d


def rnd(t, dt):
    return t.to(dt).float() if dt is not None else t


def winograd(x, w, b, dt):
    """3x3 / pad 1 / stride 1 convolution of x [B,C,H,W] (H, W even) by F(2x2,3x3) with the transformed operands rounded to dt"""
    Bn, C, H, W = x.shape
    xp = F.pad(rnd(x, dt), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [B,C,H/2,W/2,4,4]
    V = rnd(torch.einsum("ik,bcxykl,jl->bcxyij", BT, d, BT), dt)
    U = rnd(torch.einsum("ik,ockl,jl->ocij", G, w, G), dt)
    M = torch.einsum("ocij,bcxyij->boxyij", U, V)
    Y = torch.einsum("ki,boxyij,lj->boxykl", AT, M, AT)          # [B,O,H/2,W/2,2,2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], H, W)
    return y + b.view(1, -1, 1, 1) if b is not None else y


WIDE = {(128, 64), (256, 128), (512, 256)}          # (cout, cin) of enc.conv2-4; dec.conv1 is the transposed (512 -> 256)


def patched(dt, use_winograd):
    def conv2d(x, w, b=None, stride=1, padding=0, **kw):
        if w.dim() == 4 and w.shape[2] == 3 and use_winograd and (w.shape[0], w.shape[1]) in WIDE:
            return winograd(x, w, b, dt)
        return _conv2d(rnd(x, dt), rnd(w, dt), b, stride=stride, padding=padding, **kw)

    def convT(x, w, b=None, stride=1, padding=0, **kw):
        if use_winograd and tuple(w.shape[:2]) == (512, 256):
            return winograd(x, w.flip(2, 3).transpose(0, 1).contiguous(), b, dt)
        return _convT(rnd(x, dt), rnd(w, dt), b, stride=stride, padding=padding, **kw)
    return conv2d, convT


def main():
    torch.manual_seed(0)
    sd = P.make_state_dict(42)
    x, _ = P.lowlight_batch(1, 2, 128, 128)
    with torch.no_grad():
        ref = O.cdan_forward(sd, x)
        # ---- one layer: enc.conv4's weights on a post-ReLU-like input of its shape ----
        w, b = sd["encoder.conv4.conv.weight"], sd["encoder.conv4.conv.bias"]
        xin = torch.relu(torch.randn(2, 256, 32, 32)) * 0.7
        exact = _conv2d(xin, w, b, padding=1)
        print("enc.conv4 alone (256 -> 512 at 32x32), max |err| / max |exact fp32 conv|:")
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            direct = _conv2d(rnd(xin, dt), rnd(w, dt), b, padding=1)
            wino = winograd(xin, w, b, dt)
            e = lambda a: float((a - exact).abs().max() / exact.abs().max())
            print(f"  {name}: direct (operands rounded) {e(direct):.2e}   Winograd F(2x2,3x3) (transformed operands rounded) {e(wino):.2e}   ratio {e(wino) / e(direct):.2f}")
        # ---- the whole network with the four layers replaced ----
        print("whole network (2 x 3 x 128 x 128), max |y - fp32 oracle| / max |oracle|  (every convolution's operands rounded to the type, fp32 accumulation):")
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            out = {}
            for use in (False, True):
                O.F.conv2d, O.F.conv_transpose2d = patched(dt, use)
                try:
                    out[use] = O.cdan_forward(sd, x)
                finally:
                    O.F.conv2d, O.F.conv_transpose2d = _conv2d, _convT
            e = lambda a: float((a - ref).abs().max() / ref.abs().max())
            print(f"  {name}: direct everywhere {e(out[False]):.2e}   enc.conv2-4 + dec.conv1 by Winograd {e(out[True]):.2e}")


if __name__ == "__main__":
    main()
