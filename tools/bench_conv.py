#!/usr/bin/env python3
"""Micro-benchmark of single conv launches at the network's shapes (B=32, 256x256 input).
usage: bench_conv.py [bf16|fp32] [name ...]   names: first conv2 conv3 conv4 dec1 dec2 d1l0 d1l3 d1tr d2l3 d3l3 fl0 fl3 ftr"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
import mdie_amd.lib as L

# name: (cin segments, cout, H, ksize, pool, pre)
SHAPES = {
    "conv2": ([64], 128, 128, 3, True, False), "conv3": ([128], 256, 64, 3, True, False), "conv4": ([256], 512, 32, 3, False, False),
    "dec1": ([512], 256, 32, 3, False, False), "dec2": ([256], 128, 32, 3, False, False), "dec3": ([128], 64, 64, 3, False, False),
    "d1l0": ([64], 16, 128, 3, False, True), "d1l3": ([64, 16, 16, 16], 16, 128, 3, False, True),
    "d1tr": ([64, 16, 16, 16, 16], 64, 128, 1, False, True),
    "d2tr": ([128, 16, 16, 16, 16], 128, 64, 1, False, True), "d3tr": ([256, 16, 16, 16, 16], 256, 32, 1, False, True),
    "d2l3": ([128, 16, 16, 16], 16, 64, 3, False, True), "d3l3": ([256, 16, 16, 16], 16, 32, 3, False, True),
    "fl0": ([16], 16, 256, 3, False, True), "fl3": ([16, 16, 16, 16], 16, 256, 3, False, True),
    "ftr": ([16, 16, 16, 16, 16], 16, 256, 1, False, True),
}


REPS = 20


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    names = sys.argv[2:] or ["first", "conv2", "conv3", "conv4", "dec1", "dec2", "d1l0", "d1l3", "d1tr", "d2l3", "d3l3", "fl0", "fl3", "ftr"]
    dt = E.dtype_id(prec)
    td = E.TORCH_DTYPE[dt]
    B = 32
    reps = int(os.environ.get("REPS", 20))
    global REPS
    REPS = reps
    for name in names:
        if name == "first":
            x = torch.rand(B, 3, 256, 256, device="cuda")
            w = torch.randn(64, 3, 3, 3)
            s, t = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
            us = timeit(lambda: E.conv_first_fwd(x, w, s, t, dtype=dt, act=L.ACT_RELU, pool=True))
            flops = 2 * 27 * 64 * 256 * 256 * B
            byts = B * (3 * 256 * 256 * 4 + 64 * 128 * 128 * (2 if prec == "bf16" else 4))
        else:
            segc, cout, H, ks, pool, pre = SHAPES[name]
            segs = [torch.randn(B, H, H, c, device="cuda").to(td) for c in segc]
            cin = sum(segc)
            w = E.pack_conv_weight(torch.randn(cout, cin, ks, ks) * 0.05, dt).cuda()
            s, t = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
            ps = torch.ones(cin, device="cuda") if pre else None
            pt = torch.zeros(cin, device="cuda") if pre else None
            Ho = H // 2 if pool else H
            out = torch.empty(B, Ho, Ho, cout, device="cuda", dtype=td)
            us = timeit(lambda: E.conv_fwd(segs, w, s, t, dtype=dt, ksize=ks, cout=cout, act=L.ACT_RELU, pool=pool,
                                           pre_scale=ps, pre_shift=pt, out=out))
            flops = 2 * cin * cout * ks * ks * H * H * B
            esz = 2 if prec == "bf16" else 4
            byts = B * (cin * H * H + cout * Ho * Ho) * esz
        print(f"{name:6s} {us:8.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  {byts / us / 1e3:8.1f} GB/s")


if __name__ == "__main__":
    main()
