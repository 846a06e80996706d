#!/usr/bin/env python3
"""mdie_bn_bwd_apply_multi alone: GB/s per (pixels, segment channels, consuming layers).  python tools/bench_bn_multi.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.lib as L

dt, td = L.BF16, torch.bfloat16
dev = "cuda"
for name, N, Cs, c0 in (("final_dense seg", 8 * 512 * 512, 16, 16), ("final_dense x", 8 * 512 * 512, 16, 16), ("dense1 seg", 8 * 256 * 256, 16, 64), ("dense1 x", 8 * 256 * 256, 64, 64),
                         ("dense2 x", 8 * 128 * 128, 128, 128), ("dense3 x", 8 * 64 * 64, 256, 256)):
    for nl in ((1, 2, 4) if "seg" in name else (5,)):
        x = torch.randn(N, Cs, device=dev).to(td)
        g = torch.empty_like(x)
        mean, inv = torch.zeros(Cs, device=dev), torch.ones(Cs, device=dev)
        das = [torch.randn(1 if Cs == 16 else Cs // 16, N, 16, device=dev).to(td) for _ in range(nl)]
        consts = [torch.rand(4, 320, device=dev) for _ in range(nl)]
        m = L.BnBwdMultiDesc()
        m.dtype, m.N, m.C = dt, N, Cs
        m.x, m.x_stride, m.g, m.g_stride = x.data_ptr(), Cs, g.data_ptr(), Cs
        m.mean, m.invstd, m.nlayer = mean.data_ptr(), inv.data_ptr(), nl
        for j in range(nl):
            m.da[j], m.da_stride[j], m.da_plane[j] = das[j].data_ptr(), 16, N * 16
            m.scale[j], m.shift[j], m.coef[j], m.coef_stride[j] = consts[j][0].data_ptr(), consts[j][1].data_ptr(), consts[j][2].data_ptr(), 320
        def run():
            L.check(L.lib.mdie_bn_bwd_apply_multi(C.byref(m), None), "multi")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        byts = N * Cs * 2 * (nl + 2)
        print(f"{name:16s} N={N:8d} C={Cs:3d} layers={nl}: {us:7.1f} us  {byts / us / 1e6:6.2f} TB/s")
