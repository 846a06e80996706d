#!/usr/bin/env python3
"""In-kernel stamps of conv_wide_kernel (diagnostic build with -DEXP_STAMPS, loaded through MDIE_LIB): per stage of
wave 0 -- wait for the stage's own DMA pieces, barrier (+ the next item's DMA address setup), the 144-MFMA phase with the
next stage's DMA pieces issued inside it, what follows (epilogue on an item's last chunk) -- and the shader clock the kernel ran at (s_memtime over s_memrealtime).
  build:  cd multi-degradation-image-enhancement_amd/csrc && mkdir -p ../../build/exp && for f in conv conv_wide; do
            hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DEXP_STAMPS -c $f.hip -o ../../build/exp/$f.o; done &&
          hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/exp/libmdie_STAMPS.so ../../build/exp/conv.o ../../build/exp/conv_wide.o \
            $(ls *.o | grep -v '^conv.o$' | grep -v '^conv_wide.o$')
  run:    MDIE_LIB=build/exp/libmdie_STAMPS.so python tools/stamp_wide.py [conv2 conv3 conv4 dec1 dec2 dec3]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
import mdie_amd.lib as L

SHAPES = {"conv2": (64, 128, 128, True), "conv3": (128, 256, 64, True), "conv4": (256, 512, 32, False),
          "dec1": (512, 256, 32, False), "dec2": (256, 128, 32, False), "dec3": (128, 64, 64, False)}
dt, td, B = L.BF16, torch.bfloat16, 32
for name in sys.argv[1:] or list(SHAPES):
    cin, cout, H, pool = SHAPES[name]
    x = torch.randn(B, H, H, cin, device="cuda").to(td)
    w = E.pack_conv_weight(torch.randn(cout, cin, 3, 3) * 0.05, dt).cuda()
    s, t = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    Ho = H // 2 if pool else H
    out = torch.empty(B, Ho, Ho, cout, device="cuda", dtype=td)
    items = B * (H // 16) * (H // 32) * (cout // 64)
    grid = 8 * min(-(-items // 8), 32)
    dbg = torch.zeros(grid * 100 + 16, dtype=torch.int64, device="cuda")

    def run(stamp):
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, H, H, 3, 1
        d.inp[0] = L.Seg(x.data_ptr(), cin, cin)
        d.cin, d.cout = cin, cout
        d.weight, d.post_scale, d.post_shift = w.data_ptr(), s.data_ptr(), t.data_ptr()
        d.act, d.pool = L.ACT_RELU, int(pool)
        if stamp:
            d.residual, d.res_stride = dbg.data_ptr(), -12345
        d.out, d.out_stride = out.data_ptr(), cout
        L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "conv")

    for _ in range(20):      # let the clock settle under this load
        run(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run(False)
    e1.record()
    run(True)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    st = dbg[:grid * 96].view(grid, 24, 4).cpu().double()
    real = dbg[grid * 96:grid * 98].view(grid, 2).cpu().double()
    tend = dbg[grid * 98:grid * 99].cpu().double()
    nchunk = cin // 32
    nst = min(24, -(-items // grid) * nchunk)
    live = st[:, 0, 0] > 0
    st, real, tend = st[live], real[live], tend[live]
    cyc = tend - st[:, 0, 0]
    clk = (cyc / ((real[:, 1] - real[:, 0]) * 10.0)).median().item()       # s_memrealtime ticks at 100 MHz
    flops = 2.0 * cin * cout * 9 * H * H * B
    print(f"== {name}: {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s; {int(live.sum())} workgroups x {nst}+ stages; in-kernel clock {clk:.2f} GHz; "
          f"wave-0 lifetime median {cyc.median().item():.0f} cycles")
    print("   stage   wait for dma  barrier+setup    mfma+dma  to next stage (epilogue)")
    for k in range(nst):
        a = st[:, k]
        nxt = st[:, k + 1, 0] if k + 1 < nst else tend
        row = [(a[:, 1] - a[:, 0]), (a[:, 2] - a[:, 1]), (a[:, 3] - a[:, 2]), (nxt - a[:, 3])]
        print(f"   {k:5d} " + " ".join(f"{r.median().item():12.0f}" for r in row))
