#!/usr/bin/env python3
"""In-kernel stamps of conv_thin_kernel (diagnostic build with -DEXP_TSTAMPS, loaded through MDIE_LIB): where a tile's time
goes for wave 0 of each persistent workgroup, summed over its tiles.
  build:  cd multi-degradation-image-enhancement_amd/csrc && mkdir -p ../../build/exp &&
          hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DEXP_TSTAMPS -c conv_thin.hip -o ../../build/exp/conv_thin.o &&
          hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/exp/libmdie_TSTAMPS.so ../../build/exp/conv_thin.o $(ls *.o | grep -v '^conv_thin.o$')
  run:    MDIE_LIB=build/exp/libmdie_TSTAMPS.so python tools/stamp_thin.py [growth maps in the input: 1 2 3] [B] [S]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mdie_amd.engine as E
import mdie_amd.lib as L

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt, td = L.BF16, torch.bfloat16
segs = [8] + [16] * ng
if os.environ.get("SEGS"):     # e.g. SEGS=8,8,8,8,8 : the same 5 columns, each in a buffer of its own (contiguous 16-byte pixels)
    segs = [int(v) for v in os.environ["SEGS"].split(",")]
cin = sum(segs)
bufs = [torch.randn(B, S, S, c, device="cuda").to(td) for c in segs]
w = E.pack_conv_weight(torch.randn(16, cin, 3, 3) * 0.1, dt, cin_stored=cin).cuda()
sc, sh = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
ps, pt = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
out = torch.empty(B, S, S, 16, device="cuda", dtype=td)
dbg = torch.zeros(4096 * 12, dtype=torch.int64, device="cuda")


def run(stamp):
    if hasattr(L.lib, "mdie_exp_set_thin_dbg"):
        L.lib.mdie_exp_set_thin_dbg(C.c_void_p(dbg.data_ptr() if stamp else None))
    d = L.ConvDesc()
    d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, S, S, 3, len(bufs)
    for i, (b, c) in enumerate(zip(bufs, segs)):
        d.inp[i] = L.Seg(b.data_ptr(), c, c)
    d.cin, d.cout = cin, 16
    d.pre_scale, d.pre_shift = ps.data_ptr(), pt.data_ptr()
    d.weight, d.post_scale, d.post_shift = w.data_ptr(), sc.data_ptr(), sh.data_ptr()
    d.act, d.pool = L.ACT_NONE, 0
    d.out, d.out_stride = out.data_ptr(), 16
    L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "conv")


for _ in range(20):
    run(False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run(False)
e1.record()
run(True)
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
st = dbg.view(-1, 12).cpu().double().numpy()
st = st[st[:, 6] > 0]
nt = st[:, 6]
clk = np.median(st[:, 7] / ((st[:, 9] - st[:, 8]) * 10.0 + 1e-9))
byts = B * S * S * (cin + 16) * 2
print(f"conv_thin bf16 B={B} {S}x{S}, {cin} stored input channels ({(cin + 7) // 8} columns): {us:.1f} us/launch = {byts / us / 1e3:.0f} GB/s of the {byts / 1e6:.0f} MB it must move; "
      f"{len(st)} workgroups x {np.median(nt):.0f} tiles, shader clock ~{clk:.2f} GHz")
names = ["wait for loads, pre-activation, LDS write", "barrier (planes written)", "issue next tile's loads", "MFMA phase", "epilogue + stores", "barrier (planes read)"]
seg = st[:, :6] / nt[:, None]
print("  wave 0 of each workgroup, cycles per tile, median over workgroups [p10 .. p90]:")
for i, n in enumerate(names):
    print(f"    {n:42s} {np.median(seg[:, i]):8.0f} [{np.percentile(seg[:, i], 10):6.0f} .. {np.percentile(seg[:, i], 90):6.0f}]")
print(f"    {'sum':42s} {np.median(seg.sum(1)):8.0f};  lifetime / tiles {np.median(st[:, 7] / nt):.0f}")
lt = (st[:, 9] - st[:, 8]) * 0.01
print("  workgroup lifetime p10/p50/p90/max: " + " ".join(f"{np.percentile(lt, p):.1f}" for p in (10, 50, 90, 100)) + " us")
