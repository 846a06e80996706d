#!/usr/bin/env python3
"""In-kernel stamps of conv_first_kernel (diagnostic build with -DEXP_FSTAMPS, loaded through MDIE_LIB): where a
workgroup's lifetime goes -- input patch loads, barrier, the four pixel subtiles, draining the stores -- which CU it ran on
and when, and so how many workgroups a CU really overlapped.
  build:  cd multi-degradation-image-enhancement_amd/csrc && mkdir -p ../../build/exp &&
          hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DEXP_FSTAMPS -c conv.hip -o ../../build/exp/conv.o &&
          hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/exp/libmdie_FSTAMPS.so ../../build/exp/conv.o $(ls *.o | grep -v '^conv.o$')
  run:    MDIE_LIB=build/exp/libmdie_FSTAMPS.so python tools/stamp_first.py [B] [S]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mdie_amd.lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt, td = L.BF16, torch.bfloat16
x = torch.rand(B, 3, S, S, device="cuda")
wn = (np.random.default_rng(0).standard_normal((64, 3, 3, 3)) * 0.2).astype(np.float32)
wp = torch.zeros(L.lib.mdie_conv_first_weight_bytes(dt, 64), dtype=torch.uint8)
L.check(L.lib.mdie_pack_conv_first_weight(dt, wn.ctypes.data, 64, 64, wp.data_ptr()), "pack")
wp = wp.cuda()
sc, sh = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
out = torch.empty(B, S // 2, S // 2, 64, device="cuda", dtype=td)
grid = B * (S // 16) ** 2
dbg = torch.zeros(grid * 12, dtype=torch.int64, device="cuda")


def run(stamp):
    d = L.ConvFirstDesc()
    d.dtype, d.B, d.H, d.W, d.cout = dt, B, S, S, 64
    d.x, d.weight, d.post_scale, d.post_shift = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr()
    d.act, d.pool = L.ACT_RELU, 1
    d.out, d.out_stride = out.data_ptr(), 64
    L.lib.mdie_exp_set_dbg(C.c_void_p(dbg.data_ptr() if stamp else None))
    L.check(L.lib.mdie_conv_first_fwd(C.byref(d), None), "conv_first")


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
st = dbg.view(grid, 12).cpu().double().numpy()
bidx = np.nonzero(st[:, 0] > 0)[0]
st = st[st[:, 0] > 0]          # the kernel is persistent: fewer workgroups than tiles
tiles, grid = grid, len(st)
real0, real1, hw, ntile = st[:, 10], st[:, 9], st[:, 11].astype(np.int64), st[:, 8]
life = st[:, 7] - st[:, 0]
clk = np.median(life / ((real1 - real0) * 10.0 + 1e-9))       # s_memrealtime ticks at 100 MHz
seg = st[:, 1:7] / ntile[:, None]                             # per-tile averages of each workgroup
names = ["wait for loads + patch write", "barrier", "issue next patch + subtile 0", "subtile 1", "subtile 2", "subtile 3"]
print(f"conv_first (pooled 16-bit kernel) bf16 B={B} {S}x{S}: {us:.1f} us/launch, {tiles} tiles on {grid} workgroups ({np.median(ntile):.0f} tiles each), "
      f"shader clock ~{clk:.2f} GHz")
print("  wave 0 of each workgroup, cycles per tile (mean over its tiles), median over workgroups [p10 .. p90]:")
for i, n in enumerate(names):
    print(f"    {n:30s} {np.median(seg[:, i]):8.0f} [{np.percentile(seg[:, i], 10):6.0f} .. {np.percentile(seg[:, i], 90):6.0f}]")
tot = seg.sum(axis=1)
print(f"    {'sum':30s} {np.median(tot):8.0f} [{np.percentile(tot, 10):6.0f} .. {np.percentile(tot, 90):6.0f}]")
print(f"    {'lifetime / tiles':30s} {np.median(life / ntile):8.0f}   (lifetime {np.median(life):.0f})")
span = (real1.max() - real0.min()) * 10.0   # ns
print(f"  first start .. last end: {span / 1e3:.1f} us; workgroup lifetimes {np.median((real1 - real0) * 0.01):.1f} us median, {np.max((real1 - real0) * 0.01):.1f} us max")
cu = ((hw >> 32) & 0xF) * 1024 + ((hw >> 13) & 0x7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)     # XCC, SE, SH, CU of HW_ID
ids, cnt = np.unique(cu, return_counts=True)
print(f"  placement: {len(ids)} distinct CUs used; workgroups per CU: " + ", ".join(f"{n} x{(cnt == n).sum()}" for n in sorted(set(cnt))))
lt = (real1 - real0) * 0.01
for n in sorted(set(cnt)):
    sel = np.isin(cu, ids[cnt == n])
    print(f"    CUs holding {n}: workgroup lifetime median {np.median(lt[sel]):.1f} us")
xcc = (hw >> 32) & 0xF
print("  lifetime median by XCC: " + " ".join(f"{x}:{np.median(lt[xcc == x]):.1f}" for x in sorted(set(xcc))))
q = (bidx >> 3) * 4 // max((bidx >> 3).max() + 1, 1)
print("  lifetime median by quarter of the XCD's run list: " + " ".join(f"{np.median(lt[q == k]):.1f}" for k in range(4)))
simd_wave = hw & 0x3F
print("  lifetime p10/p50/p90/max: " + " ".join(f"{np.percentile(lt, p):.1f}" for p in (10, 50, 90, 100)))
if os.environ.get("STAMP_VERBOSE"):
    j = bidx >> 3
    for c in ids[:6]:
        sel = np.nonzero((cu == c) & (xcc == 0))[0]
        if len(sel):
            print(f"    CU {c}: " + "  ".join(f"j={j[i]} simd/wave={simd_wave[i]:02x} {lt[i]:.1f}us" for i in sel))
