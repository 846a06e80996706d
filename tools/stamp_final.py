#!/usr/bin/env python3
"""In-kernel stamps of final_block_kernel (diagnostic build with -DEXP_FBSTAMPS, loaded through MDIE_LIB): where a tile's time goes, per
wave, summed over the tiles of its persistent workgroup.
  build:  tools/variant_lib.sh fbstamps final_block.hip -fno-slp-vectorize -DEXP_FBSTAMPS
  run:    MDIE_LIB=multi-degradation-image-enhancement_amd/libmdie_hip_fbstamps.so python tools/stamp_final.py [B] [S]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mdie_amd.engine as E
import mdie_amd.lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt, td = L.BF16, torch.bfloat16
g = torch.Generator().manual_seed(1)
lo = (torch.randn(B, S // 2, S // 2, 16, generator=g) * 0.5).cuda().to(td)
x = torch.rand(B, 3, S, S, generator=g).cuda()
ws = [torch.randn(16, 3 + 16 * l, 3, 3, generator=g) * (0.3 if l == 0 else 0.12) for l in range(4)]
w0 = torch.zeros(L.lib.mdie_conv_first_weight_bytes(dt, 16), dtype=torch.uint8)
L.check(L.lib.mdie_pack_conv_first_weight(dt, np.ascontiguousarray(ws[0].numpy()).ctypes.data, 16, 16, w0.data_ptr()), "pack")
w0 = w0.cuda()
wl = [E.pack_conv_weight(ws[l], dt, cin_stored=8 + 16 * l, split=3, gap=5).cuda() for l in (1, 2, 3)]
wtp = E.pack_conv_weight(torch.randn(3, 67, 1, 1, generator=g) * 0.2, dt, cout_stored=16, cin_stored=72, split=3, gap=5).cuda()
ps = [(torch.rand(8 + 16 * l, generator=g) + 0.5).cuda() for l in range(4)]
pb = [(torch.randn(8 + 16 * l, generator=g) * 0.3).cuda() for l in range(4)]
pst, pbt = (torch.rand(72, generator=g) + 0.5).cuda(), (torch.randn(72, generator=g) * 0.3).cuda()
ones, bias = torch.ones(16, device="cuda"), (torch.randn(16, generator=g) * 0.2).cuda()
y = torch.empty(B, 3, S, S, device="cuda")
dbg = torch.zeros(2048 * 4 * 16, dtype=torch.int64, device="cuda")


def run(stamp):
    if hasattr(L.lib, "mdie_exp_set_fb_dbg"):
        L.lib.mdie_exp_set_fb_dbg(C.c_void_p(dbg.data_ptr() if stamp else None))
    f = L.FinalDenseDesc()
    f.dtype, f.B, f.H, f.W = dt, B, S, S
    f.lo, f.lo_stride, f.x, f.w0 = lo.data_ptr(), 16, x.data_ptr(), w0.data_ptr()
    for l in range(3):
        f.w[l] = wl[l].data_ptr()
    for l in range(4):
        f.pre_scale[l], f.pre_shift[l], f.post_scale[l], f.post_shift[l] = ps[l].data_ptr(), pb[l].data_ptr(), ones.data_ptr(), bias.data_ptr()
    f.wt, f.tr_pre_scale, f.tr_pre_shift, f.tr_post_scale, f.tr_post_shift, f.y = wtp.data_ptr(), pst.data_ptr(), pbt.data_ptr(), ones.data_ptr(), bias.data_ptr(), y.data_ptr()
    L.check(L.lib.mdie_final_dense_fwd(C.byref(f), None), "mdie_final_dense_fwd")


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
st = dbg.view(-1, 4, 16).cpu().double().numpy()
st = st[st[:, 0, 12] > 0]
nt = st[:, 0, 12]
clk = np.median(st[:, 0, 13] / ((st[:, 0, 15] - st[:, 0, 14]) * 10.0 + 1e-9))
print(f"final_block bf16 B={B} {S}x{S}: {us:.1f} us/launch (stamps off); {len(st)} workgroups x {np.median(nt):.0f} tiles of 16x8, shader clock ~{clk:.2f} GHz")
names = ["P0 base patch: loads, interpolation, LDS writes", "B1 wait", "P1 layer 0 + epilogue (-> A1 A2 A3, tr)", "B2 wait", "P2 layer 1 band (27 MFMAs)", "B3 wait", "P3 layer 2 band (54 / 36 MFMAs)",
         "B4 wait", "P4 layer 3 band (36 MFMAs, weights from LDS)", "P2 epilogue + halo group", "P3 epilogue + halo group", "P4 transition, sigmoid, stores"]
order = [0, 1, 2, 3, 4, 9, 5, 6, 10, 7, 8, 11]
print("  cycles per tile, median over workgroups;   wave 0      wave 1      wave 2      wave 3")
tot = np.zeros(4)
for i in order:
    v = [np.median(st[:, w, i] / nt) for w in range(4)]
    tot += v
    print(f"    {names[i]:52s} " + " ".join(f"{q:10.0f}" for q in v))
print(f"    {'sum':52s} " + " ".join(f"{q:10.0f}" for q in tot) + f";  lifetime / tiles {np.median(st[:, 0, 13] / nt):.0f}")
lt = (st[:, 0, 15] - st[:, 0, 14]) * 0.01
print("  workgroup lifetime p10/p50/p90/max: " + " ".join(f"{np.percentile(lt, p):.1f}" for p in (10, 50, 90, 100)) + " us")
