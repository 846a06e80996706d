#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the conv kernel (EXP_STAMPS build only): per-phase cycles of a workgroup."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
import mdie_amd.lib as L
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(os.path.dirname(__file__), "bench_conv.py"))
SHAPES = {
    "conv2": ([64], 128, 128, 3, True, False), "conv4": ([256], 512, 32, 3, False, False),
    "d1l3": ([64, 16, 16, 16], 16, 128, 3, False, True), "fl3": ([16, 16, 16, 16], 16, 256, 3, False, True),
    "ftr": ([16, 16, 16, 16, 16], 16, 256, 1, False, True),
    "fl0": ([8], 16, 256, 3, False, True), "fl1": ([8, 16], 16, 256, 3, False, True),
    "d1l0": ([64], 16, 128, 3, False, True), "d2l0": ([128], 16, 64, 3, False, True), "d2l3": ([128, 16, 16, 16], 16, 64, 3, False, True),
    "d3l0": ([256], 16, 32, 3, False, True), "d3l3": ([256, 16, 16, 16], 16, 32, 3, False, True),
}
dt = L.BF16
td = torch.bfloat16
B = 32
LABELS = ["start", "setup done", "loads0 issued", "c0 pre-store", "c0 stored", "c0 next-loads issued", "c0 mfma done",
          "c1 pre-store", "c1 stored", "c1 next-loads issued", "c1 mfma done", "loop done", "epilogue done"]
for name in sys.argv[1:] or list(SHAPES):
    segc, cout, H, ks, pool, pre = SHAPES[name]
    segs = [torch.randn(B, H, H, c, device="cuda").to(td) for c in segc]
    cin = sum(segc)
    w = E.pack_conv_weight(torch.randn(cout, cin, ks, ks) * 0.05, dt).cuda()
    s, t = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    ps = torch.ones(cin, device="cuda") if pre else None
    pt = torch.zeros(cin, device="cuda") if pre else None
    Ho = H // 2 if pool else H
    out = torch.empty(B, Ho, Ho, cout, device="cuda", dtype=td)
    dbg = torch.zeros(65536 * 16, dtype=torch.int64, device="cuda")
    def run(stamp):
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize = dt, B, H, H, ks
        d.nseg = len(segs)
        for i, sg in enumerate(segs):
            d.inp[i] = L.Seg(sg.data_ptr(), sg.shape[3], sg.stride(2))
        d.cin, d.cout = cin, cout
        d.pre_scale, d.pre_shift = (ps.data_ptr() if pre else None), (pt.data_ptr() if pre else None)
        d.weight, d.post_scale, d.post_shift = w.data_ptr(), s.data_ptr(), t.data_ptr()
        d.act, d.pool = L.ACT_RELU, int(pool)
        if stamp:
            d.residual, d.res_stride = dbg.data_ptr(), -12345
        d.out, d.out_stride = out.data_ptr(), cout
        d.out_nchw3 = None
        L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "conv")
    for _ in range(3):
        run(False)
    torch.cuda.synchronize()
    run(True)
    torch.cuda.synchronize()
    st = dbg.view(-1, 16).cpu()
    st = st[st[:, 0] > 0]
    rel = (st[:, :13] - st[:, :1]).double()
    n = st.shape[0]
    span = (st[:, 12].max() - st[:, 0].min()).item()
    print(f"== {name}: {n} workgroups, kernel span {span} ticks (s_memtime @100MHz? see ratio), median per-WG lifetime {rel[:, 12].median().item():.0f}")
    for i, lab in enumerate(LABELS):
        col = rel[:, i]
        col = col[st[:, i] > 0]
        if col.numel():
            print(f"   {lab:24s} median {col.median().item():9.0f}  p10 {col.quantile(0.1).item():9.0f}  p90 {col.quantile(0.9).item():9.0f}")
    starts = (st[:, 0] - st[:, 0].min()).double()
    print(f"   WG start times: p10 {starts.quantile(0.1).item():.0f} median {starts.median().item():.0f} p90 {starts.quantile(0.9).item():.0f} max {starts.max().item():.0f}")
