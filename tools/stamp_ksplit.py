#!/usr/bin/env python3
"""In-kernel s_memtime stamps of conv_ksplit_kernel (csrc/conv_ksplit.hip, EXP_KSTAMPS build only): where a workgroup's cycles go.
  cd multi-degradation-image-enhancement_amd/csrc && hipcc <CXXFLAGS> -DEXP_KSTAMPS -c conv_ksplit.hip -o /tmp/ks.o && hipcc -shared ... -o ../libmdie_hip_kstamps.so
  MDIE_LIB=.../libmdie_hip_kstamps.so python tools/stamp_ksplit.py [d2l0 d2l3 d3l0 d3l3]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
import mdie_amd.lib as L
SHAPES = {"d2l0": ([128], 64), "d2l3": ([128, 16, 16, 16], 64), "d3l0": ([256], 32), "d3l3": ([256, 16, 16, 16], 32)}
LABELS = ["start", "first loads issued", "pre-act constants in LDS", "chunk 0 staged (its data arrived)", "next loads issued", "chunk 0 MFMAs done",
          "loop done", "last chunk staged", "last MFMAs done", "partials in LDS", "barrier passed", "end"]
dt, td, B = L.BF16, torch.bfloat16, int(os.environ.get("B", 32))
for name in sys.argv[1:] or list(SHAPES):
    segc, H = SHAPES[name]
    segs = [torch.randn(B, H, H, c, device="cuda").to(td) for c in segc]
    cin = sum(segc)
    w = E.pack_conv_weight(torch.randn(16, cin, 3, 3) * 0.05, dt).cuda()
    s, t = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
    ps, pt = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    out = torch.empty(B, H, H, 16, device="cuda", dtype=td)
    ntile = B * ((H + 7) // 8) ** 2
    dbg = torch.zeros((ntile + 8) * 2 * 16, dtype=torch.int64, device="cuda")

    def run(stamp):
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, H, H, 3, len(segs)
        for i, sg in enumerate(segs):
            d.inp[i] = L.Seg(sg.data_ptr(), sg.shape[3], sg.stride(2))
        d.cin, d.cout = cin, 16
        d.pre_scale, d.pre_shift = ps.data_ptr(), pt.data_ptr()
        d.weight, d.post_scale, d.post_shift = w.data_ptr(), s.data_ptr(), t.data_ptr()
        d.act, d.pool = L.ACT_NONE, 0
        if stamp:
            d.residual, d.res_stride = dbg.data_ptr(), -12345
        d.out, d.out_stride = out.data_ptr(), 16
        L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "conv")
    for _ in range(3):
        run(False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run(False)
    e1.record()
    torch.cuda.synchronize()
    print(f"== {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (unstamped path of the stamp build)")
    run(True)
    torch.cuda.synchronize()
    st = dbg.view(-1, 2, 16).cpu()
    for wv, wname in ((0, "wave 0"), (1, "wave 3")):
        a = st[:, wv]
        a = a[a[:, 0] > 0]
        rel = (a[:, :12] - a[:, :1]).double()
        span = (a[:, 11].max() - a[:, 0].min()).item()
        print(f"  {wname}: {a.shape[0]} workgroups, kernel span {span} cycles, median lifetime {rel[:, 11].median().item():.0f}")
        for i, lab in enumerate(LABELS):
            col = rel[:, i][a[:, i] > 0]
            if col.numel():
                print(f"     {lab:36s} median {col.median().item():8.0f}  p10 {col.quantile(0.1).item():8.0f}  p90 {col.quantile(0.9).item():8.0f}")
        starts = (a[:, 0] - a[:, 0].min()).double()
        print(f"     workgroup start times: p10 {starts.quantile(0.1).item():.0f} median {starts.median().item():.0f} p90 {starts.quantile(0.9).item():.0f} max {starts.max().item():.0f}")
