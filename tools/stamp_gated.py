#!/usr/bin/env python3
"""In-kernel s_memtime stamps of conv_gated_kernel (csrc/conv_gated.hip, EXP_GSTAMPS build of conv_gated.hip + cbam.hip only).
  MDIE_LIB=.../libmdie_hip_gstamps.so python tools/stamp_gated.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
import mdie_amd.lib as L
dt, td, B, H, Cc = L.BF16, torch.bfloat16, 32, 128, 64
x = torch.randn(B, H, H, Cc, device="cuda").to(td)
mul = torch.randn(B, H, H, Cc, device="cuda").to(td)
out = torch.empty(B, H, H, 16, device="cuda", dtype=td)
w = E.pack_conv_weight(torch.randn(16, Cc, 3, 3) * 0.05, dt).cuda()
f32 = lambda *s: torch.randn(*s, device="cuda") * 0.1
w1, b1, w2, b2, w7, bn = f32(Cc // 16, Cc), f32(Cc // 16), f32(Cc, Cc // 16), f32(Cc), f32(98), torch.tensor([1.0, 0.0], device="cuda")
sc, sh = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
nws = L.lib.mdie_cbam_workspace_bytes(B, H, H, Cc)
ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
ntile = B * (H // 16) ** 2
dbg = torch.zeros((ntile + 8) * 16, dtype=torch.int64, device="cuda")
f = L.CbamConvDesc()
d = f.cbam
d.dtype, d.B, d.H, d.W, d.C = dt, B, H, H, Cc
d.x, d.x_stride = x.data_ptr(), Cc
d.w1, d.b1, d.w2, d.b2, d.w7, d.bn = [t.data_ptr() for t in (w1, b1, w2, b2, w7, bn)]
d.mul, d.mul_stride = mul.data_ptr(), Cc
d.out, d.out_stride = dbg.data_ptr(), Cc
d.workspace, d.workspace_bytes = ws.data_ptr(), nws
f.weight, f.post_scale, f.post_shift, f.act = w.data_ptr(), sc.data_ptr(), sh.data_ptr(), L.ACT_RELU
f.out, f.out_stride = out.data_ptr(), 16
for _ in range(3):
    L.check(L.lib.mdie_cbam_conv_fwd(C.byref(f), None), "cbam_conv")
torch.cuda.synchronize()
dbg.zero_()
L.check(L.lib.mdie_cbam_conv_fwd(C.byref(f), None), "cbam_conv")
torch.cuda.synchronize()
st = dbg.view(-1, 16).cpu()
st = st[st[:, 0] > 0]
LABELS = {0: "start", 1: "chunk 0 loads issued", 2: "prologue in LDS (barrier)", 3: "spatial gate of the patch done (barrier)", 4: "chunk 0 staged", 5: "barrier",
          6: "chunk 0 MFMAs done", 8: "chunk 1 staged", 9: "barrier", 10: "chunk 1 MFMAs done", 12: "end"}
rel = (st - st[:, :1]).double()
print(f"{st.shape[0]} workgroups; median lifetime {rel[:, 12].median().item():.0f} cycles")
for i, lab in LABELS.items():
    col = rel[:, i]
    print(f"   {lab:42s} median {col.median().item():8.0f}  p10 {col.quantile(0.1).item():8.0f}  p90 {col.quantile(0.9).item():8.0f}")
