#!/usr/bin/env python3
"""In-kernel stamps of up_dense0_kernel (diagnostic build with -DEXP_UDSTAMPS, loaded through MDIE_LIB): the phases of one tile, wave 0 of
every workgroup.   build: tools/variant_lib.sh udstamps updense0.hip -DEXP_UDSTAMPS ;  run: MDIE_LIB=.../libmdie_hip_udstamps.so python tools/stamp_updense0.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from models.cdan import CDAN
from mdie_amd import synthetic as P
import mdie_amd.lib as L

B, S = 32, 256
net = CDAN(precision="bf16")
net.load_state_dict(P.make_state_dict(42), strict=True)
net = net.eval().cuda()
x, _ = P.lowlight_batch(1, B, S, S)
x = x.cuda()
eng = net._engine(x.device)
y = torch.empty_like(x)
dbg = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
for _ in range(5):
    eng.forward(x, out=y)
torch.cuda.synchronize()
L.lib.mdie_exp_set_ud_dbg(C.c_void_p(dbg.data_ptr()))
eng.forward(x, out=y)
torch.cuda.synchronize()
L.lib.mdie_exp_set_ud_dbg(C.c_void_p(None))
st = dbg.view(-1, 8).cpu().double().numpy()
st = st[st[:, 0] > 0]
names = ["weights/consts, lo stage load, addresses, x loads issued", "barrier 1 (lo staged)", "taps from LDS, interpolation, patch + base + trpatch writes", "barrier 2", "gather + MFMA + g0 / partial stores"]
print(f"up_dense0 bf16 B={B} {S}x{S}: {len(st)} workgroups; cycles of wave 0, median [p10 .. p90]")
for i, n in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {n:62s} {np.median(d):8.0f} [{np.percentile(d, 10):6.0f} .. {np.percentile(d, 90):6.0f}]")
tot = st[:, 5] - st[:, 0]
print(f"  {'workgroup lifetime':62s} {np.median(tot):8.0f} [{np.percentile(tot, 10):6.0f} .. {np.percentile(tot, 90):6.0f}]")
r = st[:, 6]
print(f"  first end .. last end: {(r.max() - r.min()) * 0.01:.1f} us")
