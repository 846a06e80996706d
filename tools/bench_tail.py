#!/usr/bin/env python3
"""Micro-benchmark of the fused tail kernel alone (B=32, 256x256)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, S = 32, 256
dt = E.dtype_id(prec)
sd = P.make_state_dict(42)
params = E.pack_tail(sd, dt, prefix="decoder.final_dense").cuda()
x = torch.rand(B, 3, S, S, device="cuda")
lo = torch.rand(B, S // 2, S // 2, 16, device="cuda").to(E.TORCH_DTYPE[dt])
for _ in range(3):
    y = E.tail_fwd(x, params, dtype=dt, lo=lo)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", 20))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = E.tail_fwd(x, params, dtype=dt, lo=lo)
e1.record()
torch.cuda.synchronize()
print(f"tail[{prec}] {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
