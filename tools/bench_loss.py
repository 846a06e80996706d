#!/usr/bin/env python3
"""mdie_loss_fwd_bwd alone (value + gradient of the configured terms in one call) at the training shape.
  python tools/bench_loss.py [B] [S] [terms, e.g. charbonnier:1,ssim:0.5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdie_amd import host as H
from mdie_amd import synthetic as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spec = sys.argv[3] if len(sys.argv) > 3 else "charbonnier:1,ssim:0.5"
losses = H.build_losses({"enabled": True, "terms": [{"name": s.split(":")[0], "weight": float(s.split(":")[1])} for s in spec.split(",")]})
x, t = P.lowlight_batch(5, B, S, S)
x, t = x.cuda().requires_grad_(True), t.cuda()
for _ in range(3):
    total, values = losses(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 50
for _ in range(n):
    total, values = losses(x, t)
e1.record()
torch.cuda.synchronize()
print(f"loss[{spec}] {B}x3x{S}x{S}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call (value + gradient), values {[round(v, 6) for v in values.tolist()]}")
