#!/usr/bin/env python3
"""Host time of one eager training step: how long the Python / autograd / launch side takes to ENQUEUE a step when the GPU is idle at its
start (synchronise, then time step() without waiting for the device) next to the step's time with the device in the loop.
  python tools/host_time_train.py [bf16|fp16] [B] [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.cdan import CDAN
from mdie_amd import host as H
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
x, t = P.lowlight_batch(100, B, S, S)
x, t = x.cuda(), t.cuda()
losses = H.build_losses({"enabled": True, "terms": [{"name": "charbonnier", "weight": 1.0}, {"name": "ssim", "weight": 0.5}]})
torch.manual_seed(42)
net = CDAN(precision=prec).cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    total, _ = losses(net(x), t)
    total.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
enq, full = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0)
    full.append(t2 - t0)
enq.sort(); full.sort()
print(f"host_time[{prec}] B={B} {S}x{S}: enqueue {enq[5] * 1e3:.2f} ms (min {enq[0] * 1e3:.2f}), enqueue + drain {full[5] * 1e3:.2f} ms")
