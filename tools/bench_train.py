#!/usr/bin/env python3
"""Training-step timing (BASELINE configs[2] shape: blur-style loss, 512x512, batch 8 per GPU).
  python tools/bench_train.py [bf16|fp32] [B] [S] [loss terms, e.g. charbonnier:1,ssim:0.5]      (under torchrun: one rank per GPU, bucketed RCCL all-reduce)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.cdan import CDAN
from mdie_amd import host as H
from mdie_amd import train as T
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl")
torch.manual_seed(42)
net = CDAN(precision=prec).cuda().train()
x, t = P.lowlight_batch(100 + rank, B, S, S)
x, t = x.cuda(), t.cuda()
spec = sys.argv[4] if len(sys.argv) > 4 else "charbonnier:1,ssim:0.5"
losses = H.build_losses({"enabled": True, "terms": [{"name": s.split(":")[0], "weight": float(s.split(":")[1])} for s in spec.split(",")]})
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
buckets = T.GradBuckets(net.parameters()) if dist is not None else None

def step():
    opt.zero_grad(set_to_none=True)
    out = net(x)
    total, _ = losses(out, t)
    total.backward()
    if buckets is not None:
        buckets.finish()
    opt.step()
    return total

for _ in range(3):
    l = step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
if rank == 0:
    print(f"train[{prec}] B={B}x{world} {S}x{S} loss={spec}: {dt*1e3:.1f} ms/step, {B*world/dt:.1f} img/s, loss {l.item():.4f}")
