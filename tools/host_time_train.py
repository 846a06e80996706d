#!/usr/bin/env python3
"""How long the HOST needs to enqueue one eager training step (forward + loss + backward + Adam), next to what the GPU needs for it:
  python tools/host_time_train.py [bf16|fp16] [B] [S] [--profile]
The enqueue time is measured with the GPU kept busy by a long dummy kernel queue in front (so no launch ever waits for the device), the
step time with a synchronize per step.  --profile: cProfile of 5 steps, top 25 by cumulative time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.cdan import CDAN
from mdie_amd import host as H
from mdie_amd import synthetic as P

args = [a for a in sys.argv[1:] if not a.startswith("--")]
prec = args[0] if len(args) > 0 else "bf16"
B = int(args[1]) if len(args) > 1 else 8
S = int(args[2]) if len(args) > 2 else 512
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
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
    torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / n * 1e3
# enqueue time: park the device behind a few hundred ms of queued work so that the host never waits on it
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
for _ in range(40):
    big.add_(1.0)
t0 = time.perf_counter()
for _ in range(n):
    step()
host_ms = (time.perf_counter() - t0) / n * 1e3
torch.cuda.synchronize()
print(f"train[{prec}] B={B} {S}x{S} eager: {gpu_ms:.2f} ms per step with a synchronize each; the host enqueues a step in {host_ms:.2f} ms")
if "--profile" in sys.argv:
    import cProfile, pstats
    for _ in range(40):
        big.add_(1.0)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
