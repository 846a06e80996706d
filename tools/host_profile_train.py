#!/usr/bin/env python3
"""cProfile of the host side of eager training steps (where the 8 ms of enqueue time go).  python tools/host_profile_train.py [S]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.cdan import CDAN
from mdie_amd import host as H
from mdie_amd import synthetic as P

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x, t = P.lowlight_batch(100, 8, S, S)
x, t = x.cuda(), t.cuda()
losses = H.build_losses({"enabled": True, "terms": [{"name": "charbonnier", "weight": 1.0}, {"name": "ssim", "weight": 0.5}]})
net = CDAN(precision="bf16").cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    total, _ = losses(net(x), t)
    total.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
