#!/usr/bin/env python3
"""Which Python lines of an eager training step launch ATen fill kernels (aten::zero_ / fill_ / zeros / ones ...): round 5's trace counted
40 FillFunctor launches per step that are not this library's kernels.  python tools/find_fills.py [S]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from models.cdan import CDAN
from mdie_amd import host as H
from mdie_amd import synthetic as P

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
names = ("aten::zero_", "aten::fill_", "aten::copy_", "aten::add_", "aten::mul_", "aten::div_", "aten::clone")
by = collections.Counter()
for e in prof.events():
    if e.name in names:
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 6:
            chain.append(q.name)
            q = q.cpu_parent
        st = [s for s in (e.stack or []) if "mdie" in s or "multi-degradation" in s or "host.py" in s or "train.py" in s or "tools/" in s]
        shape = tuple(e.input_shapes[0]) if e.input_shapes else ()
        by[(e.name, " <- ".join(chain) or (st[0] if st else "?"), shape)] += 1
for (n, where, shape), c in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:14s} {str(shape):22s} {where[:200]}")
