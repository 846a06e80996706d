#!/usr/bin/env python3
"""A/B in ONE process: the whole forward with decoder.final_dense's transition folded into its producers (default) against
the general chain (MDIE_FWD_GENERAL_TAIL), alternating rounds, eager launches and hipGraph replay.  python tools/bench_tail_ab.py [prec] [B] [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.cdan import CDAN
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
net = CDAN(precision=prec)
net.load_state_dict(P.make_state_dict(42), strict=True)
net = net.eval().cuda()
x = P.lowlight_batch(1, B, S, S)[0].cuda()
eng = net._engine(x.device)
ys = {g: torch.empty_like(x) for g in (False, True)}


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


graphs = {}
for g in (False, True):
    eng.forward(x, out=ys[g], general_tail=g)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.forward(x, out=ys[g], general_tail=g)
    torch.cuda.current_stream().wait_stream(side)
    graphs[g] = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graphs[g]):
        eng.forward(x, out=ys[g], general_tail=g)
print(f"max |folded - general| = {(ys[False] - ys[True]).abs().max().item():.3e}")
res = {(g, form): [] for g in (False, True) for form in ("eager", "graph")}
for r in range(5):
    for g in (False, True):
        res[(g, "eager")].append(timed(lambda: eng.forward(x, out=ys[g], general_tail=g)))
        res[(g, "graph")].append(timed(graphs[g].replay))
for (g, form), v in res.items():
    v = sorted(v)
    print(f"{'general chain' if g else 'folded       '} {form}: median {v[len(v) // 2]:.4f} ms, min {v[0]:.4f} ms  ({B / v[len(v) // 2] * 1e3:.0f} img/s)")
