#!/usr/bin/env python3
"""Experiment: k whole batches in flight -- k engines (own workspace, own side streams), k captured graphs, replayed
round-robin on k HIP streams, so that one batch's tail (the thin final_dense layers) overlaps the next batch's head.
  bench_inflight.py [k ...]   (B=32, 256x256, bf16; MDIE_INFLIGHT_B=16: the batch size of each engine -- k = 2 with B = 16 is ONE batch
                               of 32 cut in two halves that run side by side; MDIE_INFLIGHT_EAGER=1: eager launches instead of graphs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
from mdie_amd import synthetic as P

B, S = int(os.environ.get("MDIE_INFLIGHT_B", 32)), 256
EAGER = os.environ.get("MDIE_INFLIGHT_EAGER") == "1"
sd = P.make_state_dict(42)
dev = torch.device("cuda", 0)
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
for k in ks:
    engs = [E.CdanEngine(dev, "bf16").load(sd) for _ in range(k)]
    xs = [P.lowlight_batch(1000 + i, B, S, S)[0].to(dev) for i in range(k)]
    ys = [torch.empty_like(x) for x in xs]
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    graphs = []
    with torch.no_grad():
        for i in range(k):
            engs[i].forward(xs[i], out=ys[i])
        torch.cuda.synchronize()
        for i in range(k):
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                engs[i].forward(xs[i], out=ys[i])
            torch.cuda.current_stream(dev).wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                engs[i].forward(xs[i], out=ys[i])
            graphs.append(g)
        def run(n):
            for it in range(n):
                with torch.cuda.stream(streams[it % k]):
                    if EAGER:
                        engs[it % k].forward(xs[it % k], out=ys[it % k])
                    else:
                        graphs[it % k].replay()
        run(10); torch.cuda.synchronize()
        n = 60
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"in flight={k} x B={B} ({'eager' if EAGER else 'graphs'}): {dt*1e3:.4f} ms per batch of {B}  {B/dt:.0f} img/s")
