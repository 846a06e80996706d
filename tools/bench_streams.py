#!/usr/bin/env python3
"""Experiment: split the batch over k HIP streams (separate workspaces), one captured graph.
Run with MDIE_SIDE_STREAMS=0: capturing a fork inside a fork (sub-batch stream -> the engine's own side streams) crashes
hipStreamEndCapture on ROCm 7.2.  Measured (B=32, 256x256, bf16): 1 stream 1.52 ms, 2 streams 1.48 ms, 4 streams 1.94 ms --
the engine's own DenseBlock side streams (1.44 ms) already take what concurrency there is."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdie_amd.engine as E
from mdie_amd import synthetic as P

B, S = 32, 256
sd = P.make_state_dict(42)
x, _ = P.lowlight_batch(1000, B, S, S)
x = x.cuda()
dev = x.device
for k in (1, 2, 4, 8):
    engs = [E.CdanEngine(dev, "bf16").load(sd) for _ in range(k)]
    xs = list(x.chunk(k))
    ys = [torch.empty_like(c) for c in xs]
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    def step():
        cur = torch.cuda.current_stream(dev)
        for i in range(k):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                engs[i].forward(xs[i], out=ys[i])
        for i in range(k):
            cur.wait_stream(streams[i])
    step(); torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"streams={k}: {dt*1e3:.3f} ms/step  {B/dt:.0f} img/s")
