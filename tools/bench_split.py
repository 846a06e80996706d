#!/usr/bin/env python3
"""ONE batch of B images as k sub-batches on k streams (one engine + workspace each), joined at the end of every step,
against the same batch as one forward: does splitting fill the bubbles of the dependent chain?  python tools/bench_split.py [prec] [B] [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdie_amd import engine as EG
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
sd = P.make_state_dict(42)
x = P.lowlight_batch(1, B, S, S)[0].to(dev)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


ref_eng = EG.CdanEngine(dev, prec).load(sd)
y_ref = torch.empty_like(x)
results = {}
for k in (1, 2, 4):
    engs = [EG.CdanEngine(dev, prec).load(sd) for _ in range(k)]
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    xs = [c.contiguous() for c in x.chunk(k)]
    y = torch.empty_like(x)
    ys = list(y.chunk(k))
    main = torch.cuda.current_stream(dev)
    evs = [torch.cuda.Event() for _ in range(k)]

    def step():
        fork = torch.cuda.Event()
        fork.record(main)
        for i in range(k):
            streams[i].wait_event(fork)
            with torch.cuda.stream(streams[i]):
                engs[i].forward(xs[i], out=ys[i])
                evs[i].record(streams[i])
        for i in range(k):
            main.wait_event(evs[i])
    step()
    torch.cuda.synchronize()
    if k == 1:
        y_ref.copy_(y)
    else:
        assert torch.equal(y, y_ref), "sub-batches must reproduce the whole batch bit for bit"
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        step()
    main.wait_stream(side)
    try:
        with torch.cuda.graph(g):
            step()
        tg = min(timed(g.replay) for _ in range(3))
    except Exception as e:
        tg = float("nan"); print("graph capture failed:", repr(e)[:200])
    te = min(timed(step) for _ in range(3))
    print(f"{k} sub-batch(es) of {B // k}: eager {te:.4f} ms ({B / te * 1e3:.0f} img/s), graph {tg:.4f} ms ({B / tg * 1e3:.0f} img/s)", flush=True)
