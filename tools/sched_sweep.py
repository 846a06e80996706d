#!/usr/bin/env python3
"""Schedule exploration on ONE box, in ONE process: which of the wide layers run on conv_kernel (co-resident with other workgroups) instead
of conv_wide (a workgroup per CU holding 154 KB of LDS), and where the three encoder DenseBlock branches fork -- interleaved rounds of
eager steps per configuration, every configuration's output compared bit for bit with the default's.
  tools/variant_lib_multi.sh sched "conv.hip engine.hip" -DEXP_SCHED
  MDIE_LIB=.../libmdie_hip_sched.so python tools/sched_sweep.py [bf16|fp16] [rounds] [steps] < configs   (one per line: "<nowide labels or ->  <fork p1,p2,p3 or ->")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.cdan import CDAN
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
configs = [tuple(l.split()) for l in sys.stdin.read().splitlines() if l.strip() and not l.startswith("#")]
configs = [("-", "-")] + [c for c in configs if tuple(c) != ("-", "-")]
net = CDAN(precision=prec)
net.load_state_dict(P.make_state_dict(42), strict=True)
net = net.eval().cuda()
x, _ = P.lowlight_batch(1000, 32, 256, 256)
x = x.cuda()
eng = net._engine(x.device)
ys = [torch.empty_like(x) for _ in configs]


def setenv(c):
    for var, v in (("MDIE_EXP_NOWIDE", c[0]), ("MDIE_EXP_FORK", c[1]), ("MDIE_EXP_WIDE_WGS", c[2] if len(c) > 2 else "-"), ("MDIE_EXP_STREAM_WGS", c[3] if len(c) > 3 else "-"), ("MDIE_EXP_LATE_JOIN", c[4] if len(c) > 4 else "-")):
        if v == "-":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v


def timed(c, y, n):
    setenv(c)
    for _ in range(5):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


res = [[] for _ in configs]
with torch.no_grad():
    for r in range(rounds):
        for i, c in enumerate(configs):
            res[i].append(timed(c, ys[i], steps))
base = sorted(res[0])[len(res[0]) // 2]
print(f"# {prec} B=32 256x256 eager, {rounds} interleaved rounds of {steps} steps; default schedule: {base:.1f} us")
print(f"# {'conv_kernel instead of conv_wide':44s} {'fork d1,d2,d3':14s} {'step us':>8s} {'vs default':>10s}  same bits   rounds")
for c, t, y in zip(configs, res, ys):
    m = sorted(t)[len(t) // 2]
    print(f"  {(c[0] + (' wgs ' + c[2] if len(c) > 2 else '') + (' stream x' + c[3] if len(c) > 3 and c[3] != '-' else '') + (' late-join' if len(c) > 4 and c[4] != '-' else '')):44s} {c[1]:14s} {m:8.1f} {m - base:+10.1f}  {str(bool(torch.equal(y, ys[0]))):9s}   " + " ".join(f"{v:.1f}" for v in t))
