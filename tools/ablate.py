#!/usr/bin/env python3
"""What each stage COSTS THE STEP (as opposed to its serial time, tools/profile_layers.py): the eager forward of the bench workload
(B = 32, 256x256) timed with the stage's launches LEFT OUT -- one process, one box, the sets interleaved over several rounds.
Results are garbage by construction (a left-out stage leaves its output buffer as it was): only the step time is read.
  tools/ablate.sh build                                 here: engine.hip with -DEXP_ABLATE, linked with the shipped objects
  MDIE_LIB=.../libmdie_hip_ablate.so python tools/ablate.py [bf16|fp16] [rounds]      on the GPU box"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MDIE_LIB", os.path.join(ROOT, "multi-degradation-image-enhancement_amd", "libmdie_hip_ablate.so"))
import torch
from mdie_amd import engine as E
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = E.CdanEngine("cuda", prec).load(P.make_state_dict(42))
x, _ = P.lowlight_batch(1000, 32, 256, 256)
x = x.cuda()
y = torch.empty_like(x)

SETS = ["", "enc.conv1", "enc.conv2", "enc.conv3", "enc.conv4", "dense1", "dense2", "dense3", "dense1,dense2,dense3", "dense1.tr", "dense2.tr", "dense3.tr",
        "dense1.l3,dense1.tr", "bott", "dec.conv1", "cbam1", "dec.conv2", "up2", "cbam2", "dec.conv3", "up3", "cbam3", "dec.conv4", "final.l0", "final.l1", "final.l2", "final.l3",
        "final.", "bott,cbam1,cbam2,cbam3", "enc.conv4,bott,dec.conv1,cbam1,dec.conv2", "enc.conv4,bott,dec.conv1,cbam1,dec.conv2,dense"]


def timed(n=200, warm=10):
    for _ in range(warm):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


res = {s: [] for s in SETS}
for r in range(rounds):
    for s in SETS:
        os.environ["MDIE_ABLATE"] = s
        res[s].append(timed())
os.environ["MDIE_ABLATE"] = ""
base = sorted(res[""])[len(res[""]) // 2]
print(f"# {prec} B=32 256x256 eager, {rounds} interleaved rounds of 200 steps; step with nothing left out: {base:.1f} us")
print(f"# {'left out':58s} {'step us':>8s} {'costs the step':>15s}   rounds")
for s in SETS:
    v = sorted(res[s])
    m = v[len(v) // 2]
    print(f"  {s or '(nothing)':58s} {m:8.1f} {base - m:15.1f}   {' '.join(f'{q:.1f}' for q in res[s])}")
