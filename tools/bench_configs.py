#!/usr/bin/env python3
"""Timing of the BASELINE.json configurations that are not the bench line (bench.py measures configs[1]):
  routed   configs[3]: 9 task weight sets, B=32 x 256x256 per GPU, images pre-labelled by a stub router
  large    configs[4]: config/pixelation_hard.json 1024x1024 fp16 (its stated dtype), batch 1 and 4
  python tools/bench_configs.py [routed|large] [bf16|fp16|fp32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdie_amd import engine as E
from mdie_amd import synthetic as P

what = sys.argv[1] if len(sys.argv) > 1 else "routed"
prec = sys.argv[2] if len(sys.argv) > 2 else ("fp16" if what == "large" else "bf16")


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if what == "routed":
    tasks = ["blur", "color_distortion", "high_light", "jpeg", "low_contrast", "low_light", "motion_blur", "noise", "pixelation"]
    eng = E.RoutedEngine("cuda", prec)
    for i, t in enumerate(tasks):
        eng.load_task(t, P.make_state_dict(100 + i))
    x, _ = P.lowlight_batch(1, 32, 256, 256)
    x = x.cuda()
    g = torch.Generator().manual_seed(0)
    # the classifier itself (seeded random parameters: no ImageNet / trained weights offline); its labels are not used for
    # the timing below (random weights put every image in one class), only its cost is reported
    from mdie_amd import router as R
    router = R.DegradationRouter("cuda", prec).load(P.make_state_dict(7, R.router_param_spec()))
    dtr = timed(lambda: router.forward(x))
    print(f"router[{prec}] ResNet18 + 2 heads, B=32 256x256: {dtr*1e3:.2f} ms/batch = {32/dtr:.0f} img/s")
    # the grouping changes with every batch: 16 label lists drawn like a router's output, cycled
    lists = [[tasks[i] for i in torch.randint(0, len(tasks), (32,), generator=g).tolist()] for _ in range(16)]
    k = [0]

    def step(e):
        e.forward(x, lists[k[0] % len(lists)])
        k[0] += 1
    dtc = timed(lambda: step(eng), n=48)                 # one launch chain, per-image weight-set lookup
    one = E.CdanEngine("cuda", prec).load(P.make_state_dict(100))
    dt1 = timed(lambda: one.forward(x))
    print(f"routed[{prec}] 9 tasks, B=32 256x256, a different grouping every batch: {dtc*1e3:.2f} ms/batch = {32/dtc:.0f} img/s as ONE launch chain (weight set looked up per image)"
          f"  (single weight set, eager: {dt1*1e3:.2f} ms = {32/dt1:.0f} img/s)")
else:
    eng = E.CdanEngine("cuda", prec).load(P.make_state_dict(42))
    for B in (1, 4):
        x, _ = P.lowlight_batch(2, B, 1024, 1024)
        x = x.cuda()
        dt = timed(lambda: eng.forward(x), n=10)
        gb = B * 16 * (209.6e6 if prec == "fp32" else 104.8e6) / 1e9
        print(f"large[{prec}] B={B} 1024x1024: {dt*1e3:.2f} ms = {B/dt:.1f} img/s, algorithmic {gb/dt/1e3:.2f} TB/s")
