#!/usr/bin/env python3
"""Per-launch times of one CDAN forward (instrumented mode: a hipEvent pair around every launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from models.cdan import CDAN
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
general = os.environ.get("GENERAL_TAIL", "0") == "1"    # decoder.final_dense as the general chain (no transition folding)
net = CDAN(precision=prec)
net.load_state_dict(P.make_state_dict(42), strict=True)
net = net.eval().cuda()
x, _ = P.lowlight_batch(1, B, S, S)
x = x.cuda()
eng = net._engine(x.device)
for _ in range(3):
    eng.forward(x, general_tail=general)
acc = None
reps = 5
for _ in range(reps):
    _, ex = eng.forward(x, profile=True, general_tail=general)
    ms = [m for _, m in ex["launches"]]
    acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
kinds = [k for k, _ in ex["launches"]]
info = ex["launch_info"]          # (label, algorithmic bytes, FLOPs) per launch, from the engine's own launch list (include/mdie.h: mdie_launch_info)
tot = 0
print(f"# {prec} B={B} {S}x{S}: per launch, serial (instrumented mode: side branches in line), mean of {reps} passes; bytes / FLOPs = the launch's share of the SURVEY 8d model")
print(f"# {'#':>2s} {'layer':34s} {'kind':14s} {'us':>8s} {'alg MB':>8s} {'TB/s':>6s} {'of 8':>5s} {'TFLOP/s':>8s} {'of 2.5PF':>8s}")
for i, (k, m) in enumerate(zip(kinds, acc)):
    us = m / reps * 1e3
    tot += us
    label, by, fl = info[i]
    tbs = by / us / 1e6 if us else 0.0
    tfs = fl / us / 1e6 if us else 0.0
    peak = 157.3 if prec == "fp32" else 2500.0
    print(f"{i:4d} {label:34s} {k:14s} {us:8.1f} {by / 1e6:8.1f} {tbs:6.2f} {tbs / 8:5.2f} {tfs:8.1f} {tfs / peak:8.3f}")
print(f"total {tot:.1f} us  ({B / tot * 1e6:.0f} img/s kernel-time bound); model: {sum(b for _, b, _ in info) / 1e9:.4f} GB, {sum(f for _, _, f in info) / 1e9:.2f} GFLOP per step")
