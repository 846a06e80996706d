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
fused = os.environ.get("FUSED_TAIL", "0") == "1"
general = os.environ.get("GENERAL_TAIL", "0") == "1"    # decoder.final_dense as the general chain (no transition folding)
NAMES = ["enc.conv1+pool"] + [f"dense1.l{i}" for i in range(4)] + ["dense1.tr", "enc.conv2+pool"] + \
        [f"dense2.l{i}" for i in range(4)] + ["dense2.tr", "enc.conv3+pool"] + [f"dense3.l{i}" for i in range(4)] + \
        ["dense3.tr", "enc.conv4+pool-stats", "bott.gate", "bott.chanpool", "bott.spatial", "dec.conv1+skip+pool-stats",
         "cbam1.gate", "cbam1.chanpool", "cbam1.spatial*d3", "dec.conv2", "up2+skip1+pool",
         "cbam2.gate+chanpool", "cbam2.spatial*d2", "dec.conv3", "up3+skip0+pool",
         "cbam3.gate+chanpool", "cbam3.spatial*d1", "dec.conv4"]
NAMES += (["tail(fused)"] if fused else (["up4+x+final.l0"] + [f"final.l{i}" for i in range(1, 4)] + ["final.tr+sigmoid->nchw"] if general or prec == "fp32" else
           ["up4+x+final.l0+tr"] + [f"final.l{i}+tr" for i in range(1, 3)] + ["final.l3+tr+sigmoid->nchw"]))
net = CDAN(precision=prec)
net.load_state_dict(P.make_state_dict(42), strict=True)
net = net.eval().cuda()
x, _ = P.lowlight_batch(1, B, S, S)
x = x.cuda()
eng = net._engine(x.device)
for _ in range(3):
    eng.forward(x, fused_tail=fused, general_tail=general)
acc = None
reps = 5
for _ in range(reps):
    _, ex = eng.forward(x, profile=True, fused_tail=fused, general_tail=general)
    ms = [m for _, m in ex["launches"]]
    acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
kinds = [k for k, _ in ex["launches"]]
tot = 0
for i, (k, m) in enumerate(zip(kinds, acc)):
    us = m / reps * 1e3
    tot += us
    print(f"{i:3d} {NAMES[i] if i < len(NAMES) else '?':20s} {k:14s} {us:8.1f} us")
print(f"total {tot:.1f} us  ({B / tot * 1e6:.0f} img/s kernel-time bound)")
