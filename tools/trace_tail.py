#!/usr/bin/env python3
"""Aggregate the LAST part of a rocprofv3 --kernel-trace CSV (steady state, after warm-up / library auto-tuning).
  python tools/trace_tail.py <kernel_trace.csv> <fraction of the time window, e.g. 0.4> <steps inside it> [top N]"""
import csv, sys
from collections import defaultdict

path, frac, steps = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
rows = list(csv.DictReader(open(path)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
agg = defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) >= cut:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"window {(t1 - cut) / 1e6:.1f} ms, kernel time {tot / 1e6:.1f} ms, {sum(v[0] for v in agg.values())} launches; per step ({steps:g} steps): "
      f"{tot / steps / 1e6:.2f} ms, {sum(v[0] for v in agg.values()) / steps:.0f} launches")
for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{ns / steps / 1e3:9.0f} us/step {n / steps:7.1f} calls {ns / n / 1e3:9.1f} us  {name[:110]}")
