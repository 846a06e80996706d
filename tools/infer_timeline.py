#!/usr/bin/env python3
"""One inference step kernel by kernel from a rocprofv3 kernel trace (graph replay or eager), with the queue each kernel ran on and
how long each main-chain kernel took next to its serial time.

  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --launch graph --steps 20 --warmup 5 --no-cpu --no-extra
  infer_timeline.py <dir> [out.txt]

A step starts at conv_first_pool_kernel.  The median step (by span) of the last ten is listed: start offset, duration, queue, idle gap
in front of it ON ITS QUEUE, the kernels of other queues that overlapped it."""
import csv, glob, os, re, sys

root = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdie" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "conv_first" in r[2]]
steps = [(rows[b][0] - rows[a][0], a, b) for a, b in zip(marks[-11:-1], marks[-10:])]
steps.sort()
span, a, b = steps[len(steps) // 2]
step = rows[a:b]


def short(n):
    m = re.search(r"mdie::(\w+)", n) or re.search(r"_ZN4mdie\d+(\w+?)I", n)
    return m.group(1) if m else n[:30]


t0 = step[0][0]
queues = sorted({r[3] for r in step}, key=lambda q: -sum(1 for r in step if r[3] == q))
print(f"# median step: {span / 1e3:.1f} us first kernel to next step's first; {len(step)} kernels on {len(queues)} queues; main queue = {queues[0]}", file=out)
print(f"{'start':>8s} {'us':>7s} {'queue':>6s} {'gap':>6s}  kernel   | overlapping kernels of other queues", file=out)
last_end = {}
for s, e, n, q, st in step:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    ov = [short(r[2]) for r in step if r[3] != q and r[0] < e and r[1] > s]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {q:>6s} {gap:6.1f}  {short(n):28s} | {' '.join(ov[:6])}", file=out)
main = [r for r in step if r[3] == queues[0]]
busy = sum(e - s for s, e, *_ in main)
print(f"\n# main queue: kernel time {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us of {span / 1e3:.1f}", file=out)
