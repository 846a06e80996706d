#!/usr/bin/env python3
"""One training step kernel by kernel, in launch order, from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/bench_train.py bf16 8 512 charbonnier:1,ssim:0.5 eager
  train_timeline.py <dir> [out.txt]

A step starts at nchw3_to_nhwc16_kernel (the feed conversion: once per forward).  The LAST complete step of the trace is
listed: start offset from the step's first kernel, duration, grid, workgroup, short name; then the totals per kernel family
and the gap time (the stream idle between kernels: launch-bound stretches show up here)."""
import collections, csv, glob, os, re, sys

root = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size") or r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1), int(r.get("Workgroup_Size") or r["Workgroup_Size_X"])))
rows.sort()
marks = [i for i, r in enumerate(rows) if "nchw3_to_nhwc16" in r[2]]
if len(marks) < 3:
    sys.exit("fewer than three steps in the trace")
a, b = marks[-2], marks[-1]
step = rows[a:b]


def short(n):
    m = re.search(r"mdie::(\w+)<?([^(]*)", n)
    if m:
        return (m.group(1) + " " + re.sub(r"\s+", "", m.group(2))[:40]).strip()
    m = re.search(r"_ZN4mdie\d+(\w+?)I(.*?)EEv", n)
    if m:
        return m.group(1) + " " + m.group(2)[:40]
    m = re.search(r"(multi_tensor_apply_kernel|FillFunctor<\w+>|CatArrayBatchedCopy\w*|copyBuffer|CUDAFunctor\w*<[^>]*>|\w+Functor)", n)
    return "torch:" + (m.group(1) if m else n[:40])


t0 = step[0][0]
busy, fam = 0, collections.OrderedDict()
print(f"# {len(step)} kernels, {(rows[b][0] - t0) / 1e3:.1f} us from the step's first kernel to the next step's first", file=out)
print(f"{'start us':>9s} {'us':>8s} {'gap us':>7s} {'grid':>9s} {'wg':>5s}  kernel", file=out)
prev_end = t0
for s, e, n, g, wg in step:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {max(0, s - prev_end) / 1e3:7.1f} {g:9d} {wg:5d}  {short(n)}", file=out)
    busy += e - s
    k = short(n).split(" ")[0]
    f = fam.setdefault(k, [0, 0])
    f[0] += e - s
    f[1] += 1
    prev_end = max(prev_end, e)
print(f"\n# kernel time {busy / 1e3:.1f} us; wall {(rows[b][0] - t0) / 1e3:.1f} us", file=out)
for k, (ns, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:40s} {ns / 1e3:9.1f} us {c:4d}", file=out)
