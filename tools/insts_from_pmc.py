#!/usr/bin/env python3
"""Dynamic instruction counts per launch of one bench step from a rocprofv3 --pmc pass.

  insts_from_pmc.py <dir with *_counter_collection.csv> [out.txt]

Prints, for every dispatch of the last full step (conv_first_kernel .. next conv_first_kernel), the counters divided by
SQ_WAVES: instructions a wave executes in that launch.  Used to see which launches are issue-bound rather than HBM-bound."""
import csv, glob, os, sys, collections, re

root = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
by_disp = collections.OrderedDict()
for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdie" not in r["Kernel_Name"]:
            continue
        d = by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "?"), "wg": r.get("Workgroup_Size", "?")})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = [by_disp[k] for k in sorted(by_disp)]
starts = [i for i, d in enumerate(disp) if "conv_first_kernel" in d["name"]]
a, b = starts[-2], starts[-1]
names = sorted({k for d in disp[a:b] for k in d if k not in ("name", "grid", "wg")})
print(f"{'kernel':60s} {'grid':>9s} {'wg':>4s} " + " ".join(f"{n[-12:]:>12s}" for n in names), file=out)
for d in disp[a:b]:
    n = re.sub(r"^void mdie::|\(.*$", "", d["name"])[:60]
    w = d.get("SQ_WAVES", 0.0) or 1.0
    print(f"{n:60s} {d['grid']:>9s} {d['wg']:>4s} " + " ".join(f"{(d.get(k, 0.0) / (1.0 if k == 'SQ_WAVES' else w)):12.1f}" for k in names), file=out)
