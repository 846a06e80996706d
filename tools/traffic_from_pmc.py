#!/usr/bin/env python3
"""HBM traffic of one bench step from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE).

  traffic_from_pmc.py <dir with *_counter_collection.csv (searched recursively)> <out.json>

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: both
counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read, so it is doubled.  One "step" = the dispatches from one first-layer kernel
(conv_first_kernel) up to the next one; the median step is reported."""
import csv, glob, json, os, sys, collections

root, out = sys.argv[1], sys.argv[2]
per_counter = {}
for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "mdie" in r["Kernel_Name"]]
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [by_disp[k] for k in sorted(by_disp)]
    starts = [i for i, d in enumerate(disp) if "conv_first" in d["name"]]     # the first launch of a step (conv_first_kernel / conv_first_pool_kernel)
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        if not disp or cname not in disp[0]:
            continue
        steps = []
        for a, b in zip(starts, starts[1:]):
            steps.append(sum(d.get(cname, 0.0) for d in disp[a:b]))
        if steps:
            steps.sort()
            per_counter[cname] = {"kib_per_step_median": steps[len(steps) // 2], "steps_seen": len(steps), "launches_per_step": starts[1] - starts[0]}
            # per-kernel split of the median-like (last full) step
            a, b = starts[-2], starts[-1]
            split = collections.defaultdict(float)
            for d in disp[a:b]:
                n = d["name"]
                key = "conv3x3" if ("conv_kernel" in n and "Li3E" in n) or "conv_first" in n or "conv_wide" in n or "conv_thin" in n or "up_dense0" in n else "conv1x1" if ("conv_kernel" in n or "conv1x1" in n) else \
                      "cbam" if "cbam" in n else "upsample_add" if "upsample" in n else "tail" if "tail" in n else "layout"
                if "conv_kernel<" in n:  # demangled template form
                    key = "conv"
                split[key] += d.get(cname, 0.0)
            per_counter[cname]["kib_by_kind"] = dict(split)
fetch = per_counter.get("FETCH_SIZE", {}).get("kib_per_step_median")
write = per_counter.get("WRITE_SIZE", {}).get("kib_per_step_median")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_sha16: bench.py reports this file only for the kernel sources it was measured on)
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --launch eager",
       "kernel_source_sha16": bench.source_sha16(),
       "fetch_size_kib": fetch, "write_size_kib": write,
       "hbm_bytes_per_step": (2.0 * fetch + write) * 1024 if fetch is not None and write is not None else None,
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128 B request)",
       "detail": per_counter}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("fetch_size_kib", "write_size_kib", "hbm_bytes_per_step")}))
