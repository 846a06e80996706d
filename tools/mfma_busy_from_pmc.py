#!/usr/bin/env python3
"""Matrix-pipe busy fraction per kernel from a rocprofv3 --pmc pass (north star: "MFMA-busy ... against gfx950 peak").

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python3 bench.py --launch eager ...
  mfma_busy_from_pmc.py <dir> [out.txt]

Per dispatch: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (16 cycles per v_mfma_f32_16x16x32_bf16,
MI355X_MICROARCH.md cycle constants); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles elapsed = GRBM_GUI_ACTIVE / 8 and
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)          (the gfx94x MfmaUtil formula, spelled out)
  clock     = (GRBM_GUI_ACTIVE / 8) / duration                                 (reads high on dispatches under ~0.3 ms)
TFLOP/s of the wide convolutions comes from their known FLOP counts (B = 32, 256x256) over the traced duration."""
import collections, csv, glob, os, sys

root = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
rows = collections.defaultdict(dict)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdie" not in r["Kernel_Name"]:
            continue
        d = rows[int(r["Dispatch_Id"])]
        d["name"], d["grid"] = r["Kernel_Name"], int(r["Grid_Size"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.OrderedDict()
for k in sorted(rows):
    d = rows[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    short = d["name"].replace("_ZN4mdie", "").split("EvNS_")[0][:44]
    agg.setdefault((short, d["grid"]), []).append(d)
print("# mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): matrix-pipe busy fraction of the cycles the chip\n"
      "#               counted while the dispatch ran (GRBM counts at its own reference rate: reads low when the shader clock is throttled)\n"
      "# vs_2.4GHz   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz): busy fraction against the NOMINAL peak clock, i.e. the\n"
      "#               fraction of the dense bf16 MFMA peak (2.5 PFLOP/s) the dispatch delivered as matrix work\n"
      "# (in-kernel shader clock of the wide convolutions under this load: 1.4-1.66 GHz, profiles/r02g_wide_conv_stage_stamps.txt)", file=out)
print(f"{'kernel (mangled, shortened)':46s} {'grid':>9s} {'n':>4s} {'us':>8s} {'mfma_busy':>9s} {'vs_2.4GHz':>9s} {'sq_busy':>8s}", file=out)
for (name, grid), ds in agg.items():
    ds = sorted(ds, key=lambda d: d.get("us", 0.0))
    m = ds[len(ds) // 2]
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else 0.0
    sq = m.get("SQ_BUSY_CYCLES", 0.0) / (32.0 * cyc) if cyc else 0.0          # per shader engine (32 of them)
    us = m.get("us", 0.0)
    nominal = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * us * 2400.0) if us else 0.0
    print(f"{name:46s} {grid:9d} {len(ds):4d} {us:8.1f} {busy:9.3f} {nominal:9.3f} {sq:8.3f}", file=out)
