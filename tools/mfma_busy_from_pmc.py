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


# ---- the six wide 3x3 layers one by one (north star: MFMA-busy of the MFMA-bound convolutions, per layer) -------------------------------
# conv_wide_kernel launches all sit on the forward's main stream, so their dispatch order within a step is the network's:
# encoder.conv2, conv3, conv4, decoder.conv1, conv2, conv3 (models/cdan.py:59-61,103-111) -- the k-th of them (mod 6) is layer k.
LAYERS = [("enc.conv2+pool", 64, 128, 128), ("enc.conv3+pool", 128, 256, 64), ("enc.conv4", 256, 512, 32), ("dec.conv1", 512, 256, 32),
          ("dec.conv2", 256, 128, 32), ("dec.conv3", 128, 64, 64)]          # name, cin, cout, map edge at 256x256 input
wide = [rows[k] for k in sorted(rows) if "conv_wide_kernel" in rows[k].get("name", "") and "SQ_VALU_MFMA_BUSY_CYCLES" in rows[k] and "GRBM_GUI_ACTIVE" in rows[k]]
if wide and len(wide) % 6 == 0:
    B = int(os.environ.get("BATCH", 32))
    print("\n# the six wide 3x3 layers, median over the profiled steps (B = %d, 256x256; PMC passes serialise the launches: times are the" % B, file=out)
    print("# kernel alone).  TFLOP/s = 2 * 9 * cin * cout * pixels * B / time; frac = TFLOP/s / 2500 (dense bf16 MFMA peak)", file=out)
    print(f"{'layer':16s} {'n':>3s} {'us':>8s} {'mfma_busy':>9s} {'vs_2.4GHz':>9s} {'sq_busy':>8s} {'TFLOP/s':>8s} {'frac':>6s}", file=out)
    for i, (name, cin, cout, edge) in enumerate(LAYERS):
        ds = sorted(wide[i::6], key=lambda d: d.get("us", 0.0))
        m = ds[len(ds) // 2]
        cyc, us = m["GRBM_GUI_ACTIVE"] / 8.0, m.get("us", 0.0)
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else 0.0
        nominal = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * us * 2400.0) if us else 0.0
        sq = m.get("SQ_BUSY_CYCLES", 0.0) / (32.0 * cyc) if cyc else 0.0
        tf = 2.0 * 9 * cin * cout * edge * edge * B / (us * 1e-6) / 1e12 if us else 0.0
        print(f"{name:16s} {len(ds):3d} {us:8.1f} {busy:9.3f} {nominal:9.3f} {sq:8.3f} {tf:8.0f} {tf / 2500.0:6.3f}", file=out)
elif wide:
    print(f"\n# {len(wide)} conv_wide dispatches: not a multiple of 6, per-layer table skipped", file=out)
