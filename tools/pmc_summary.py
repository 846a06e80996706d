#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: pmc_summary.py <glob of *_counter_collection.csv> [kernel-substring]"""
import csv, collections, glob, sys
files = glob.glob(sys.argv[1], recursive=True)
sub = sys.argv[2] if len(sys.argv) > 2 else "mdie"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if sub not in n: continue
        key = (n.replace('_ZN4mdie', '').replace('EEvNS_8ConvArgsE', '')[:48], int(r['Grid_Size']))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for key in sorted(agg):
    waves = key[1] / 64
    d = sorted(dur[key])[len(dur[key]) // 2]
    print(f"\n== {key[0]} grid={key[1]} waves={waves:.0f} median {d:.1f} us")
    for k in sorted(agg[key]):
        v = sum(agg[key][k]) / len(agg[key][k])
        extra = ""
        if k in ("FETCH_SIZE", "WRITE_SIZE"):
            extra = f"  = {v * 1024 / 1e6:9.1f} MB (KiB units; FETCH_SIZE under-counts wide reads 2x on gfx950)"
        print(f"   {k:24s} {v:16.0f}  per-wave {v / waves:10.1f}{extra}")
