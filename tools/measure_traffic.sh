#!/bin/bash
# The two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate, as MI355X_MICROARCH.md prescribes) + the traffic file bench.py echoes,
# keyed to the hash of the kernel sources it was measured on:  tools/measure_traffic.sh <tag>   (a subset of tools/measure_all.sh)
set -u
TAG=${1:-traffic}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/fetch" -- python3 "$ROOT/bench.py" --launch eager --steps 10 --warmup 2 --no-cpu --no-extra > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/write" -- python3 "$ROOT/bench.py" --launch eager --steps 10 --warmup 2 --no-cpu --no-extra > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
python tools/traffic_from_pmc.py "$OUT/pmc" "$OUT/traffic_bf16_b32_256.json" > "$OUT/traffic_summary.txt" 2>&1
cp "$OUT/traffic_bf16_b32_256.json" profiles/traffic_bf16_b32_256.json 2>/dev/null
timeout -k 10 300 python bench.py > "$OUT/bench_bf16.txt" 2> "$OUT/bench_bf16.err"
timeout -k 10 300 python tools/profile_layers.py bf16 > "$OUT/layers_bf16.txt" 2>&1
tail -1 "$OUT/traffic_summary.txt"; tail -c 600 "$OUT/bench_bf16.txt" | head -c 300
