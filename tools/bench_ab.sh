#!/bin/bash
# Same-box A/B of two builds of the library on the headline workload: alternating `bench.py` runs (boxes differ by +-3 % in
# sustained clock, so a build is only ever compared with another build on ONE box, interleaved).
#   tools/bench_ab.sh <other libmdie_hip .so> [rounds]      -> one line per run: which build, images/s, ms/step, serial kernel ms
OTHER=$1; ROUNDS=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for i in $(seq 1 "$ROUNDS"); do
  for lib in "$OTHER" ""; do
    if [ -n "$lib" ]; then name=$(basename "$lib"); else name="libmdie_hip.so (this tree)"; fi
    MDIE_LIB=$lib python bench.py --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f'round $i  {\"$name\":34s} {d[\"value\"]:9.1f} img/s  {d[\"ms_per_step\"]:.4f} ms/step  kernel_ms {d[\"roofline\"][\"kernel_ms\"]:.4f}  frac {d[\"roofline\"][\"frac\"]:.4f}  frac_step {d[\"roofline\"][\"frac_step\"]:.4f}  launch {d[\"config\"][\"launch\"]}')"
  done
done
