#!/bin/bash
# Same-box A/B of decoder.final_dense as ONE launch (csrc/final_block.hip, MDIE_BLOCK_TAIL=1) against the four-launch chain (the default): alternating
# `bench.py` runs on one box.   tools/ab_tail.sh [rounds] [extra bench.py arguments]
ROUNDS=${1:-3}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for i in $(seq 1 "$ROUNDS"); do
  for chain in 1 0; do
    MDIE_BLOCK_TAIL=$((1 - chain)) python bench.py --no-cpu --no-extra "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f'round $i  {\"four-launch chain\" if $chain else \"one-launch block \"}  {d[\"value\"]:9.1f} img/s  {d[\"ms_per_step\"]:.4f} ms/step  kernel_ms {d[\"roofline\"][\"kernel_ms\"]:.4f}  frac {d[\"roofline\"][\"frac\"]:.4f}  frac_step {d[\"roofline\"][\"frac_step\"]:.4f}  launch {d[\"config\"][\"launch\"]}  form {d[\"config\"][\"conv4_form\"].get(\"form\")}')"
  done
done
