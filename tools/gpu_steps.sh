#!/bin/bash
# Runs named GPU steps one after another on the gpurun box; each under its own `timeout -k 10`, output under gpurun_out/<tag>/.
# A step that is KILLED (timeout, signal) stops the sequence: after a hung or faulted GPU step nothing else is started.
#   tools/gpu_steps.sh <tag> "<name>|<seconds>|<command>" ...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd" | tee -a "$out/steps.log"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "   rc=$rc after $(( $(date +%s) - t0 ))s" | tee -a "$out/steps.log"
  tail -3 "$out/$name.log" | cut -c1-300 | sed 's/^/   | /' | tee -a "$out/steps.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then
    echo "   step was killed: stopping here" | tee -a "$out/steps.log"; exit 1
  fi
done
exit 0
