#!/bin/bash
# What each stage COSTS THE STEP (as opposed to its serial time): bench.py with the stage's launches left out, same box, same process order.
#   tools/ablate.sh [reps]        (on the GPU box; needs multi-degradation-image-enhancement_amd/libmdie_hip_ablate.so:
#                                  `tools/ablate.sh build` here compiles engine.hip with -DEXP_ABLATE and links it with the shipped objects)
# Results are garbage by construction (a left-out stage leaves its output buffer as it was): only the step time is read.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PKG=$ROOT/multi-degradation-image-enhancement_amd
if [ "${1:-}" = build ]; then
  cd "$PKG/csrc" && make -j8 > /dev/null || exit 1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops \
      -DEXP_ABLATE -c engine.hip -o /tmp/engine_ablate.o || exit 1
  objs=$(ls *.o | grep -v '^engine.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/libmdie_hip_ablate.so" $objs /tmp/engine_ablate.o && echo "built $PKG/libmdie_hip_ablate.so"
  exit $?
fi
REPS=${1:-2}
cd "$ROOT"
export MDIE_LIB=$PKG/libmdie_hip_ablate.so
one() {   # label list -> median ms_per_step of REPS runs
  local best=""
  for r in $(seq "$REPS"); do
    v=$(MDIE_ABLATE="$1" python bench.py --no-cpu --no-extra --no-check --launch eager --steps 200 --warmup 20 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
    best="$best $v"
  done
  python - "$1" $best <<'PY'
import sys
v = sorted(float(x) for x in sys.argv[2:])
print(f"{sys.argv[1] or '(nothing left out)':40s} {v[len(v)//2]*1e3:8.1f} us   ({' '.join(f'{x*1e3:.1f}' for x in v)})")
PY
}
one ""
for s in enc.conv1 enc.conv2 enc.conv3 enc.conv4 dense1 dense2 dense3 "dense1,dense2,dense3" dense1.tr dense2.tr dense3.tr bott dec.conv1 cbam1 dec.conv2 up2 cbam2 dec.conv3 up3 cbam3 dec.conv4 \
         final.l0 final.l1 final.l2 final.l3 "final." "bott,cbam1,cbam2,cbam3" "enc.conv4,bott,dec.conv1,cbam1,dec.conv2"; do
  one "$s"
done
one ""
