#!/bin/bash
# A diagnostic / A-B variant of the library with SEVERAL source files recompiled with extra flags (tools/variant_lib.sh does one):
#   tools/variant_lib_multi.sh <name> "<a.hip b.hip ...>" <flags...>  ->  multi-degradation-image-enhancement_amd/libmdie_hip_<name>.so  (select with MDIE_LIB=...)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/multi-degradation-image-enhancement_amd
NAME=$1; SRCS=$2; shift 2
cd "$PKG/csrc"
make -j8 > /dev/null
objs=$(ls *.o)
extra=""
for SRC in $SRCS; do
  NOPK=""
  if make -pn 2>/dev/null | grep -qE "^(.* )?${SRC%.hip}\.o( .*)?: CXXFLAGS \+="; then NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -Wall -Wno-unused-function $NOPK "$@" -c "$SRC" -o "/tmp/variant_${NAME}_${SRC%.hip}.o" 2> >(grep -v "not a recognized feature" >&2)
  objs=$(echo "$objs" | grep -v "^${SRC%.hip}\.o$")
  extra="$extra /tmp/variant_${NAME}_${SRC%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/libmdie_hip_${NAME}.so" $objs $extra
echo "built $PKG/libmdie_hip_${NAME}.so ($SRCS $*)"
