#!/bin/bash
# A diagnostic / A-B variant of the library: ONE source file recompiled with extra flags, linked with the shipped objects of the rest.
#   tools/variant_lib.sh <name> <file.hip> <flags...>   ->  multi-degradation-image-enhancement_amd/libmdie_hip_<name>.so   (select with MDIE_LIB=...)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/multi-degradation-image-enhancement_amd
NAME=$1; SRC=$2; shift 2
cd "$PKG/csrc"
make -j8 > /dev/null
FLAGS=$(make -pn 2>/dev/null | awk -v f="${SRC%.hip}.o" '$0 ~ "^"f": CXXFLAGS \\+=" {sub(/^[^=]*= */, ""); print}' | head -1 || true)
NOPK=""
if make -pn 2>/dev/null | grep -qE "^(.* )?${SRC%.hip}\.o( .*)?: CXXFLAGS \+="; then NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"; fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -Wall -Wno-unused-function $NOPK "$@" -c "$SRC" -o "/tmp/variant_${NAME}.o" 2> >(grep -v "not a recognized feature" >&2)
objs=$(ls *.o | grep -v "^${SRC%.hip}\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/libmdie_hip_${NAME}.so" $objs "/tmp/variant_${NAME}.o"
echo "built $PKG/libmdie_hip_${NAME}.so ($SRC $*)"
