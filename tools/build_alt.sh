#!/bin/bash
# An alternative build of libwafer_hip.so for same-box A/B runs (tools/gpu_batch.sh ab_alt*): the same sources with extra
# compile-time definitions, into wafer_amd/build/alt_<name>/ (travels to the GPU box; WAFER_HIP_LIB selects it).
#   bash tools/build_alt.sh <name> "-DWAFER_DIAG=8"
set -e
cd "$(dirname "$0")/../wafer_amd/csrc"
NAME=$1; DEFS=$2
OUT=../build/alt_$NAME
mkdir -p $OUT
SRCS=$(python3 -c "import sys; sys.path.insert(0, '../..'); from wafer_amd import build; print(' '.join(build.SOURCES))")
for s in $SRCS; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC $DEFS -c $s -o $OUT/${s%.hip}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $(for s in $SRCS; do echo $OUT/${s%.hip}.o; done) -o $OUT/libwafer_hip.so
rm -f $OUT/*.o
ls -la $OUT/libwafer_hip.so
