#!/bin/bash
# The three-step kernel on a 128 x 8 tile (WaferF3Cfg::RY = 1: eight waves of one row each) against the kernel as built
# (128 x 16), same box: the form the round-5 review asked to be built for 256^3.  The tile height is ONE constant of
# wafer_stencil_fused3.hip.h; this script builds a copy of the sources with it changed (here, before the GPU call):
#   bash tools/ab_tile_128x8.sh build          # -> wafer_amd/build/alt_ry1/libwafer_hip.so (travels with the snapshot)
#   gpurun -- 'bash tools/ab_tile_128x8.sh run' # -> gpurun_out/ry1/{ab.log,parity.log}
# Result (profiles/r06_ab_tile_128x8.jsonl): bit-exact; 128^3 +5..12 %, 256^3 -11 %, 384^3 -17 %, 512^3 -19 %: not built in.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  SRC=$ROOT/wafer_amd/build/src_ry1; OUT=$ROOT/wafer_amd/build/alt_ry1
  rm -rf $SRC; mkdir -p $SRC $OUT
  cp $ROOT/wafer_amd/csrc/*.h $ROOT/wafer_amd/csrc/*.hip $ROOT/wafer_amd/csrc/*.inc $SRC/
  sed -i 's/static constexpr int RY = 2;/static constexpr int RY = 1;/' $SRC/wafer_stencil_fused3.hip.h
  cd $SRC
  SRCS=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); from wafer_amd import build; print(' '.join(build.SOURCES))")
  for s in $SRCS; do
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -I$ROOT/wafer_amd/csrc -c $s -o $OUT/${s%.hip}.o &
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC $(for s in $SRCS; do echo $OUT/${s%.hip}.o; done) -o $OUT/libwafer_hip.so
  rm -f $OUT/*.o
  exit 0
fi
cd $ROOT; mkdir -p gpurun_out/ry1
ALT=$ROOT/wafer_amd/build/alt_ry1/libwafer_hip.so
for rep in 1 2; do
  for g in 256,256,256 128,128,128 384,384,384 512,512,512 64,64,64; do
    for lib in base alt; do
      if [ $lib = alt ]; then export WAFER_HIP_LIB=$ALT; else unset WAFER_HIP_LIB; fi
      echo "== $g $lib rep $rep" >> gpurun_out/ry1/ab.log
      timeout 300 python tools/stencil_sweep.py --grid $g --rounds 5 --steps 60 --configs "v=-1" >> gpurun_out/ry1/ab.log 2>&1
    done
  done
done
export WAFER_HIP_LIB=$ALT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "three_step or config2 or between_the_powers or ragged" > gpurun_out/ry1/parity.log 2>&1
tail -3 gpurun_out/ry1/parity.log
