#!/bin/bash
# 8 against 4 waves per workgroup in the one-step excited-state kernels, per storage type, stencil and number of stored states
# (512^3, ms per step): the measurements behind wafer_excited_nw (wafer_stencil_lds.hip.h)
for spec in "f64 1 3" "f64 2 1" "f64 2 2" "f64 2 3" "f64 3 1" "f64 3 2" "f64 3 3" "f32 1 3" "f32 2 1" "f32 2 2" "f32 2 3" "f32 3 1" "f32 3 2" "f32 3 3"; do set -- $spec
  WAFER_X2=0 timeout 300 python3 tools/stencil_sweep.py --grid 512,512,512 --dtype $1 --cd $2 --wnum $3 --rounds 3 --steps 30 --configs "v=-1,xfnw=4" "v=-1,xfnw=8" "v=-1" 2>&1 | grep config | cut -c12-70 | sed "s/^/$1 cd=$2 k=$3 /"
done
