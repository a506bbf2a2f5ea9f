for grid in 160,160,160 176,176,176 192,192,192 200,200,200 320,320,320 64,64,64 96,96,96; do
  for lib in default vec1; do
    if [ $lib = vec1 ]; then export WAFER_HIP_LIB=$PWD/wafer_amd/build/alt_vec1/libwafer_hip.so; else unset WAFER_HIP_LIB; fi
    WAFER_FUSE3_MIN_NY=1 timeout 200 python3 tools/stencil_sweep.py --grid $grid --rounds 5 --steps 60 --configs "v=3" "v=2" 2>&1 | grep config | cut -c1-105 | sed "s/^/$grid $lib /"
  done
done
