cd $GRAFT_REPO_ROOT
O=gpurun_out/r05div; mkdir -p $O
run() { # label args...
  L=$1; shift
  for i in 1 2; do
    timeout 300 python3 tools/stencil_sweep.py "$@" 2>&1 | grep config | sed "s/^/$L planned /"
    WAFER_HIP_LIB=$PWD/wafer_amd/build/alt_unplanned/libwafer_hip.so timeout 300 python3 tools/stencil_sweep.py "$@" 2>&1 | grep config | sed "s/^/$L unplanned /"
  done
}
{
run f64_512 --grid 512,512,512 --rounds 5 --steps 60 --configs v=3 v=2 v=1
run f64_1024 --grid 1024,1024,1024 --rounds 3 --steps 30 --configs v=3
run f64_384 --grid 384,384,384 --rounds 5 --steps 60 --configs v=3
run f32_512 --grid 512,512,512 --dtype f32 --rounds 5 --steps 60 --configs v=3
run five_512 --grid 512,512,512 --cd 2 --rounds 5 --steps 60 --configs v=2 v=1
run seven_512 --grid 512,512,512 --cd 3 --rounds 5 --steps 30 --configs v=1
run x1_512 --grid 512,512,512 --wnum 1 --rounds 3 --steps 40 --configs v=1
run x3_512 --grid 512,512,512 --wnum 3 --rounds 3 --steps 40 --configs v=1
} > $O/ab_planned_division.jsonl 2>&1
cut -c1-140 $O/ab_planned_division.jsonl
