# same-box A/B of the default build against wafer_amd/build/alt_$1/ over the kernels of the path: bash tools/ab_div.sh <alt name> <tag>
cd $GRAFT_REPO_ROOT
ALT=$1; O=gpurun_out/$2; mkdir -p $O
run() { # label args...
  L=$1; shift
  for i in 1 2; do
    timeout 300 python3 tools/stencil_sweep.py "$@" 2>&1 | grep config | sed "s/^/$L default /"
    WAFER_HIP_LIB=$PWD/wafer_amd/build/alt_$ALT/libwafer_hip.so timeout 300 python3 tools/stencil_sweep.py "$@" 2>&1 | grep config | sed "s/^/$L $ALT /"
  done
}
{
run f64_512 --grid 512,512,512 --rounds 5 --steps 60 --configs v=3 v=2 v=1
run f64_1024 --grid 1024,1024,1024 --rounds 3 --steps 30 --configs v=3
run f64_384 --grid 384,384,384 --rounds 5 --steps 60 --configs v=3
run f64_256 --grid 256,256,256 --rounds 5 --steps 90 --configs v=3
run f64_128 --grid 128,128,128 --rounds 5 --steps 300 --configs v=3
run f32_512 --grid 512,512,512 --dtype f32 --rounds 5 --steps 60 --configs v=3
run f32fast_512 --grid 512,512,512 --dtype f32fast --rounds 5 --steps 60 --configs v=3
run five_512 --grid 512,512,512 --cd 2 --rounds 5 --steps 60 --configs v=2 v=1
run five_256 --grid 256,256,256 --cd 2 --rounds 5 --steps 90 --configs v=2
run seven_512 --grid 512,512,512 --cd 3 --rounds 5 --steps 30 --configs v=1
run x1_512 --grid 512,512,512 --wnum 1 --rounds 3 --steps 40 --configs v=1
run x2_512 --grid 512,512,512 --wnum 2 --rounds 3 --steps 40 --configs v=1
run x3_512 --grid 512,512,512 --wnum 3 --rounds 3 --steps 40 --configs v=1
run x1_256 --grid 256,256,256 --wnum 1 --rounds 3 --steps 80 --configs v=1
} > $O/ab_$ALT.jsonl 2>&1
cut -c1-125 $O/ab_$ALT.jsonl
