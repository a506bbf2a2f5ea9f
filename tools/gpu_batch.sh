# One GPU-box session: everything writes under gpurun_out/$1/ (merged back by gpurun).  One-off steps of earlier rounds (stamps,
# halo attribution, slab tuning sweeps ...) are in git history; what they found is in profiles/NOTES.md.
# usage: bash tools/gpu_batch.sh <tag> <step> [<step> ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case $step in
    tests)        timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log ;;
    tests_mp)     timeout 900 python -m pytest tests/test_gpu_multiprocess.py -x -q > $O/tests_mp.log 2>&1; tail -15 $O/tests_mp.log; cp gpurun_out/mailbox_latency_world*.txt $O/ 2>/dev/null ;;
    tests_slab)   timeout 900 python -m pytest tests/test_gpu_slab.py -x -q > $O/tests_slab.log 2>&1; tail -5 $O/tests_slab.log ;;
    bench)        timeout 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-300 $O/bench_n1.json ;;
    bench_short)  timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_short.json 2> $O/bench_n1_short.err; cut -c1-300 $O/bench_n1_short.json ;;
    # the default build against every wafer_amd/build/alt_*/ library (tools/build_alt.sh), interleaved, same box:
    #   AB_ARGS: stencil_sweep.py arguments (default: the headline kernel at 512^3)
    ab_alt)       for i in 1 2 3; do
                    timeout 200 python3 tools/stencil_sweep.py ${AB_ARGS:---grid 512,512,512 --rounds 5 --steps 60 --configs v=3} 2>&1 | grep config | sed "s/^/default /"
                    for d in wafer_amd/build/alt_*; do
                      WAFER_HIP_LIB=$PWD/$d/libwafer_hip.so timeout 200 python3 tools/stencil_sweep.py ${AB_ARGS:---grid 512,512,512 --rounds 5 --steps 60 --configs v=3} 2>&1 | grep config | sed "s/^/$(basename $d) /"
                    done
                  done > $O/ab_alt.jsonl; cut -c1-130 $O/ab_alt.jsonl ;;
    slab)         NCCL_MAX_P2P_NCHANNELS=8 timeout 300 python3 tools/slab_overhead.py --rccl --steps 60 --modes ${SLAB_MODES:-3,4,2,1,0} > $O/slab_overhead.json 2> $O/slab_overhead.err; cat $O/slab_overhead.json ;;
    hv_sweep)     timeout 240 python3 tools/hv_sweep.py > $O/hv_sweep.jsonl 2> $O/hv_sweep.err; cat $O/hv_sweep.jsonl; tail -3 $O/hv_sweep.err ;;
    ar_latency)   timeout 300 python3 tools/allreduce_latency.py 2>/dev/null | grep "^{" > $O/allreduce_latency.json; cat $O/allreduce_latency.json ;;
    rows)         timeout 1500 python3 tools/secondary_rows.py $O/rows > $O/rows.log 2>&1; cat $O/rows.log | cut -c1-600 ;;
    # BASELINE config #5's literal flow on ONE device: ./input/potential.csv (64^3) -> device resampler -> 2048^3 fp32 storage,
    # a fixed number of blocks (max_steps; the run then ends with MaxStep, exit code 1), nothing written but the table
    cli_fromfile) python3 examples/make_potential_64.py /tmp/in5/potential.csv
                  sed "s/max_steps: 2000000/max_steps: ${CLI_STEPS:-4000}/; s/save_wavefns: true/save_wavefns: false/" examples/fromfile_2048_f32.yaml > /tmp/fromfile_2048_f32.yaml
                  ( time timeout 1200 wafer_amd/wafer-hip -c /tmp/fromfile_2048_f32.yaml --progress --input-dir /tmp/in5 --output-dir /tmp/out5 ) > $O/fromfile_2048_cli.log 2>&1
                  cat $O/fromfile_2048_cli.log ;;
    # BASELINE config #3 (512^3 Coulomb, ground + 3 excited states) end to end through the driver, in fp64 and on fp32 storage
    cli_hydrogen) for dt in f64 f32; do
                    ( cat examples/hydrogen_512.yaml; printf "gpu:\n    dtype: $dt\n" ) > /tmp/hydrogen_512_$dt.yaml
                    ( time timeout 900 wafer_amd/wafer-hip -c /tmp/hydrogen_512_$dt.yaml --input-dir /tmp/none --output-dir /tmp/out_h512_$dt ) > $O/hydrogen_512_${dt}_cli.log 2>&1
                    grep "^state\|energy =" $O/hydrogen_512_${dt}_cli.log
                  done ;;
    *)            echo "unknown step $step" ;;
  esac
done
echo "=== done ($(date +%T))"
