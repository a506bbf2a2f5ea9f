# One GPU-box session of the round: everything writes under gpurun_out/$1/ (merged back by gpurun).
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
    tests)        timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log ;;
    tests_slab)   timeout 600 python -m pytest tests/test_gpu_slab.py -x -q > $O/tests_slab.log 2>&1; tail -5 $O/tests_slab.log ;;
    bench)        timeout 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-300 $O/bench_n1.json ;;
    bench_short)  timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_short.json 2> $O/bench_n1_short.err; cut -c1-300 $O/bench_n1_short.json ;;
    slab)         NCCL_MAX_P2P_NCHANNELS=8 timeout 600 python3 tools/slab_overhead.py --rccl --steps 60 --modes 2,1,0 > $O/slab_overhead.json 2> $O/slab_overhead.err; cat $O/slab_overhead.json ;;
    slab_gate)    WAFER_GATE=1 NCCL_MAX_P2P_NCHANNELS=8 timeout 600 python3 tools/slab_overhead.py --rccl --steps 60 --modes 2 > $O/slab_overhead_gate.json 2> $O/slab_overhead_gate.err; cat $O/slab_overhead_gate.json ;;
    trace2)       NCCL_MAX_P2P_NCHANNELS=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -o t -- python3 tools/slab_trace.py --rccl --mode 2 > $O/trace2.log 2>&1
                  python3 tools/slab_trace.py --parse $(find $O/trace2 -name "*kernel_trace.csv" | head -1) > $O/trace2_timeline.txt 2>&1; cat $O/trace2_timeline.txt; find $O/trace2 -name "*.csv" -size +2M -delete ;;
    trace1)       NCCL_MAX_P2P_NCHANNELS=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -o t -- python3 tools/slab_trace.py --rccl --mode 1 > $O/trace1.log 2>&1
                  python3 tools/slab_trace.py --parse $(find $O/trace1 -name "*kernel_trace.csv" | head -1) > $O/trace1_timeline.txt 2>&1; cat $O/trace1_timeline.txt; find $O/trace1 -name "*.csv" -size +2M -delete ;;
    *)            echo "unknown step $step" ;;
  esac
done
echo "=== done ($(date +%T))"
