# The judged pair of the round: the bench line, the same command under rocprofv3 --kernel-trace --stats, and the
# per-kernel stats / PMC passes of the whole path (tools/pmc_path_run.sh).  Writes under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py > $O/bench_n1_under_rocprof.json 2> /dev/null
find $O/bench_stats -name "*kernel_trace.csv" -delete
bash tools/pmc_path_run.sh > $O/path_run.log 2>&1
cat $O/bench_n1.json | cut -c1-400
