# The judged set of the round, written under gpurun_out/final/: the bench line, the same command under rocprofv3
# --kernel-trace --stats, HBM-side traffic of its kernels (two --pmc passes), the slab overhead at the bench slab with the
# kernel trace of the single-launch pass, the secondary rows.  Copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py > $O/bench_n1_under_rocprof.json 2> /dev/null
cp $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_512.csv
rm -rf $O/bench_stats
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/fetch $O/write $O/pmc_traffic.json
rm -rf $O/fetch $O/write
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/sq2 -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
python3 tools/pmc_counters.py $O/sq1 $O/sq2 --match wafer_k_step3_fused > $O/sq_counters_fused3.json
rm -rf $O/sq1 $O/sq2
for try in 1 2; do   # (once in this round the first attempt left an empty file: the c10d store's port was still in TIME_WAIT)
  NCCL_MAX_P2P_NCHANNELS=8 MASTER_PORT=$((29455 + try)) python3 tools/slab_overhead.py --rccl --steps 60 --modes 3,4,2,1,0 2> $O/slab_overhead.err | grep "^{" > $O/slab_overhead.json
  [ -s $O/slab_overhead.json ] && break
done
NCCL_MAX_P2P_NCHANNELS=8 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -o t -- python3 tools/slab_trace.py --rccl --mode 2 > /dev/null 2>&1
python3 tools/slab_trace.py --parse $(find $O/trace2 -name "*kernel_trace.csv" | head -1) > $O/slab_single_launch_timeline.txt 2>&1
rm -rf $O/trace2
# every kernel family of the path (excited-state steps k = 1..3, observables, reductions): times, --stats summary, traffic
python3 tools/path_bench.py > $O/path_512.jsonl 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/path_stats -- python3 tools/path_bench.py > /dev/null 2>&1
cp $(find $O/path_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_path_512.csv
rm -rf $O/path_stats
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pfetch -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pwrite -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pfetch $O/pwrite $O/pmc_path_512.json
rm -rf $O/pfetch $O/pwrite
# SQ counters of the excited-state kernels (one and two steps per pass)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/psq1 -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/psq2 -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/psq3 -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
python3 tools/pmc_counters.py $O/psq1 $O/psq2 $O/psq3 --match step > $O/sq_counters_path_512.json
rm -rf $O/psq1 $O/psq2 $O/psq3
# the schedules of the slab pass, interleaved in one process (ten contexts, thirty streams: more hardware queues than the default four)
GPU_MAX_HW_QUEUES=16 python3 tools/hv_sweep.py --configs undecomposed,undecomposed_halves_schedule_no_shorts,mode3,mode3_halves,mode4,mode4_boundary_first,mode4_exchange_after,mode2,mode1,mode0 2>/dev/null | grep variant > $O/slab_pass_breakdown.jsonl
# excited-state steps, one against two steps per pass, same box
for w in 1 2 3; do python3 tools/stencil_sweep.py --grid 512,512,512 --wnum $w --rounds 4 --steps 62 --configs x2=0 x2=1 x2=0 x2=1 2>&1 | grep config | sed "s/^/k=$w /"; done > $O/sweep_x2.jsonl
python3 tools/secondary_rows.py $O/rows > $O/rows.log 2>&1
python3 tools/allreduce_latency.py 2> /dev/null | grep "^{" > $O/allreduce_latency.json
cut -c1-300 $O/bench_n1.json; cat $O/slab_overhead.json; cat $O/sq_counters_fused3.json | head -40
