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
    tests_cfg)    timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q > $O/tests_cfg.log 2>&1; tail -5 $O/tests_cfg.log ;;
    tests_mp)     timeout 900 python -m pytest tests/test_gpu_multiprocess.py -x -q > $O/tests_mp.log 2>&1; tail -15 $O/tests_mp.log; cp gpurun_out/mailbox_latency_world*.txt $O/ 2>/dev/null ;;
    ar_latency)   timeout 300 python3 tools/allreduce_latency.py 2>/dev/null | grep "^{" > $O/allreduce_latency.json; cat $O/allreduce_latency.json ;;
    tests_f32)    timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "f32 or fp32" > $O/tests_f32.log 2>&1; tail -5 $O/tests_f32.log ;;
    sweep_f32)    for cd in 1 2; do for dt in f32 f32fast; do timeout 300 python3 tools/stencil_sweep.py --grid 512,512,512 --cd $cd --dtype $dt --rounds 5 --steps 60 --configs "v=2" "v=1" "v=2" "v=1" 2>&1 | grep config | sed "s/^/cd=$cd $dt /"; done; done > $O/sweep_f32.jsonl; cat $O/sweep_f32.jsonl ;;
    tests_f3c)    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_configs.py -x -q -k "three_step or fused3 or single_launch or 512 or config2 or thousand" > $O/tests_f3c.log 2>&1; tail -5 $O/tests_f3c.log ;;
    sweep_f3c)    timeout 300 python3 tools/stencil_sweep.py --grid 512,512,512 --rounds 7 --steps 60 --configs "v=3" "v=2" "v=3" "v=2" > $O/sweep_f3c.jsonl 2>&1; cat $O/sweep_f3c.jsonl ;;
    ab_prev)      for i in 1 2 3; do
                    WAFER_HIP_LIB=$PWD/wafer_amd/build/prev/libwafer_hip.so timeout 200 python3 tools/stencil_sweep.py --grid 512,512,512 --rounds 5 --steps 60 --configs "v=3" 2>&1 | grep config | sed "s/^/prev /"
                    timeout 200 python3 tools/stencil_sweep.py --grid 512,512,512 --rounds 5 --steps 60 --configs "v=3" 2>&1 | grep config | sed "s/^/new  /"
                  done > $O/ab_prev.jsonl; cat $O/ab_prev.jsonl ;;
    ab_excited)   for i in 1 2; do for w in ${AB_K:-1 2 3}; do
                    WAFER_HIP_LIB=$PWD/wafer_amd/build/prev/libwafer_hip.so timeout 200 python3 tools/stencil_sweep.py --grid 512,512,512 --wnum $w --rounds 5 --steps 30 --configs "v=-1" 2>&1 | grep config | sed "s/^/prev k=$w /"
                    timeout 200 python3 tools/stencil_sweep.py --grid 512,512,512 --wnum $w --rounds 5 --steps 30 --configs "v=-1" 2>&1 | grep config | sed "s/^/new  k=$w /"
                  done; done > $O/ab_excited.jsonl; cut -c1-120 $O/ab_excited.jsonl ;;
    tests_x2)     timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "two_excited or excited" > $O/tests_x2.log 2>&1; tail -15 $O/tests_x2.log ;;
    sweep_x2)     for w in ${AB_K:-1 2 3}; do timeout 300 python3 tools/stencil_sweep.py --grid 512,512,512 --wnum $w --potential ${X2_POT:-Coulomb} --rounds 4 --steps 62 --configs ${X2_CONFIGS:-x2=0 x2=1 x2=0 x2=1} 2>&1 | grep config | sed "s/^/k=$w /"; done > $O/sweep_x2.jsonl; cut -c1-140 $O/sweep_x2.jsonl ;;
    ab_alt_x2)    # the excited-state steps of the default build against every wafer_amd/build/alt_*/ library, interleaved, same box
                  for i in 1 2; do for w in ${AB_K:-1 2 3}; do
                    timeout 200 python3 tools/stencil_sweep.py --grid 512,512,512 --wnum $w --rounds 4 --steps 62 --configs ${AB_CFGS:-x2=1} 2>&1 | grep config | sed "s/^/default k=$w /"
                    for d in wafer_amd/build/alt_*; do
                      WAFER_HIP_LIB=$PWD/$d/libwafer_hip.so timeout 200 python3 tools/stencil_sweep.py --grid 512,512,512 --wnum $w --rounds 4 --steps 62 --configs ${AB_CFGS:-x2=1} 2>&1 | grep config | sed "s/^/$(basename $d) k=$w /"
                    done
                  done; done > $O/ab_alt_x2.jsonl; cut -c1-130 $O/ab_alt_x2.jsonl ;;
    prof_x2)      # the kernels of the whole path with the two-step excited kernels: --stats summary, HBM-side traffic, SQ counters
                  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/px_stats -- python3 tools/path_bench.py --steps 22 > $O/path_x2.log 2>&1; grep op $O/path_x2.log
                  cp $(find $O/px_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_path_512.csv; find $O/px_stats -name "*.csv" -size +2M -delete
                  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/px_f -- python3 tools/path_bench.py --steps 22 > /dev/null 2>&1
                  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/px_w -- python3 tools/path_bench.py --steps 22 > /dev/null 2>&1
                  python3 tools/pmc_summary.py $O/px_f $O/px_w $O/pmc_path_512.json | grep -i "step\|apply"
                  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/px_s1 -- python3 tools/path_bench.py --steps 22 > /dev/null 2>&1
                  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/px_s2 -- python3 tools/path_bench.py --steps 22 > /dev/null 2>&1
                  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/px_s3 -- python3 tools/path_bench.py --steps 22 > /dev/null 2>&1
                  python3 tools/pmc_counters.py $O/px_s1 $O/px_s2 $O/px_s3 --match step > $O/sq_path.json; rm -rf $O/px_f $O/px_w $O/px_s1 $O/px_s2 $O/px_s3
                  python3 - <<PY
import json
d = json.load(open("$O/sq_path.json"))
for name, c in d.items():
    g = lambda k: c.get(k, {}).get("avg", 0.0)
    wc = g("SQ_WAVE_CYCLES") or 1.0
    print(name[:100])
    print("   VALU %.1f M  LDS %.1f M  VMEM_RD %.2f M  VMEM_WR %.2f M  SALU %.1f M  bank_conflict_cycles %.1f M  active_valu %.3f  active_any %.3f  wait_inst %.3f  wait_any %.3f" % (
        g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_VMEM_RD") / 1e6, g("SQ_INSTS_VMEM_WR") / 1e6, g("SQ_INSTS_SALU") / 1e6, g("SQ_LDS_BANK_CONFLICT") / 1e6,
        g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc))
PY
                  ;;
    sweep_xf)     for w in 1 2 3; do timeout 300 python3 tools/stencil_sweep.py --grid 512,512,512 --wnum $w --rounds 5 --steps 30 --configs ${XF_CONFIGS:-"xfnw=8" "xfnw=4"} 2>&1 | grep config | sed "s/^/k=$w /"; done > $O/sweep_xf.jsonl; cut -c1-130 $O/sweep_xf.jsonl ;;
    ab_alt)       for i in 1 2 3; do
                    timeout 200 python3 tools/stencil_sweep.py --grid ${AB_GRID:-512,512,512} --rounds 5 --steps 60 --configs ${AB_CFGS:-v=3} 2>&1 | grep config | sed "s/^/default /"
                    for d in wafer_amd/build/alt_*; do
                      WAFER_HIP_LIB=$PWD/$d/libwafer_hip.so timeout 200 python3 tools/stencil_sweep.py --grid ${AB_GRID:-512,512,512} --rounds 5 --steps 60 --configs ${AB_CFGS:-v=3} 2>&1 | grep config | sed "s/^/$(basename $d) /"
                    done
                  done > $O/ab_alt.jsonl; cut -c1-110 $O/ab_alt.jsonl ;;
    sq_f3c)       for k in 0 1; do
                    WAFER_F3_KERNEL=$k timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1_$k -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
                    WAFER_F3_KERNEL=$k timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/sq2_$k -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
                    python3 tools/pmc_counters.py $O/sq1_$k $O/sq2_$k --match wafer_k_step3_fused > $O/sq_f3_kernel$k.json; rm -rf $O/sq1_$k $O/sq2_$k
                    python3 - <<PY
import json
d = json.load(open("$O/sq_f3_kernel$k.json"))
for name, c in d.items():
    print("kernel $k", {k: round(v["avg"] / 1e6, 1) for k, v in c.items()})
PY
                  done ;;
    sweep_sizes)  for grid in 256,256,256 320,320,320 384,384,384 448,448,448 640,640,640 1024,1024,128; do timeout 300 python3 tools/stencil_sweep.py --grid $grid --rounds 5 --steps 60 --configs "v=-1" 2>&1 | grep config | sed "s/^/$grid /"; done > $O/sweep_sizes.jsonl; cut -c1-150 $O/sweep_sizes.jsonl ;;
    sweep_f3c_sizes) for grid in 256,256,256 384,384,384 1024,1024,128; do timeout 300 python3 tools/stencil_sweep.py --grid $grid --rounds 7 --steps 60 --configs "v=3" "v=2" "v=3" "v=2" 2>&1 | grep config | sed "s/^/$grid /"; done > $O/sweep_f3c_sizes.jsonl; cat $O/sweep_f3c_sizes.jsonl ;;
    tests_slab)   timeout 600 python -m pytest tests/test_gpu_slab.py -x -q > $O/tests_slab.log 2>&1; tail -5 $O/tests_slab.log ;;
    bench)        timeout 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-300 $O/bench_n1.json ;;
    bench_short)  timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_short.json 2> $O/bench_n1_short.err; cut -c1-300 $O/bench_n1_short.json ;;
    slab)         NCCL_MAX_P2P_NCHANNELS=8 timeout 300 python3 tools/slab_overhead.py --rccl --steps 60 --modes ${SLAB_MODES:-3,2,1,0} > $O/slab_overhead.json 2> $O/slab_overhead.err; cat $O/slab_overhead.json ;;
    slab_noshort) WAFER_HV_DEBUG=8 NCCL_MAX_P2P_NCHANNELS=8 timeout 240 python3 tools/slab_overhead.py --rccl --steps 60 --modes 2 > $O/slab_overhead_noshort.json 2> $O/slab_overhead_noshort.err; cat $O/slab_overhead_noshort.json ;;
    trace2)       NCCL_MAX_P2P_NCHANNELS=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -o t -- python3 tools/slab_trace.py --rccl --mode 2 > $O/trace2.log 2>&1
                  python3 tools/slab_trace.py --parse $(find $O/trace2 -name "*kernel_trace.csv" | head -1) > $O/trace2_timeline.txt 2>&1; cat $O/trace2_timeline.txt; find $O/trace2 -name "*.csv" -size +2M -delete ;;
    trace1)       NCCL_MAX_P2P_NCHANNELS=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -o t -- python3 tools/slab_trace.py --rccl --mode 1 > $O/trace1.log 2>&1
                  python3 tools/slab_trace.py --parse $(find $O/trace1 -name "*kernel_trace.csv" | head -1) > $O/trace1_timeline.txt 2>&1; cat $O/trace1_timeline.txt; find $O/trace1 -name "*.csv" -size +2M -delete ;;
    hv_sweep)     timeout 240 python3 tools/hv_sweep.py > $O/hv_sweep.jsonl 2> $O/hv_sweep.err; cat $O/hv_sweep.jsonl; tail -3 $O/hv_sweep.err ;;
    hv_sweep_rccl) timeout 240 python3 tools/hv_sweep.py --rccl > $O/hv_sweep_rccl.jsonl 2> $O/hv_sweep_rccl.err; cat $O/hv_sweep_rccl.jsonl; tail -3 $O/hv_sweep_rccl.err ;;
    trace2lb)     timeout 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/trace2lb -o t -- python3 tools/slab_trace.py --mode 2 > $O/trace2lb.log 2>&1
                  python3 tools/trace_tail.py $(find $O/trace2lb -name "*kernel_trace.csv" | head -1) 70 > $O/trace2lb_kernels.txt 2>&1; cat $O/trace2lb_kernels.txt
                  python3 - <<PY > $O/trace2lb_hip_api.txt 2>&1
import csv, glob, collections
f = glob.glob("$O/trace2lb/**/*hip_api_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[len(rows) // 2:]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Function"]][0] += 1
    agg[r["Function"]][1] += d
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{k:40s} calls {n:6d}  total {t / 1e3:10.1f} us  mean {t / n / 1e3:8.2f} us")
PY
                  cat $O/trace2lb_hip_api.txt; find $O/trace2lb -name "*.csv" -size +3M -delete ;;
    slab_tune)    : > $O/slab_tune.jsonl
                  while read -r envs; do
                    [ -z "$envs" ] && continue
                    echo "{\"env\": \"$envs\", \"result\": $(env $envs timeout 200 python3 tools/slab_overhead.py --rccl --steps 60 --modes 2 2>/dev/null | grep "^{" | tail -1)}" >> $O/slab_tune.jsonl
                  done < $R/tools/slab_tune_envs.txt
                  python3 - <<PY
import json
for line in open("$O/slab_tune.jsonl"):
    try:
        d = json.loads(line)
        r = d["result"]
        print(f"{d['env']:70s} undecomposed {r['undecomposed_ms_per_step']:.4f}  loopback {r['slab_ms_per_step_overlap_2']:.4f}  rccl-self {r['slab_native_rccl_self_ms_per_step_overlap_2']:.4f}  ratio {r['slab_native_rccl_self_ms_per_step_overlap_2'] / r['undecomposed_ms_per_step']:.3f}")
    except Exception as e:
        print("bad line", line[:200], e)
PY
                  ;;
    # FETCH_SIZE per kernel of the path under WAFER_XCD_SWIZZLE = $HALO_SWZ (1 = the XCD-contiguous tile order, 0 = off).  The
    # attribution runs of round 3 (profiles/r03_halo_attribution.json) used further bits of that variable in a scratch build of
    # wafer_stencil_lds.hip.h (redirected / time-shifted halo requests); those bits do not exist in the committed kernels.
    pmc_halo)     rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt
                  for swz in ${HALO_SWZ:-1 3 5 7}; do
                    WAFER_XCD_SWIZZLE=$swz timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/halo_f$swz -- python3 tools/path_bench.py --steps 10 > $O/halo_path_$swz.log 2>&1
                    python3 - <<PY
import csv, glob, collections
for d in ("$O/halo_f$swz", "$O/halo_t$swz"):
    v = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            v[(row["Kernel_Name"][:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), x in sorted(v.items()):
        if "step" in k or "observ" in k: print("swz=$swz", c, round(sum(x) / len(x)), len(x), k)
PY
                    grep evolve $O/halo_path_$swz.log | cut -c1-60
                    rm -rf $O/halo_f$swz $O/halo_t$swz
                  done > $O/pmc_halo.txt 2>&1; cat $O/pmc_halo.txt ;;
    sq_path)      timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sqp1 -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
                  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/sqp2 -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
                  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/sqp3 -- python3 tools/path_bench.py --steps 10 > /dev/null 2>&1
                  python3 tools/pmc_counters.py $O/sqp1 $O/sqp2 $O/sqp3 --match wafer_k_step > $O/sq_path.json; rm -rf $O/sqp1 $O/sqp2 $O/sqp3
                  python3 - <<PY
import json
d = json.load(open("$O/sq_path.json"))
for name, c in d.items():
    g = lambda k: c.get(k, {}).get("avg", 0.0)
    wc = g("SQ_WAVE_CYCLES") or 1.0
    print(name[:90])
    print("   VALU %.1f M  LDS %.1f M  VMEM_RD %.2f M  VMEM_WR %.2f M  SALU %.1f M  active_valu %.3f  active_any %.3f  wait_inst %.3f  wait_any %.3f" % (
        g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_VMEM_RD") / 1e6, g("SQ_INSTS_VMEM_WR") / 1e6, g("SQ_INSTS_SALU") / 1e6,
        g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc))
PY
                  ;;
    rows)         timeout 1500 python3 tools/secondary_rows.py $O/rows > $O/rows.log 2>&1; cat $O/rows.log | cut -c1-600 ;;
    *)            echo "unknown step $step" ;;
  esac
done
echo "=== done ($(date +%T))"
