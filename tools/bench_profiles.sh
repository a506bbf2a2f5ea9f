cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_bench; mkdir -p $O; cd $R
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py > $O/bench_n1_under_rocprof.json 2> /dev/null
cp $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_512.csv
rm -rf $O/bench_stats
for f in bench_n1 bench_n1_driver_form bench_n1_under_rocprof; do python3 -c "
import json; d=json.load(open('$O/$f.json')); e=d['excited_state_step']
print('$f', round(d['value']/1e9,1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['parity']['max_ulp'], round(d['end_to_end']['ms_per_step'],4), [round(e[k]['ms_per_step'],3) for k in ('k1','k2','k3')], round(d['roofline']['frac'],3), round(d['roofline']['frac_traffic'],3), round(d['roofline']['copy_ceiling_frac'],3), round(d['cpu_baseline']['value']/1e9,2))"; done
grep -E "step3_fused|xstep2" $O/kernel_stats_bench_512.csv | cut -c1-45,150-250
