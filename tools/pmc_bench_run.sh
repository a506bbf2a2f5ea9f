# HBM-side traffic of the bench kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one) over a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcb
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/fetch $O/write $O/pmc_traffic_new.json
