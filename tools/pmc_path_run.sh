cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/path_bench.py > $O/path.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/path_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/path_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 tools/path_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/sq2 -- python3 tools/path_bench.py > /dev/null 2>&1
cat $O/path.log | grep op
find $O -name "*.csv" | head -20
# keep only small files
find $O -name "*kernel_trace.csv" -size +5M -delete
