# HBM-side read / write bytes per launch of the stencil kernels of one stencil_sweep.py configuration (two rocprofv3 --pmc passes,
# tools/pmc_summary.py's corrections) next to its time per step.
#   bash tools/traffic_probe.sh <out_dir> <tag> <stencil_sweep.py arguments ...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/$1; TAG=$2; shift 2
mkdir -p $O; cd $R
python3 tools/stencil_sweep.py "$@" 2>/dev/null | grep config | sed "s/^/$TAG /" >> $O/times.jsonl
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$TAG -- python3 tools/stencil_sweep.py "$@" --rounds 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$TAG -- python3 tools/stencil_sweep.py "$@" --rounds 1 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/f_$TAG $O/w_$TAG $O/pmc_$TAG.json | grep -i "step" | sed "s/^/$TAG /" >> $O/traffic.txt
rm -rf $O/f_$TAG $O/w_$TAG
