#!/bin/bash
# A broad SQ / LDS / TA counter sweep of the three-step kernel (separate --pmc passes over a short bench run).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4sq}; mkdir -p $O
# usage: bash tools/sq_probe.sh <tag> ["extra bench.py arguments", e.g. "--dtype f32"] [kernel-name substring to summarise]
B="python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0 ${2:-}"
MATCH=${3:-wafer_k_step3_fused}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_TRANS_F64" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $B > /dev/null 2> $O/p$i.err || echo "set $i failed: $set"
done
python3 tools/pmc_counters.py $(for k in $(seq 1 $i); do echo $O/p$k; done) --match "$MATCH" > $O/sq_fused3.json 2> $O/sq.err
python3 - <<PY
import json
d = json.load(open("$O/sq_fused3.json"))
for name, c in d.items():
    print(name[:60])
    for k, v in c.items():
        print("  %-34s %14.1f" % (k, v["avg"]))
PY
tail -2 $O/sq.err
find $O -name "*.csv" -size +1M -delete
