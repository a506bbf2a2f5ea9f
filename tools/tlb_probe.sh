cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4tlb; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep "Counter_Name" | grep -E "TCP_|TCC_EA0_RDREQ|TCC_.*LATENCY|TA_" | awk '{print $3}' | tr '\n' ' ' > $O/tcp_counters.txt
B="python3 bench.py --no-cpu-baseline --no-parity --no-excited --steps 60 --warmup 6 --preheat 0"
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TCP_UTCL1_STALL_INFLIGHT_MAX" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $B > /dev/null 2> $O/p$i.err || echo "set $i failed: $set"
done
python3 tools/pmc_counters.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 --match wafer_k_step3_fused > $O/tlb_fused3.json 2> $O/tlb.err
cat $O/tlb_fused3.json | head -80; tail -3 $O/tlb.err; for k in 1 2 3 4 5 6; do tail -2 $O/p$k.err; done
find $O -name "*.csv" -size +1M -delete
