#!/bin/bash
# Texture-addresser / L1 (TA / TCP) counters of the two edge kernels, ONE OR TWO counters per pass.  (A six-counter set of
# these was refused at profile construction in round 2 - rocprofiler_create_counter_config error 38, "request exceeds the
# capabilities of the hardware to collect", after which rocprofv3 aborts with signal 6: not a GPU fault, not a tool crash.)
# Workload: tools/edge_variants.py _one = back-to-back launches of msg_kernel_h / upd_kernel_h on bench.py's cfg2 job with
# N(0,1) operands.  usage (GPU box, repo root): bash tools/pmc_tcp.sh <outdir-under-gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_tcp}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
while read -r SET; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i --output-format csv -- python3 tools/edge_variants.py _one > $OUT/p$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $i ($SET): rc $rc"; grep -m2 -E "error code|exceeds" $OUT/p$i.log; fi
  python3 tools/pmc_summary.py $OUT/p$i 2>/dev/null | grep -A4 -E "msg_kernel_h<8, false|upd_kernel_h<8, false" > $OUT/p$i.txt
  echo "pass $i done: $SET"
done <<'SETS'
TA_BUSY_avr TA_TA_BUSY_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
SETS
cat $OUT/p*.txt > $OUT/summary.txt
cat $OUT/summary.txt
