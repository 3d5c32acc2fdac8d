#!/bin/bash
# PMC passes over the two edge kernels (tools/ablate_edge.py run1 <variant>): separate passes, --kernel-trace only.
# usage (on the GPU box, from the repo root): bash tools/pmc_edge.sh <variant> <outdir-under-gpurun_out>
set -e
V=${1:-main}; OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmc_edge}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
while read -r SET; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i --output-format csv -- python3 tools/ablate_edge.py run1 $V > $OUT/p$i.log 2>&1
  python3 tools/pmc_summary.py $OUT/p$i | grep -A14 -E "msg_kernel_h<8, false|upd_kernel_h<8, false" > $OUT/p$i.txt || true
  echo "pass $i done: $SET"
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY
SETS
cat $OUT/p*.txt
