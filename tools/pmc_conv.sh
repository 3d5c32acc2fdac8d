#!/bin/bash
# PMC passes over the matrix-pipe conv kernel on one large graph (tools/conv_probe.py): separate passes, --kernel-trace only.
# usage (on the GPU box, from the repo root): bash tools/pmc_conv.sh <outdir-under-gpurun_out> [depth]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_conv}; D=${2:-2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/conv_probe.py $D > $OUT/probe.txt 2>&1
i=0
while read -r SET; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i --output-format csv -- python3 tools/conv_probe.py $D > $OUT/p$i.log 2>&1
  python3 tools/pmc_summary.py $OUT/p$i | grep -A8 "tp_conv_mfma" > $OUT/p$i.txt || true
  rm -rf $OUT/p$i
  echo "pass $i done: $SET"
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY
SETS
cat $OUT/probe.txt $OUT/p*.txt | grep -v amdgpu
