#!/bin/bash
# The measurements behind profiles/<tag>_*: run on the GPU box from the repo root, `bash tools/profile_round.sh <tag>`.
# Separate rocprofv3 passes (kernel trace; one --pmc pass per counter group), each bounded by a timeout.
set -e
TAG=${1:-r02_final}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 3 --warmup 1 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
echo "bench cfg2 (f16x3 + f32 leg + cfg5 section + cpu baseline) done"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-leg --no-cfg5 --streams 1 > $OUT/bench_cfg2_one_stream.json 2>/dev/null
echo "bench cfg2 on one stream done"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-leg --no-cfg5 --precision f16x4 > $OUT/bench_cfg2_f16x4.json 2>/dev/null
for c in cfg3 cfg4share; do python3 bench.py --steps 2 --warmup 1 --config $c > $OUT/bench_$c.json 2>/dev/null; echo "bench $c done"; done
python3 bench.py --config cfg5 > $OUT/bench_cfg5.json 2>/dev/null; echo "bench cfg5 done"
python3 tools/two_stream_probe.py 2 > $OUT/two_stream_probe.txt 2>/dev/null
python3 tools/small_job_latency.py > $OUT/small_jobs.txt 2>/dev/null
CODLAD_EDGE_TILE_MAX_NODES=0 CODLAD_NODEQ_MAX_TILES=0 CODLAD_EDGE_WIDE_MAX_TILES=0 CODLAD_NODE_QUAD_MAX_TILES=0 python3 tools/small_job_latency.py > $OUT/small_jobs_round1_kernels.txt 2>/dev/null
echo "small jobs done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-leg --no-cfg5 --streams 1 > $OUT/trace.log 2>&1
python3 tools/kernel_stats.py $OUT/trace > $OUT/kernel_stats.txt
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
echo "trace done"
i=0
while read -r SET; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$i --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-leg --no-cfg5 --streams 1 > $OUT/pmc$i.log 2>&1
  python3 tools/pmc_summary.py $OUT/pmc$i > $OUT/pmc$i.txt
  echo "pmc pass $i done: $SET"
done <<'SETS'
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU
GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
FETCH_SIZE
WRITE_SIZE
SETS
cat $OUT/pmc*.txt > $OUT/pmc_summary.txt
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
# decoder tail (cfg 5): kernel stats, variant A/B, precision of the two message kernels against fp64
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace5 --output-format csv -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 1 > $OUT/trace5.log 2>&1
python3 tools/kernel_stats.py $OUT/trace5 14 > $OUT/decode_kernel_stats.txt
rm -rf $OUT/trace5
python3 tools/ab_variants.py DEC_EDGE_VARIANT=1 DEC_EDGE_VARIANT=0 --rounds 2 --config cfg5 > $OUT/decode_ab.txt 2>/dev/null
python3 tools/decode_precision_probe.py > $OUT/decode_precision.txt 2>/dev/null
echo "decode done"
# the e3nn encoder / prior: latency, kernel stats, counters of the conv kernel on one large graph
bash tools/profile_encoder.sh $TAG/encoder > /dev/null 2>&1
bash tools/pmc_conv.sh $TAG/pmc_conv 2 > $OUT/encoder_pmc_conv.txt 2>&1
cp $OUT/encoder/latency.txt $OUT/encoder_latency.txt; cp $OUT/encoder/kernel_stats.txt $OUT/encoder_kernel_stats.txt
echo "encoder done"
