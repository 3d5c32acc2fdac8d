#!/bin/bash
# rocprofv3 kernel stats of the e3nn encoder / prior (tools/encoder_latency.py); run on the GPU box from the repo root:
#   bash tools/profile_encoder.sh <out dir under gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_encoder}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/encoder_latency.py > "$OUT/latency.txt" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 tools/encoder_latency.py > "$OUT/enc.log" 2>&1
python3 tools/kernel_stats.py "$OUT/trace" 24 > "$OUT/kernel_stats.txt"
rm -rf "$OUT/trace"
cat "$OUT/latency.txt" "$OUT/kernel_stats.txt"
