#!/bin/bash
# rocprofv3 kernel stats of the decoder tail (bench.py --config cfg5); run on the GPU box from the repo root:
#   bash tools/profile_decode.sh <out dir under gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_decode}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 1 > "$OUT/bench.log" 2>&1
python3 tools/kernel_stats.py "$OUT/trace" 14 > "$OUT/kernel_stats.txt"
rm -rf "$OUT/trace"
cat "$OUT/kernel_stats.txt"
