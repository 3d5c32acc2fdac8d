#!/bin/bash
# encoder latency of library builds on ONE box, alternating: bash tools/ab_encoder.sh variants/libcodlad_x.so ...
for rep in 1 2; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then out=$(python3 tools/encoder_latency.py 2>/dev/null | grep -v amdgpu | tr '\n' ' ')
    else out=$(CODLAD_HIP_LIB=$PWD/$lib python3 tools/encoder_latency.py 2>/dev/null | grep -v amdgpu | tr '\n' ' '); fi
    echo "$lib: $out" | sed -E 's/ for 40 frames \([^)]*\)//g'
  done
done
