#!/bin/bash
# Builds measurement-only variants of the matrix-pipe conv kernel (CODLAD_TP_ABLATE = 1, 2, 3: see encoder_mfma_kernel.hip)
# into variants/ (here, no GPU needed); on the GPU box: bash tools/ab_encoder.sh variants/libcodlad_tpabl{1,2,3}.so
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
OBJS=$(ls codlad_amd/csrc/*.o | grep -v encoder_mfma_kernel | grep -v qlds)
for k in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DCODLAD_TP_ABLATE=$k -c codlad_amd/csrc/encoder_mfma_kernel.hip -o /tmp/tpabl$k.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libcodlad_tpabl$k.so $OBJS /tmp/tpabl$k.o
done
ls -la variants/libcodlad_tpabl*.so
