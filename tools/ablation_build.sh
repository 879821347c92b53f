#!/bin/bash
# Rounding-ablation builds of the f32 (parity) kernels: tools/ablation_build.sh 1 2 4 ...  ->  nerf-ca_amd/lib/libnerfca_hip_abl<bits>.so
# Each is the parity library with the bf16 mode's roundings named by the bits switched on (NCA_ABL in nca_kernels_f32.hip: 1 encoded
# input features, 2 hidden-layer weights, 4 hidden activations, 8 dgrad output gradients, 16 layer-0 weights); a PSNR run in "f32" against
# such a library (NERFCA_LIB=<path> python tools/psnr_run.py --variants f32 ...) shows what that rounding alone costs.  DESIGN.md 4.5.
set -e
cd "$(dirname "$0")/.."
make -j4 > /dev/null
for B in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DNCA_ABL=$B -c nerf-ca_amd/csrc/nca_kernels_f32.hip -o /tmp/nca_f32_abl$B.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC nerf-ca_amd/csrc/nca_api.o nerf-ca_amd/csrc/nca_kernels_bf16.o nerf-ca_amd/csrc/nca_kernels_loss.o /tmp/nca_f32_abl$B.o -o nerf-ca_amd/lib/libnerfca_hip_abl$B.so && echo built abl$B ) &
done
wait
