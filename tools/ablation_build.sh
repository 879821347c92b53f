#!/bin/bash
# Rounding-ablation builds of the f32 (parity) kernels: tools/ablation_build.sh 1 2 23m11 96 ...  ->  nerf-ca_amd/lib/libnerfca_hip_abl<name>.so
# Each is the parity library with the roundings named by the bits switched on (NCA_ABL in nca_kernels_f32.hip: 1 encoded input features,
# 2 hidden-layer weights, 4 hidden activations, 8 dgrad output gradients, 16 layer-0 weights; 32 / 64: what the weight-gradient kernel
# reads of the layer inputs / output gradients to e4m3's / e5m2's 4 / 3 significant bits); "<bits>m<k>" rounds to k significant bits
# instead of bf16's 8 (11 = f16's precision); 256 / 512: the stored hidden-layer inputs / output gradients as MX fp6 (e2m3 / e3m2) under one
# power-of-two scale per lane and 32 values (768 = both: the 6-bit staging of DESIGN.md 7, tools/r04_fp6_staging.patch).  A PSNR run in "f32" against such a library (NERFCA_LIB=<path> python tools/psnr_run.py
# --variants f32 ...) shows what that rounding alone costs.  DESIGN.md 4.5.
set -e
cd "$(dirname "$0")/.."
make -j4 > /dev/null
for N in "$@"; do
  B=${N%%m*}; M=8; [ "$B" != "$N" ] && M=${N##*m}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DNCA_ABL=$B -DNCA_ABL_MANT=$M -c nerf-ca_amd/csrc/nca_kernels_f32.hip -o /tmp/nca_f32_abl$N.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC nerf-ca_amd/csrc/nca_api.o nerf-ca_amd/csrc/nca_kernels_bf16.o nerf-ca_amd/csrc/nca_kernels_loss.o /tmp/nca_f32_abl$N.o -o nerf-ca_amd/lib/libnerfca_hip_abl$N.so && echo built abl$N ) &
done
wait
