#!/bin/bash
# Timing-only elimination builds of the bf16 kernels: tools/elim_build.sh 1 2 4 ...  ->  nerf-ca_amd/lib/libnerfca_hip_exp<bits>.so
# (run a bench against one with NERFCA_LIB=<path>; the results of such a library are wrong by construction -- see NCA_EXP in
# nca_kernels_bf16.hip).  The other objects come from the regular build.
set -e
cd "$(dirname "$0")/.."
make -j4 > /dev/null
for B in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DNCA_EXP=$B -c nerf-ca_amd/csrc/nca_kernels_bf16.hip -o /tmp/nca_bf16_exp$B.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC nerf-ca_amd/csrc/nca_api.o nerf-ca_amd/csrc/nca_kernels_f32.o nerf-ca_amd/csrc/nca_kernels_loss.o /tmp/nca_bf16_exp$B.o -o nerf-ca_amd/lib/libnerfca_hip_exp$B.so && echo built exp$B ) &
done
wait
