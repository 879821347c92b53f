#!/bin/bash
# A/B build of the WHOLE library with extra compiler flags (macros that more than one translation unit reads, e.g. NCA_WAVES):
#   tools/variant_build_all.sh w4 "-DNCA_WAVES=4"   ->  nerf-ca_amd/lib/libnerfca_hip_w4.so      (run with NERFCA_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
N=$1; F=$2
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude $F"
for f in nca_api nca_kernels_f32 nca_kernels_loss nca_kernels_bf16; do
  $CC -c nerf-ca_amd/csrc/$f.hip -o /tmp/${f}_$N.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/nca_api_$N.o /tmp/nca_kernels_f32_$N.o /tmp/nca_kernels_loss_$N.o /tmp/nca_kernels_bf16_$N.o -o nerf-ca_amd/lib/libnerfca_hip_$N.so
echo built $N
