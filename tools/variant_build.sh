#!/bin/bash
# A/B builds of the bf16 kernels with other values of their tuning macros (NCA_BF_PF, NCA_BF_MINBLOCKS ...; the round-2 / round-3
# experiment macros are gone from these sources: tools/r03_experiments.sh builds those from the tag r03-kernels):
#   tools/variant_build.sh pf2 "-DNCA_BF_PF=2"  [name2 "flags2" ...]   ->  nerf-ca_amd/lib/libnerfca_hip_<name>.so
# Run a bench or the tests against one with NERFCA_LIB=<path>.  These are CORRECT libraries (unlike the timing-only builds of tools/r03_experiments.sh elim): a variant
# that wins becomes the default in the source.  The other objects come from the regular build.
set -e
cd "$(dirname "$0")/.."
make -j4 > /dev/null
while [ $# -ge 2 ]; do
  N=$1; F=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $F -c nerf-ca_amd/csrc/nca_kernels_bf16.hip -o /tmp/nca_bf16_$N.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC nerf-ca_amd/csrc/nca_api.o nerf-ca_amd/csrc/nca_kernels_f32.o nerf-ca_amd/csrc/nca_kernels_loss.o /tmp/nca_bf16_$N.o -o nerf-ca_amd/lib/libnerfca_hip_$N.so && echo built $N ) &
done
wait
