#!/bin/bash
# Run ON the GPU box: where a wave's cycles go in the fused bf16 kernels (elimination build 131072: s_memtime stamps between the
# phases of a tile, printed for two waves of two workgroups per launch; tools/elim_build.sh 131072 first).
export NERFCA_LIB=$PWD/nerf-ca_amd/lib/libnerfca_hip_exp131072.so
for RS in 0 1; do for OC in 0 1; do
  echo "== resident $RS onchip $OC"
  R=force; [ $RS = 0 ] && R=0
  NCA_RESIDENT=$R NCA_ONCHIP=$OC timeout -k 10 200 python3 bench.py --eager --no-extras --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | grep -a "^mode" | sort | uniq | tail -24
done; done
