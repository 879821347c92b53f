#!/bin/bash
# Run ON the GPU box: which rounding of the bf16 mode costs held-out PSNR?  
#   bash tools/psnr_ablation.sh <seed> <steps> <out.jsonl> <name> [<name> ...]
# name = "product" (the shipped library: f32, f32 from weights moved by 1e-6, bf16 with its weights also rendered in f32) or
# "kick<eps>" (the shipped library in f32, initial weights moved once by a relative eps) or "abl<bits>" (tools/ablation_build.sh: the parity kernels with those bf16 roundings switched on, trained in "f32").
SEED=$1; STEPS=$2; OUT=$3; shift 3
EVERY=$((STEPS / 4))
for N in "$@"; do
  if [ "$N" = product ]; then
    timeout -k 10 400 python3 tools/psnr_run.py --graph --steps $STEPS --every $EVERY --seeds $SEED --variants f32,bf16,f32_perturbed --cross-eval --jsonl $OUT --label product > /dev/null 2> gpurun_out/psnr_abl_$N.err
  elif [ "${N#kick}" != "$N" ]; then          # kick<eps>: the shipped library in f32 from initial weights moved ONCE by eps (relative)
    timeout -k 10 300 python3 tools/psnr_run.py --graph --steps $STEPS --every $EVERY --seeds $SEED --variants f32_perturbed --perturb ${N#kick} --jsonl $OUT --label $N > /dev/null 2> gpurun_out/psnr_abl_$N.err
  else
    NERFCA_LIB=$PWD/nerf-ca_amd/lib/libnerfca_hip_$N.so timeout -k 10 300 python3 tools/psnr_run.py --graph --steps $STEPS --every $EVERY --seeds $SEED --variants f32 --jsonl $OUT --label $N > /dev/null 2> gpurun_out/psnr_abl_$N.err
  fi
  rc=$?
  echo "$N: rc $rc  $(tail -n 1 gpurun_out/psnr_abl_$N.err)"
  [ $rc -ge 124 ] && exit $rc
done
exit 0
