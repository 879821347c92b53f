#!/bin/bash
# Run ON the GPU box: the f32 (parity mode) bench step against the default library and A/B variants, one line per variant:
#   bash tools/ab_bench_f32.sh [steps=10] name1 name2 ...       (name "default" = nerf-ca_amd/lib/libnerfca_hip.so)
STEPS=${1:-10}; shift
for N in "$@"; do
  LIB=nerf-ca_amd/lib/libnerfca_hip_$N.so; [ "$N" = default ] && LIB=nerf-ca_amd/lib/libnerfca_hip.so
  NERFCA_LIB=$PWD/$LIB timeout -k 10 240 python3 bench.py --prec f32 --steps $STEPS --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/abf_$N.json 2> gpurun_out/abf_$N.err
  rc=$?
  [ $rc -ge 124 ] && { echo "$N: killed ($rc)"; exit $rc; }
  python3 - "$N" <<'PY'
import json, sys
n = sys.argv[1]
try:
    b = json.loads([l for l in open(f"gpurun_out/abf_{n}.json") if l.startswith("{")][-1])
    k = b["roofline"]["all_kernels"]
    print(f"{n:14s} graph {b['ms_per_step']:.3f} ms  eager {b['eager_ms_per_step']:.3f}  " + "  ".join(f"{x} {v['ms_per_step']:.3f}" for x, v in k.items()) + f"  loss {b['final_loss']:.9e}", flush=True)
except Exception as e:
    print(n, "failed:", e, open(f"gpurun_out/abf_{n}.err").read()[-400:])
PY
done
