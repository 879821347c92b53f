#!/bin/bash
# Run ON the GPU box: bench.py against the default library and each A/B variant (tools/variant_build.sh), one line per variant:
#   bash tools/ab_bench.sh [steps=20] name1 name2 ...        (name "default" = nerf-ca_amd/lib/libnerfca_hip.so)
STEPS=${1:-20}; shift
for N in "$@"; do
  LIB=nerf-ca_amd/lib/libnerfca_hip_$N.so; [ "$N" = default ] && LIB=nerf-ca_amd/lib/libnerfca_hip.so
  NERFCA_LIB=$PWD/$LIB timeout -k 10 240 python3 bench.py --steps $STEPS --warmup 3 --no-extras --no-cpu-baseline --full-record gpurun_out/ab_${N}_full.json > gpurun_out/ab_$N.json 2> gpurun_out/ab_$N.err
  rc=$?
  [ $rc -ge 124 ] && { echo "$N: killed ($rc)"; exit $rc; }
  python3 - "$N" <<'PY'
import json, sys
n = sys.argv[1]
try:
    b = json.load(open(f"gpurun_out/ab_{n}_full.json"))
    k = b["roofline"]["all_kernels"]
    print(f"{n:14s} graph {b['ms_per_step']:.3f} ms  eager {b['eager_ms_per_step']:.3f}  " + "  ".join(f"{x} {k[x]['ms_per_step']:.3f}" for x in ("fwd", "bwd_dgrad", "bwd_wgrad", "loss", "bwd_reduce")) + f"  loss {b['final_loss']:.6e}", flush=True)
except Exception as e:
    print(n, "failed:", e, open(f"gpurun_out/ab_{n}.err").read()[-400:])
PY
done
