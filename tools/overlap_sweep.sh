#!/bin/bash
# Run ON the GPU box: the graph-replayed bench step under NCA_OPT_OVERLAP_CUS = each value given (0 = the plain plan: one
# weight-gradient launch for both nets after both dgrad launches), one line per value.
#   bash tools/overlap_sweep.sh [steps=30] 0 64 96 128 ...
STEPS=${1:-30}; shift
mkdir -p gpurun_out
for V in "$@"; do
  CUS=$V
  NCA_OVERLAP_CUS=$CUS timeout -k 10 300 python3 bench.py --steps $STEPS --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/ovl_$V.json 2> gpurun_out/ovl_$V.err
  rc=$?
  [ $rc -ge 124 ] && { echo "$V: killed ($rc)"; exit $rc; }
  python3 - "$V" <<'PY'
import json, sys
n = sys.argv[1]
try:
    b = json.loads([l for l in open(f"gpurun_out/ovl_{n}.json") if l.startswith("{")][-1])
    k = b["roofline"]["all_kernels"]
    p = b["config"]["plan"]
    print(f"overlap_cus {n:6s} graph {b['ms_per_step']:.3f} ms  eager {b['eager_ms_per_step']:.3f}  " + "  ".join(f"{x} {k[x]['ms_per_step']:.3f}" for x in ("fwd", "bwd_dgrad", "bwd_wgrad", "loss", "bwd_reduce"))
          + f"  splits {p['wgrad']}  overlap {p.get('overlap')}  loss {b['final_loss']:.6e}", flush=True)
except Exception as e:
    print(n, "failed:", e, open(f"gpurun_out/ovl_{n}.err").read()[-600:])
PY
done
