#!/bin/bash
# Run ON the GPU box: the bench step with every persistent grid sized for 256 / 192 / 128 / 64 compute units (NCA_CUS).  A kernel bound by what a CU can
# issue takes 256/k times longer on k CUs; one bound by something the whole chip shares (power, HBM, fabric) takes less than that.
# Needs the override build (the shipped library does not read NCA_CUS):  tools/variant_build_all.sh cus "-DNCA_CU_OVERRIDE=1"
for LIB in _cus; do
  L=nerf-ca_amd/lib/libnerfca_hip$LIB.so; [ -f $L ] || continue
  for K in 256 192 128 64; do
    NCA_CUS=$K NERFCA_LIB=$PWD/$L timeout -k 10 200 python3 bench.py --eager --steps 6 --warmup 2 --no-extras --no-cpu-baseline > /tmp/cus.json 2>/dev/null || { echo "lib$LIB cus $K failed"; continue; }
    python3 - "$K" "lib$LIB" <<'PY'
import json, sys
b = json.loads([l for l in open("/tmp/cus.json") if l.startswith("{")][-1]); k = b["roofline"]["all_kernels"]
print(f"{sys.argv[2]:10s} CUs {sys.argv[1]:>3s}: step {b['ms_per_step']:7.3f}  fwd {k['fwd']['avg_ms']:6.3f}  dgrad {k['bwd_dgrad']['avg_ms']:6.3f}  wgrad {k['bwd_wgrad']['avg_ms']:6.3f}", flush=True)
PY
  done
done
