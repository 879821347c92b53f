#!/bin/bash
# Run ON the GPU box: bench.py against the regular library and each elimination build given (tools/elim_build.sh), with the
# plain backward from the store (NCA_ONCHIP=0) and with the on-chip layer; prints per-kernel average milliseconds.
for B in base "$@"; do
  for OC in 0 1; do
    if [ "$B" = base ]; then unset NERFCA_LIB; else export NERFCA_LIB=$PWD/nerf-ca_amd/lib/libnerfca_hip_exp$B.so; fi
    NCA_ONCHIP=$OC timeout -k 10 200 python3 bench.py --eager --no-extras --no-cpu-baseline --steps 4 --warmup 1 > /tmp/elim.json 2>/dev/null
    python3 -c "
import json,sys; d=json.load(open('/tmp/elim.json')); k=d['roofline']['all_kernels']
print('exp %-5s onchip %s  step %6.2f  fwd %5.2f  dgrad %5.2f x%d  wgrad %5.2f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], k['fwd']['avg_ms'], k['bwd_dgrad']['avg_ms'], k['bwd_dgrad']['launches']//4, k['bwd_wgrad']['avg_ms']))" $B $OC
  done
done
