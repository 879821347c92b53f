#!/bin/bash
# Run ON the GPU box: shader clock during the fused bf16 kernels (elimination builds with bit 65536 print s_memtime ticks per
# s_memrealtime tick of two workgroups per launch):  tools/elim_build.sh 65536 65537 ; tools/clock_probe.sh 65536 65537
for B in "$@"; do
  for OC in 0 1; do
    export NERFCA_LIB=$PWD/nerf-ca_amd/lib/libnerfca_hip_exp$B.so
    NCA_ONCHIP=$OC timeout -k 10 200 python3 bench.py --eager --no-extras --no-cpu-baseline --steps 4 --warmup 1 > /tmp/clk.out 2>/dev/null
    echo "== exp $B onchip $OC"
    grep GHz /tmp/clk.out | awk '{k=$1" "$2; n[k]++; g[k]+=$(NF-1); us[k]+=$9} END {for (k in n) printf "  %s: %.3f GHz  %.0f us (mean of %d)\n", k, g[k]/n[k], us[k]/n[k], n[k]}' | sort
    grep '^{' /tmp/clk.out | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels']
print('  step %6.2f  fwd %5.2f  dgrad %5.2f  wgrad %5.2f' % (d['ms_per_step'], k['fwd']['avg_ms'], k['bwd_dgrad']['avg_ms'], k['bwd_wgrad']['avg_ms']))"
  done
done
if [ -n "${PLAIN:-}" ]; then
  export NERFCA_LIB=$PWD/nerf-ca_amd/lib/libnerfca_hip_exp65536.so
  echo "== plain forward (mode 0)"
  python3 tools/time_plain_forward.py bf16 2>/dev/null | awk '/GHz/ {n++; g+=$(NF-1); us+=$9} !/GHz/ {print "  "$0} END {printf "  mode 0: %.3f GHz %.0f us (mean of %d)\n", g/n, us/n, n}'
fi
