#!/bin/bash
# Run ON the GPU box: shader clock during the fused bf16 kernels.  The diagnostic build with NCA_EXP bit 65536 (tools/elim_build.sh 65536)
# stamps s_memtime / s_memrealtime around every fused launch in two workgroups and prints shader cycles per 10 ns tick = GHz; this
# script runs 60 eager steps against it (the chip is warm by then) and averages the last launches per kernel mode.  No stamp executes in
# the product library.     bash tools/clock_probe.sh > gpurun_out/clock_probe.txt
export NERFCA_LIB=${NERFCA_LIB:-$PWD/nerf-ca_amd/lib/libnerfca_hip_exp65536.so}       # (or a stamped variant: NERFCA_LIB=... bash tools/clock_probe.sh)
[ -f "$NERFCA_LIB" ] || { echo "build it first: tools/elim_build.sh 65536"; exit 1; }
timeout -k 10 300 python3 bench.py --eager --no-extras --no-cpu-baseline --steps 60 --warmup 5 --kernel-steps 4 > /tmp/clk.out 2>/dev/null
grep GHz /tmp/clk.out | tail -400 | awk '{k=$1" "$2; n[k]++; g[k]+=$(NF-1); us[k]+=$9} END {for (k in n) printf "  %s: %.3f GHz  %.0f us per launch (mean of %d stamps)\n", k, g[k]/n[k], us[k]/n[k], n[k]}' | sort
grep '^{' /tmp/clk.out | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels']
print('  eager step %6.2f ms over %d steps;  fwd %5.2f  dgrad %5.2f  wgrad %5.2f ms per launch (stamped build)' % (d['ms_per_step'], d['steps'], k['fwd']['avg_ms'], k['bwd_dgrad']['avg_ms'], k['bwd_wgrad']['avg_ms']))"
