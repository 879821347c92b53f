#!/bin/bash
# Run ON the GPU box: board power and clocks (rocm-smi, sampled every ~0.3 s) while bench.py loops over training steps.
python3 bench.py --eager --no-extras --no-cpu-baseline --steps ${1:-600} --warmup 5 ${2:-} > /tmp/pp.json 2>/dev/null &
BP=$!
sleep 12
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|junction|Temperature" | tr -s ' ' | tr '\n' '|'
  echo
  sleep 0.3
done
wait $BP
grep '^{' /tmp/pp.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels']
print('step %6.2f  fwd %5.2f  dgrad %5.2f  wgrad %5.2f' % (d['ms_per_step'], k['fwd']['avg_ms'], k['bwd_dgrad']['avg_ms'], k['bwd_wgrad']['avg_ms']))"
