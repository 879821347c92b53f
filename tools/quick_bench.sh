#!/bin/bash
# Run ON the GPU box: per-kernel milliseconds of the bench step under the given environment settings, one line each.
#   tools/quick_bench.sh "NCA_RESIDENT=force" "NCA_RESIDENT=0" "NCA_STAGE_FP8=0" ...
for E in "$@"; do
  env $E timeout -k 10 200 python3 bench.py --eager --no-extras --no-cpu-baseline --steps 6 --warmup 2 > /tmp/qb.json 2>/tmp/qb.err || { echo "$E: bench failed"; tail -3 /tmp/qb.err; continue; }
  python3 -c "
import json,sys; d=json.loads([l for l in open('/tmp/qb.json') if l.startswith('{')][-1]); k=d['roofline']['all_kernels']
print('%-28s step %6.2f  fwd %5.2f x%d  dgrad %5.2f x%d  wgrad %5.2f  loss %.5g' % (sys.argv[1], d['ms_per_step'], k['fwd']['avg_ms'], k['fwd']['launches']//6, k['bwd_dgrad']['avg_ms'], k['bwd_dgrad']['launches']//6, k['bwd_wgrad']['avg_ms'], d['final_loss']))" "$E"
done
