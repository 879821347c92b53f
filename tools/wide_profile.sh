#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root: kernel statistics and the HBM / SQ counter passes of the general kernels
# (tools/wide_bench.py, two 256-unit nets, 8 192 rays x 192 samples) into gpurun_out/prof_wide/.  Counters in their own runs with
# --kernel-trace only; the program itself follows `--`.  tools/wide_profile_summary.py turns it into profiles/<tag>_wide_*.
set -u
OUT=gpurun_out/prof_wide
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/wide_bench.py --widths 256 --steps 3 > $OUT/bench.json 2> $OUT/bench.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 tools/wide_bench.py --widths 256 --steps 1 > /dev/null 2> $OUT/pmc_$C.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc_SQ -- python3 tools/wide_bench.py --widths 256 --steps 1 > /dev/null 2> $OUT/pmc_SQ.err
find $OUT -name "*.csv" | sort
tail -c 300 $OUT/bench.json
