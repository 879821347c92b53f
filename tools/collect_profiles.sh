#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root: takes the bench line, the rocprofv3 kernel statistics and the
# PMC passes the numbers in DESIGN.md / profiles/ come from, into gpurun_out/prof_<prec>/.  PMC counters are collected in
# their own runs with --kernel-trace only (never with --sys-trace / --hip-trace).  The program itself follows `--`.
#   usage: bash tools/collect_profiles.sh [bf16|f32] [suffix]      (NCA_STAGE_FP8=0 ... bf16 _pure: the bf16-staging variant)
set -u
PREC=${1:-bf16}
OUT=gpurun_out/prof_$PREC${2:-}
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=20; [ "$PREC" = f32 ] && STEPS=8
python3 bench.py --prec $PREC --steps $STEPS --warmup 3 --no-extras --pure-steps 0 --full-record $OUT/bench_full.json > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --prec $PREC --steps 6 --warmup 2 --no-cpu-baseline --no-extras --eager --full-record $OUT/bench_under_rocprof_full.json > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 bench.py --prec $PREC --steps 2 --warmup 1 --no-cpu-baseline --no-extras --eager --full-record $OUT/pmc_run_full.json > /dev/null 2> $OUT/pmc_$C.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc_SQ -- python3 bench.py --prec $PREC --steps 2 --warmup 1 --no-cpu-baseline --no-extras --eager --full-record $OUT/pmc_run_full.json > /dev/null 2> $OUT/pmc_SQ.err
find $OUT -name "*.csv" | sort
tail -c 400 $OUT/bench.json
