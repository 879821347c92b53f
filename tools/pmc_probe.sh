#!/bin/bash
# Run ON the GPU box: one extra PMC pass over a short bench run -- SQ issue / wait / VMEM counters -- into gpurun_out/pmc_probe/.  usage: bash tools/pmc_probe.sh [env assignments for bench.py]
set -u
OUT=gpurun_out/pmc_probe
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for V in "$@"; do export "$V"; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/sq.err
# (a second pass over the TA / TCP counters of the vector-memory path hung the run on this pool -- the silence watchdog killed it
# after seven minutes -- and is not taken any more)
python3 - <<'PY'
import csv, glob, collections
for d in ("sq",):
    f = glob.glob(f"gpurun_out/pmc_probe/{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no counters (see gpurun_out/pmc_probe/%s.err)" % d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "nca_fused_bf16" in k or "nca_wgrad_bf16" in k:
            agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: round(sum(x[-4:]) / len(x[-4:])) for c, x in v.items()})
PY
