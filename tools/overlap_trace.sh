#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace of the graph-replayed bench step with NCA_OPT_OVERLAP_CUS = $1 (default 128), then the
# start / end stamps of the backward's kernels of ONE replay: do the static net's weight gradient and the dynamic net's dgrad overlap?
V=${1:-128}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ovl_trace
NCA_OVERLAP_CUS=$V rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl_trace -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --kernel-steps 1 > gpurun_out/ovl_trace.json 2> gpurun_out/ovl_trace.err
python3 - "$V" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/ovl_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last nca_prepare_batch on
idx = [i for i, r in enumerate(rows) if "prepare_batch" in r["Kernel_Name"]]
steps = [(a, b) for a, b in zip(idx, idx[1:])]
a, b = steps[-3] if len(steps) >= 3 else steps[-1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"NCA_OVERLAP_CUS={sys.argv[1]}: kernels of one graph-replayed step (us from the step's first kernel; rocprofv3 --kernel-trace)")
print(f"{'kernel':60s} {'stream/queue':>12s} {'grid':>8s} {'start':>9s} {'end':>9s} {'dur':>8s}")
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0][:60]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{n:60s} {r.get('Queue_Id', '?'):>12s} {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8s} {s:9.1f} {e:9.1f} {e - s:8.1f}")
PY
