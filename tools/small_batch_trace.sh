#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace of the graph-replayed step at the reference's default batch (1 024 rays x 500 samples): every
# kernel of one replay with its start / end -- where the 0.7 ms go when the kernels themselves are tens of microseconds long.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sb_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sb_trace -- python3 bench.py --rays 1024 --samples 500 --steps 30 --warmup 5 --no-extras --no-cpu-baseline --kernel-steps 1 > gpurun_out/sb_trace.json 2> gpurun_out/sb_trace.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/sb_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "prepare_batch" in r["Kernel_Name"] or "begin_step" in r["Kernel_Name"]]
a, b = idx[-12], idx[-11]
t0 = int(rows[a]["Start_Timestamp"])
print("kernels of one graph-replayed step at 1 024 rays x 500 samples (us from the step's first kernel; rocprofv3 --kernel-trace)")
busy = 0.0
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0][:64]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    busy += e - s
    print(f"{n:64s} {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8s} {s:8.1f} {e:8.1f} {e - s:7.1f}")
last = (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3
nxt = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print(f"{b - a} kernels, busy {busy:.1f} us of {last:.1f} us from first start to last end; next step's first kernel starts at {nxt:.1f} us")
PY
tail -c 300 gpurun_out/sb_trace.json
