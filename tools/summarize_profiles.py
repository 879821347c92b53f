#!/usr/bin/env python3
"""Turn gpurun_out/prof_<prec>/ (written by tools/collect_profiles.sh on the GPU box) into the committed summaries:
    profiles/<tag>_<prec>_bench.json, _bench_under_rocprof.json, _bench_kernel_stats.csv,
    _pmc_FETCH_SIZE.csv / _pmc_WRITE_SIZE.csv (rows of this library's kernels), _pmc_traffic.json, _pmc_sq.json
usage: python tools/summarize_profiles.py bf16 [tag=r01] [suffix]      (suffix: the collect_profiles.sh suffix, e.g. _pure; the files
are then named <tag>_<prec><suffix>_*)"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prec = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
suffix = sys.argv[3] if len(sys.argv) > 3 else ""
src = os.path.join(ROOT, "gpurun_out", f"prof_{prec}{suffix}")
sys.path.insert(0, ROOT)
from bench import source_sha          # (hash of the kernel sources: bench.py replays a PMC summary only on the sources it was taken on)
dst = os.path.join(ROOT, "profiles")
# candidate kernel names per role, most specific first (kernel modes: 0 forward, 1 recompute backward, 2 storing forward,
# 3 backward from the store, 4 the same with the last hidden layer's weight gradient on chip, 5 backward from an fp8-staged store
# without recompute; modes 2 and 5 run one launch per net when the weight images are resident in LDS)
# (no closing bracket: the f32 kernels carry a third template argument, the split-bf16 switch)
KERNELS = {"fwd": [f"nca_fused_{prec}<128, 2", f"nca_fused_{prec}<128, 0", f"nca_fused_{prec}<128, false"],
           "bwd_dgrad": [f"nca_fused_{prec}<128, 5", f"nca_fused_{prec}<128, 4", f"nca_fused_{prec}<128, 3", f"nca_fused_{prec}<128, 1", f"nca_fused_{prec}<128, true"],
           "bwd_wgrad": ["nca_wgrad_bf16<128, true", "nca_wgrad_bf16<128, false", "nca_wgrad_bf16<128>"] if prec == "bf16" else ["nca_wgrad_f32x3", "nca_wgrad_f32"], "bwd_reduce": ["nca_reduce_f32"],
           "loss": ["nca_loss_rays"]}


def pick(table, names):
    """(kernel key in `table`, matched name) for the first candidate that occurs."""
    for name in names:
        ks = [k for k in table if name in k]
        if ks:
            return ks[0], name
    return None, None


def last_json_line(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return max(hits, key=os.path.getmtime)        # gpurun merges runs into the same directory: take the newest


# (round 6: bench.py's stdout ends with the compact headline line; the FULL record -- every kernel table -- is the file it was told to write)
full = os.path.join(src, "bench_full.json")
bench = json.load(open(full)) if os.path.exists(full) else last_json_line(os.path.join(src, "bench.json"))
json.dump(bench, open(os.path.join(dst, f"{tag}_{prec}{suffix}_bench.json"), "w"))
full_r = os.path.join(src, "bench_under_rocprof_full.json")
json.dump(json.load(open(full_r)) if os.path.exists(full_r) else last_json_line(os.path.join(src, "bench_under_rocprof.json")),
          open(os.path.join(dst, f"{tag}_{prec}{suffix}_bench_under_rocprof.json"), "w"))
with open(find("stats/**/*kernel_stats.csv")) as f, open(os.path.join(dst, f"{tag}_{prec}{suffix}_bench_kernel_stats.csv"), "w") as g:
    g.write(f.read())


def per_kernel(counter_dir, keep_rows_to=None):
    rows = list(csv.DictReader(open(find(f"{counter_dir}/**/*counter_collection.csv"))))
    ours = [r for r in rows if "nca_" in r["Kernel_Name"]]
    if keep_rows_to:
        with open(keep_rows_to, "w", newline="") as g:
            w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(ours)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in ours:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean_last(v, n=20):
    v = v[-n:]
    return sum(v) / len(v)


fetch = per_kernel("pmc_FETCH_SIZE", os.path.join(dst, f"{tag}_{prec}{suffix}_pmc_FETCH_SIZE.csv"))
write = per_kernel("pmc_WRITE_SIZE", os.path.join(dst, f"{tag}_{prec}{suffix}_pmc_WRITE_SIZE.csv"))
def _commit():
    import subprocess
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return None


traffic = {"commit": _commit(), "source_sha": source_sha(), "how": "rocprofv3 --pmc FETCH_SIZE (and, in a second run, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py "
                  "--steps 2 --warmup 1 --no-cpu-baseline; per-kernel mean over the last <=20 dispatches; bytes = counter * 1024; FETCH_SIZE "
                  "doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md section HBM); WRITE_SIZE as is",
           "config": {"prec": prec, "stage_fp8": bool(bench["config"].get("stage_fp8")), "rays_per_step": bench["config"]["rays_per_step_per_gpu"], "samples_per_ray": bench["config"]["samples_per_ray"],
                      "ray_chunks_per_step": bench["roofline"]["all_kernels"]["bwd_wgrad"]["launches"] // bench["steps"],
                      "dgrad_launches_per_step": bench["roofline"]["all_kernels"]["bwd_dgrad"]["launches"] // bench["steps"],
                      "fwd_launches_per_step": bench["roofline"]["all_kernels"]["fwd"]["launches"] // bench["steps"]},
           "kernels": {}}
for key, names in KERNELS.items():
    fk, name = pick(fetch, names)
    wk, _ = pick(write, names)
    if fk is None or wk is None:
        continue
    rd = mean_last(fetch[fk]["FETCH_SIZE"]) * 1024 * 2
    wr = mean_last(write[wk]["WRITE_SIZE"]) * 1024
    traffic["kernels"][key] = {"kernel": name, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
json.dump(traffic, open(os.path.join(dst, f"{tag}_{prec}{suffix}_pmc_traffic.json"), "w"), indent=1)
# bench.py reads roofline.traffic from the traffic file that was committed when it ran; the bench line kept here gets the
# figure of THIS collection (same box, same build) instead
dom = bench["roofline"]["kernel"]
if dom in traffic["kernels"]:
    t = traffic["kernels"][dom]["hbm_bytes_per_launch"]
    bench["roofline"]["traffic"] = t
    bench["roofline"]["staging_TBps"] = t / (bench["roofline"]["avg_launch_ms"] * 1e-3) / 1e12
    json.dump(bench, open(os.path.join(dst, f"{tag}_{prec}{suffix}_bench.json"), "w"))

sq = per_kernel("pmc_SQ")
out = {"how": "one rocprofv3 --pmc pass (8 SQ counters + GRBM_GUI_ACTIVE) --kernel-trace over bench.py --steps 2 --warmup 1; means per dispatch. "
              "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); valu_busy = SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves; an MFMA counts "
              "with its issue slot only) * 4 / (1024 SIMDs) / the same cycles = share of SIMD time spent issuing vector ALU work; wait/active fractions are of SQ_WAVE_CYCLES",
       "kernels": {}}
for key, names in KERNELS.items():
    k0, name = pick(sq, names)
    if k0 is None:
        continue
    ks = [k0]
    d = {c: sum(v) / len(v) for c, v in sq[ks[0]].items()}
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    out["kernels"][key] = {"kernel": name, "dispatches": len(sq[ks[0]]["SQ_WAVE_CYCLES"]), "gpu_cycles_per_xcd": cyc,
                           "mfma_busy": d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
                           "valu_busy": d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc if "SQ_ACTIVE_INST_VALU" in d else None,
                           "wait_any": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], "wait_inst_any": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                           "active_inst_any": d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                           # cycles in which vector-ALU and matrix instructions execute together, as a share of the SIMD cycles (same normalisation as mfma_busy)
                           "valu_mfma_coexec": d["SQ_VALU_MFMA_COEXEC_CYCLES"] / 1024 / cyc if "SQ_VALU_MFMA_COEXEC_CYCLES" in d else None,
                           "wait_inst_lds": d["SQ_WAIT_INST_LDS"] / d["SQ_WAVE_CYCLES"] if "SQ_WAIT_INST_LDS" in d else None,
                           "lds_bank_conflict_cycles": d["SQ_LDS_BANK_CONFLICT"]}
json.dump(out, open(os.path.join(dst, f"{tag}_{prec}{suffix}_pmc_sq.json"), "w"), indent=1)
print(json.dumps({k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "kernel"} for k, v in out["kernels"].items()}, indent=1))
print({k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in traffic["kernels"].items()})
print(bench["value"], bench["ms_per_step"], bench["roofline"]["frac"], bench.get("cpu_baseline", {}).get("value"))
