#!/usr/bin/env python3
"""gpurun_out/prof_wide/ (tools/wide_profile.sh) -> profiles/<tag>_wide_kernel_stats.csv and profiles/<tag>_wide_pmc.json: per general kernel
the mean launch time, HBM bytes per launch (FETCH_SIZE x 1024 x 2 -- gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md --
and WRITE_SIZE x 1024, as tools/summarize_profiles.py does) against the algorithmic bytes, and the SQ busy fractions.
usage: python tools/wide_profile_summary.py [tag=r06]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", "prof_wide")
dst = os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
from bench import source_sha


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return max(hits, key=os.path.getmtime)


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(find(f"{d}/**/*counter_collection.csv"))):
        if "nca_wide" in r["Kernel_Name"]:
            # the full-size hidden-layer launches only (grid x = 524 288 threads for the 262 144-row chunks of forward / dgrad)
            agg[(r["Kernel_Name"].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


with open(find("stats/**/*kernel_stats.csv")) as f, open(os.path.join(dst, f"{tag}_wide_kernel_stats.csv"), "w") as g:
    g.write("".join(l for l in f if "nca_" in l or l.startswith('"Name"')))
fetch, write, sq = counters("pmc_FETCH_SIZE"), counters("pmc_WRITE_SIZE"), counters("pmc_SQ")
rows, F = 262144, 256
alg = {"nca_wide_gemm<0>": (rows * F * 2 + F * F) * 4 + rows * F // 8, "nca_wide_gemm<1>": (rows * F * 2 + F * F) * 4 + rows * F // 8, "nca_wide_gemm<2>": (rows * F * 2) * 4}
out = {"source_sha": source_sha(), "how": "tools/wide_profile.sh; means over the dispatches of the LARGEST grid of each kernel (the 262 144-row hidden-layer launches)",
       "algorithmic_bytes_note": "forward: A rows x F read, C rows x F written, W, one mask bit per output written; dgrad: the same with the bits read; wgrad: both operands read once", "kernels": {}}
for kern in sorted({k[0] for k in fetch}):
    grids = [k for k in fetch if k[0] == kern]
    big = max(grids, key=lambda k: (len(fetch[k]["FETCH_SIZE"]) > 2, sum(fetch[k]["FETCH_SIZE"]) / len(fetch[k]["FETCH_SIZE"])))
    rd = sum(fetch[big]["FETCH_SIZE"]) / len(fetch[big]["FETCH_SIZE"]) * 1024 * 2
    wk = [k for k in write if k == big]
    wr = sum(write[big]["WRITE_SIZE"]) / len(write[big]["WRITE_SIZE"]) * 1024 if wk else None
    rec = {"grid": big[1], "dispatches": len(fetch[big]["FETCH_SIZE"]), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr}
    name = kern.replace("void ", "")
    for a, b in alg.items():
        if a in name:
            rec["algorithmic_bytes_per_launch"] = b
    if big in sq and "GRBM_GUI_ACTIVE" in sq[big]:
        d = {c: sum(v) / len(v) for c, v in sq[big].items()}
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        rec["mfma_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
        rec["valu_busy"] = d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
        rec["lds_bank_conflict_cycles"] = d.get("SQ_LDS_BANK_CONFLICT")
    out["kernels"][name] = rec
json.dump(out, open(os.path.join(dst, f"{tag}_wide_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
