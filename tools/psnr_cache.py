#!/usr/bin/env python3
"""Builds tests/golden/psnr_f32_controls.json -- the f32 CONTROL runs of the PSNR gates (tests/test_psnr_gates.py) -- from the JSON lines
tools/psnr_run.py --jsonl wrote on the GPU box.  The f32 parity mode is bit-reproducible (same library, same seeds -> the same
curve to the last digit, run to run and box to box: profiles/r04_psnr_bench_batch_seed_table.json and
profiles/r05_psnr_f32_controls.jsonl hold the same f32 values a round apart), so its end points are data: the gates compare
the bf16 mode's LIVE runs with them and re-run ONE cached control live per session as a spot check of the cache itself.

    python3 tools/psnr_cache.py profiles/r05_psnr_f32_controls.jsonl [more.jsonl ...]

Entry key: "<rays>x<samples>x<steps>|<variant>|<seed>" -> {"psnr_mse_db", "test_psnr_reference_def_db"} of the last evaluation.
`f32_sources_sha` names the sources the runs were taken on -- EVERY kernel source and header of the library (the planner in nca_api.hip fixes
the split counts, hence the summation order; the loss, Adam, sampler and compositing kernels are in other translation units than the f32
fused kernels) and the Python that drives a run (trainer.py, fused.py, schedules.py, synthetic.py, psnr_run.py): comments and layout of the
C sources are ignored.  A test run on other sources FAILS (tests/test_psnr_gates.py, tests/test_host_cpu.py): re-take the controls."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-ca_amd", "csrc")
PY_SOURCES = ("nerf-ca_amd/train/trainer.py", "nerf-ca_amd/fused.py", "nerf-ca_amd/schedules.py", "nerf-ca_amd/synthetic.py", "tools/psnr_run.py")


def f32_source_list():
    return sorted(n for n in os.listdir(CSRC) if n.endswith((".hip", ".hpp", ".inc"))) + list(PY_SOURCES)
OUT = os.path.join(ROOT, "tests", "golden", "psnr_f32_controls.json")


def f32_sources_sha():
    """Hash of the sources that determine the f32 trajectory (C sources: comment- and layout-insensitive; Python: whitespace-normalised)."""
    h = hashlib.sha256()
    for name in f32_source_list():
        path = os.path.join(ROOT, name) if "/" in name else os.path.join(CSRC, name)
        text = open(path, encoding="utf-8", errors="replace").read()
        if not name.endswith(".py"):
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
        h.update(name.encode())
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def key(rays, samples, steps, variant, seed):
    return f"{rays}x{samples}x{steps}|{variant}|{seed}"


def main():
    entries, sources = {}, []
    for path in sys.argv[1:]:
        sources.append(os.path.relpath(path, ROOT))
        for line in open(path):
            r = json.loads(line)
            if not r["variant"].startswith("f32"):
                continue
            last = r["curve"][-1]
            if last["step"] != r["steps"]:
                continue
            entries[key(r["rays"], r["samples"], r["steps"], r["variant"], r["seed"])] = {
                "psnr_mse_db": last["psnr_mse_db"], "test_psnr_reference_def_db": last["test_psnr_reference_def_db"]}
    out = {"what": "end points of the f32 control runs of tests/test_psnr_gates.py (tools/psnr_run.py --graph; key = rays x samples x steps | variant | seed)",
           "f32_sources_sha": f32_sources_sha(), "f32_sources": f32_source_list(), "from": sources, "entries": dict(sorted(entries.items()))}
    json.dump(out, open(OUT, "w"), indent=1)
    print(f"{len(entries)} entries -> {os.path.relpath(OUT, ROOT)} (sources {out['f32_sources_sha']})")


if __name__ == "__main__":
    main()
