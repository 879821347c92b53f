#!/usr/bin/env python3
"""Time the bench step's weight-gradient launch for several values of NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT (the share of sample splits the
jobs that rebuild their output-gradient block get: the launch is one round of one-wave jobs, its slowest wave is the launch) and print
the fastest on this box.  The library does not do this by itself at first use: the splits fix the summation order of the weight gradient,
so a calibrated value would make the bits of a run depend on a timing.  Run on the GPU box:
    python tools/calibrate_wgrad.py [--rays 65536 --samples 192]        ->  export NCA_WGRAD_W=<best>   (or _capi.set_option)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--values", default="100,108,112,115,118,122,130")
    args = ap.parse_args()
    best = None
    for w in [int(x) for x in args.values.split(",")]:
        env = dict(os.environ, NCA_WGRAD_W=str(w))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rays", str(args.rays), "--samples", str(args.samples), "--steps", "10", "--warmup", "3",
                              "--no-extras", "--no-cpu-baseline", "--kernel-steps", "8"], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            print(f"W = {w}: failed\n{out.stderr[-400:]}")
            continue
        b = json.loads(line[-1])
        k = b["roofline"]["all_kernels"]["bwd_wgrad"]["avg_ms"]
        plan = b["config"]["plan"]["wgrad"]
        print(f"W = {w:3d} %: weight gradient {k:.3f} ms per launch, step {b['ms_per_step']:.3f} ms   (splits {plan['splits']} / {plan['splits_rebuild_jobs']})", flush=True)
        if best is None or k < best[1]:
            best = (w, k)
    if best:
        print(f"fastest: NCA_WGRAD_W={best[0]}  ({best[1]:.3f} ms)")


if __name__ == "__main__":
    main()
