#!/usr/bin/env python3
"""BASELINE configs[4]: a hyper-parameter sweep as independent single-GPU jobs across one node (the reference runs
train/sweep-composite.yaml through a wandb agent per GPU; there is no communication between the jobs).  One child process
per GPU, each pinned with HIP_VISIBLE_DEVICES before it touches the GPU, each running `bench.py --gpus 1` on its own grid
point; prints the N bench lines (one JSON object per line, tagged with the GPU and the grid point).

    python tools/sweep_launch.py --gpus 8 --grid rays=16384,32768,65536 --grid samples=128,192 -- --steps 4 --warmup 1 --no-extras --no-cpu-baseline
"""
import argparse
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--grid", action="append", default=[], help="name=v1,v2,... : a bench.py option swept over the values (repeatable)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="options passed to every bench.py job (after --)")
    args = ap.parse_args()
    rest = [a for a in args.rest if a != "--"]
    axes = [(g.split("=")[0], g.split("=")[1].split(",")) for g in args.grid] or [("steps", ["4"])]
    points = [dict(zip([n for n, _ in axes], vals)) for vals in itertools.product(*[v for _, v in axes])]
    jobs, done = [], []
    for i, pt in enumerate(points):
        gpu = i % args.gpus
        if len(jobs) == args.gpus:                      # one job per GPU at a time
            done += [(j, j[0].communicate()) for j in jobs]
            jobs = []
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(gpu))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + [x for k, v in pt.items() for x in (f"--{k}", v)] + rest
        jobs.append((subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), gpu, pt))
    done += [(j, j[0].communicate()) for j in jobs]
    rc = 0
    for (proc, gpu, pt), (out, err) in done:
        lines = [l for l in out.splitlines() if l.startswith("{")]
        if proc.returncode != 0 or not lines:
            rc = 1
            print(json.dumps({"gpu": gpu, "point": pt, "error": err[-400:]}))
            continue
        rec = json.loads(lines[-1])
        print(json.dumps({"gpu": gpu, "point": pt, "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "final_loss": rec.get("final_loss")}))
    sys.exit(rc)


if __name__ == "__main__":
    main()
