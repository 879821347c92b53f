#!/usr/bin/env python3
"""Host time per graph-replayed step against the GPU time of the step, at the reference's default batch (1 024 x 500) and at the bench
batch: is the small-batch step bound by the host loop (ray ids, pinned record, replay launch) or by the GPU?
    python tools/host_overhead_probe.py        (on the GPU box)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from nerfca_amd import synthetic
    dev = torch.device("cuda", 0)
    for rays, samples in ((1024, 500), (8192, 192), (65536, 192)):
        args = bench.parse(["--no-extras", "--no-cpu-baseline", "--rays", str(rays), "--samples", str(samples)])
        data = synthetic.make_dataset(args.det, args.samples, dev, views=synthetic.TRAIN_VIEWS)
        tr = bench.make_trainer(args, "bf16", data, dev, 0, 1, False)
        it = 75000
        for _ in range(30):
            tr.step_graph(it); it += 1
        torch.cuda.synchronize()
        n = 300
        # (a) the loop as a training run issues it: no synchronisation inside
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step_graph(it); it += 1
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        # (b) the GPU time of one replay: events around a replay issued into an idle queue... and the host time of one call
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gpu = []
        host = []
        for _ in range(50):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            e0.record()
            tr.step_graph(it); it += 1
            e1.record()
            host.append(time.perf_counter() - h0)
            torch.cuda.synchronize()
            gpu.append(e0.elapsed_time(e1))
        gpu.sort(); host.sort()
        print(f"{rays:6d} rays x {samples} samples: {1e3 * t_all / n:7.3f} ms per step over {n} back-to-back steps (host had issued them after {1e3 * t_issue / n:7.3f} ms per step); "
              f"one step alone: GPU {gpu[len(gpu) // 2]:7.3f} ms, host call {1e3 * host[len(host) // 2]:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
