#!/usr/bin/env python3
"""Is the step bound by a power budget averaged over many steps?  The bench's graph-replayed bf16 step, timed with HIP events around
each replay, run back to back and with the host sleeping between steps (the chip idles): if a step runs faster after an idle gap,
the limit is an average over a window longer than a step; if not, it acts within a kernel.
    python tools/duty_cycle_probe.py > gpurun_out/duty_cycle.txt          (on the GPU box)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse(["--no-extras", "--no-cpu-baseline"])
    from nerfca_amd import synthetic
    dev = torch.device("cuda", 0)
    data = synthetic.make_dataset(args.det, args.samples, dev, views=synthetic.TRAIN_VIEWS[:args.views])
    tr = bench.make_trainer(args, "bf16", data, dev, 0, 1, False)
    it = 75000
    for _ in range(20):
        tr.step_graph(it); it += 1
    torch.cuda.synchronize()
    print("graph-replayed bf16 step (65 536 rays x 192 samples), HIP events around every step; median / min / max over 40 steps")
    for gap_ms in (0, 2, 5, 15, 50, 200):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for a, b in ev:
            if gap_ms:
                torch.cuda.synchronize()
                time.sleep(gap_ms * 1e-3)
            a.record()
            tr.step_graph(it); it += 1
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"  idle gap {gap_ms:4d} ms between steps: {ts[len(ts) // 2]:7.3f} ms per step   ({ts[0]:.3f} .. {ts[-1]:.3f})", flush=True)


if __name__ == "__main__":
    main()
