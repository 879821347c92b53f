#!/usr/bin/env python3
"""What the step's ONE collective costs on one GPU: the graph-replayed step without a process group against the same step under a one-rank
RCCL group (the all-reduce of the flat gradient captured INTO the step graph, round 6; NERFCA_GRAPH_COLLECTIVE=0: two graph segments with
a host-issued collective between them, round 5's structure), at a rank's share of the global batch for N = 1 and N = 8.
    python tools/collective_overhead.py        (on the GPU box)"""
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench  # noqa: E402


def main():
    import torch.distributed as dist
    from nerfca_amd import synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    args = bench.parse(["--no-extras", "--no-cpu-baseline"])
    data = synthetic.make_dataset(args.det, args.samples, dev, views=synthetic.TRAIN_VIEWS)
    base = {r: bench.quick_step_ms(args, "bf16", data, dev, 0, 1, False, r, steps=60, warmup=10) for r in (65536, 8192)}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        for mode in ("1", "0"):
            os.environ["NERFCA_GRAPH_COLLECTIVE"] = mode
            for r in (65536, 8192):
                ms = bench.quick_step_ms(args, "bf16", data, dev, 0, 1, True, r, steps=60, warmup=10)
                print(f"{r:6d} rays: no group {base[r]:8.4f} ms, one-rank RCCL group ({'collective captured in the graph' if mode == '1' else 'two graphs + host-issued collective'}) "
                      f"{ms:8.4f} ms: +{ms - base[r]:.4f} ms", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
