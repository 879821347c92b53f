#!/usr/bin/env python3
"""Run-to-run determinism of the training step ACROSS processes: prints, per step, the loss bits and at the end a hash of all
parameters, for the graph-replayed and the eager fused step of the bench configuration (or a smaller one).  Run it twice (two
processes) and diff the outputs; --poison first fills the free device memory with a NaN pattern and releases it, so that a kernel
that reads bytes nobody wrote sees garbage instead of whatever the last process left there.

    python tools/determinism_probe.py --mode graph --steps 12 [--poison] [--rays 65536 --samples 192 --prec bf16]
"""
import argparse
import hashlib
import os
import struct
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="graph", choices=["graph", "eager", "grads"])
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--dump", default="", help="graph mode: after the first step write per-MiB checksums of the forward store and the backward workspace to this file")
    ap.add_argument("--trace", action="store_true", help="graph mode: hash the step's inputs, gradient, parameters and optimiser state after every step")
    ap.add_argument("--poison-gb", type=float, default=200.0)
    args = ap.parse_args()
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    dev = torch.device("cuda", 0)
    if args.poison:
        n = int(args.poison_gb * (1 << 30)) // 4
        junk = torch.empty(n, dtype=torch.int32, device=dev)
        junk.fill_(0x7FC12345)          # a NaN as f32, two NaNs as bf16 pairs, large e4m3 / e5m2 bytes
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
    data = synthetic.make_dataset(args.det, args.samples, dev, views=synthetic.TRAIN_VIEWS[:2] if args.det < 256 else synthetic.TRAIN_VIEWS)
    h = hashlib.sha256()
    h.update(data.rays_train.cpu().numpy().tobytes())
    print("data", h.hexdigest()[:16], flush=True)
    torch.manual_seed(1)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    nerfca_amd.set_precision(args.prec, s, t)
    tr = CompositeTrainer(TrainConfig(depth_samples_per_ray_coarse=args.samples, img_sample_size=args.rays), s, t, data, dev, seed=0)
    kept = {}
    if args.dump:
        from nerfca_amd import fused as FU
        orig_alloc, orig_fwd = FU._alloc_workspace, FU.render_forward_raw

        def alloc(size_for_cap, dev_):
            w, n = orig_alloc(size_for_cap, dev_)
            kept.setdefault("work", []).append(w)
            return w, n

        def fwd(*a, **k):
            r = orig_fwd(*a, **k)
            kept.setdefault("store", []).append(r[3][6])
            return r
        FU._alloc_workspace, FU.render_forward_raw = alloc, fwd
    for it in range(args.steps):
        if args.dump and it == 1:
            break
        if args.mode == "grads":
            terms, gs, gd = tr.fused_gradients(75000)
            hh = hashlib.sha256(gs.cpu().numpy().tobytes() + gd.cpu().numpy().tobytes()).hexdigest()[:16]
            print(it, struct.pack(">d", float(terms[0])).hex(), hh, flush=True)
            continue
        out = tr.step_graph(75000 + it) if args.mode == "graph" else tr.step(75000 + it)
        print(it, struct.pack(">d", float(out[0])).hex(), f"{float(out[0]):.9e}", flush=True)
        if args.mode == "graph" and args.trace:
            torch.cuda.synchronize()
            hx = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
            print("   ids", hx(tr._ids_buf), "rec", hx(tr._rec_dev), "flat", hx(tr._graph_out["flat"]), "terms", hx(tr._graph_out["terms"]),
                  "params", hx(torch.cat([p.detach().flatten() for p in tr.params])), "m", hx(torch.cat(tr.adam.exp_avg)), "v", hx(torch.cat(tr.adam.exp_avg_sq)),
                  "count", int(tr.adam.step_count), flush=True)
            if it == 0:       # which parameters' gradients: [dynamic net | static net] in parameters() order
                flat, off = tr._graph_out["flat"], 0
                for tag, m in (("t", tr.t), ("s", tr.s)):
                    for name, prm in m.named_parameters():
                        g = flat[off:off + prm.numel()]
                        off += prm.numel()
                        print(f"      {tag}.{name:32s} {hx(g)} |g|max {float(g.abs().max()):.6e} sum {float(g.double().sum()):.9e}", flush=True)
    if args.dump:
        torch.cuda.synchronize()
        out = {}
        for k, lst in kept.items():
            t = lst[-1]                 # the capture's buffers (the warm-up's came first)
            if t is None:
                continue
            n = t.numel() // (1 << 20) * (1 << 20)
            out[k] = t[:n].view(torch.int32).view(-1, 1 << 18).to(torch.int64).sum(1).cpu()
            out[k + "_bytes"] = t.numel()
        torch.save(out, args.dump)
        torch.save(kept["work"][-1][: 80 << 20].cpu(), args.dump + ".slab")        # the split slabs sit at the start of the workspace
        print("dumped", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()}, flush=True)
    hp = hashlib.sha256(torch.cat([p.detach().flatten() for p in tr.params]).cpu().numpy().tobytes()).hexdigest()[:16]
    print("params", hp, flush=True)


if __name__ == "__main__":
    main()
