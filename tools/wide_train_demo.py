#!/usr/bin/env python3
"""End-to-end training on the GENERAL kernels (nets beyond 128 units): the composite model with nets of 128 units (fused f32
kernels) and of 256 units (general f32 kernels) trained on bench.py's synthetic data set from the same seed through the same graph-replayed step
(forward + all losses + backward + Adam), held-out PSNR every --every steps and the time per step.  One JSON line per width.

    python tools/wide_train_demo.py [--rays 8192] [--samples 192] [--steps 400] [--every 200] [--widths 128,256]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--every", type=int, default=200)
    ap.add_argument("--widths", default="128,256")
    a = ap.parse_args()
    from nerfca_amd import _capi, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    dev = torch.device("cuda:0")
    data = synthetic.make_dataset(a.det, a.samples, dev, views=synthetic.TRAIN_VIEWS)
    for F in [int(w) for w in a.widths.split(",")]:
        torch.manual_seed(1)
        sdef, tdef = synthetic.net_definitions(dev, F=F)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        cfg = TrainConfig(depth_samples_per_ray_coarse=a.samples, img_sample_size=a.rays, static_pos_enc_window_decay_steps=a.steps,
                          temp_pos_enc_window_decay_steps=a.steps, lr_decay_steps=a.steps)
        tr = CompositeTrainer(cfg, s, t, data, dev, seed=0)
        curve = []

        def point(it):
            tr.update_windows(it)
            e = tr.evaluate(it)
            curve.append({"step": it, "psnr_mse_db": round(float(e["test_psnr_mse"]), 3), "test_loss": float(e["test_loss"])})

        point(0)
        step_s = 0.0
        first = last = None
        for it in range(a.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss, _, _ = tr.step_graph(it)
            torch.cuda.synchronize()
            if it >= 5:
                step_s += time.perf_counter() - t0
            first = float(loss) if first is None else first
            last = float(loss)
            if (it + 1) % a.every == 0:
                point(it + 1)
        print(json.dumps({"num_filters": F, "kernel_width": s._binding.net.F, "kernels": "general" if _capi.net_is_general(s._binding.net) else "fused", "dtype": "f32",
                          "rays": a.rays, "samples": a.samples, "steps": a.steps, "ms_per_step": round(step_s / (a.steps - 5) * 1e3, 3),
                          "rays_per_s": round(a.rays / (step_s / (a.steps - 5))), "train_loss_first_last": [first, last], "held_out": curve,
                          "fwd_store_format": _capi.last_plan()["fwd_store_format"]}), flush=True)


if __name__ == "__main__":
    main()
