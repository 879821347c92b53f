#!/usr/bin/env python3
"""PSNR at matched steps on the BENCH configuration (BASELINE.json: "rays/sec ...; PSNR vs ref").

Trains the composite model on the synthetic 256^2 x 192-sample data set of bench.py (40 training images, one held-out
view) from the same initial weights, ray batches and depth jitter in f32 (the mode that is within 1e-5 of the
reference's arithmetic per step, tests/test_hip_parity.py) and in bf16 (the throughput mode), and evaluates the
held-out view every `--every` steps with CompositeTrainer.evaluate (MSE PSNR and the reference's own test_psnr,
run_composite.py:391).  One JSON line; run on the GPU box:

    python tools/psnr_run.py --steps 300 --every 100 > gpurun_out/psnr.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(prec, args, dev, data):
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    torch.manual_seed(1)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    nerfca_amd.set_precision(prec, s, t)
    # schedules compressed to the length of the run (the reference anneals over 150 k steps of 1 024 rays)
    cfg = TrainConfig(depth_samples_per_ray_coarse=args.samples, img_sample_size=args.rays, static_pos_enc_window_decay_steps=args.steps,
                      temp_pos_enc_window_decay_steps=args.steps, lr_decay_steps=args.steps)
    tr = CompositeTrainer(cfg, s, t, data, dev, seed=0)
    tr.update_windows(0)
    curve = []

    def point(it):
        e = tr.evaluate(it)
        curve.append({"step": it, "psnr_mse_db": float(e["test_psnr_mse"]), "test_psnr_reference_def_db": float(e["test_psnr"]), "test_loss": float(e["test_loss"])})

    point(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss, _, _ = tr.step(it)
        if (it + 1) % args.every == 0:
            point(it + 1)
    torch.cuda.synchronize()
    return {"curve": curve, "final_train_loss": float(loss), "wall_s_incl_eval": time.perf_counter() - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--samples", type=int, default=192)
    args = ap.parse_args()
    from nerfca_amd import _capi, synthetic
    _capi.lib()
    dev = torch.device("cuda", 0)
    data = synthetic.make_dataset(args.det, args.samples, dev, views=synthetic.TRAIN_VIEWS)
    out = {"config": f"{args.det}^2 detector x {args.samples} samples/ray, {args.rays} rays/step, {args.steps} steps, 4 views x 10 phases + 1 held-out view, synthetic phantom",
           "f32": run("f32", args, dev, data), "bf16": run("bf16", args, dev, data)}
    out["final_psnr_gap_db"] = out["f32"]["curve"][-1]["psnr_mse_db"] - out["bf16"]["curve"][-1]["psnr_mse_db"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
