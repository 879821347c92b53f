#!/usr/bin/env python3
"""PSNR at matched steps (BASELINE.json: "rays/sec ...; PSNR vs ref").

Trains the composite model on the synthetic data set of bench.py (40 training images of det^2, one held-out view) from the
same initial weights, ray batches and depth jitter in every arithmetic the library offers --

    f32             the parity mode (within 1e-5 of the reference's arithmetic per step, tests/test_hip_parity.py)
    bf16_store      bf16 MFMA operands everywhere, nothing in 8 bits: the forward leaves the BF16 store (layer inputs as bf16 fragments, masks, raw
                    outputs), the backward recomputes nothing and writes bf16 output gradients (NCA_OPT_STAGE_FP8 = 0; round 5)
    bf16_nostore    the same arithmetic constraints without any store: the backward recomputes the layers (NCA_OPT_STAGE_FP8 = 0, NCA_OPT_BF16_STORE = 0)
    bf16_fp8stage   bf16 MFMA operands, the forward store staged as e4m3 / e5m2 (NCA_OPT_STAGE_FP8 = 1)
    bf16            bf16 with the planner's own choice for this batch size (the library default: the 8-bit staged store)
    f32_kick<eps>   f32 from initial weights moved once by a relative eps N(0, 1) (e.g. f32_kick2e-3: the round-4 control)
    f32_bf16init    f32 from initial weights ROUNDED TO BF16 once (round to nearest even: at most 2^-9 relative, 1.1e-3 rms) -- the
                    smallest thing the bf16 mode does to the weights, done once, to the parity mode
    f32_resample    f32 from the SAME initial weights with another stream of ray batches and depth jitter (the trainer's seed + 7919):
                    what the mini-batch sampling alone moves the result by

-- and evaluates the held-out view every `--every` steps with CompositeTrainer.evaluate (MSE PSNR and the reference's own
test_psnr, run_composite.py:391).  One JSON line; run on the GPU box:

    python tools/psnr_run.py --steps 300 --every 100 > gpurun_out/psnr.json                                  # bench batch
    python tools/psnr_run.py --rays 1024 --samples 500 --steps 5000 --every 500 --variants f32,bf16_nostore,bf16_fp8stage
                                                                             # the reference's default batch (composite.txt:25,40)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {"f32": ("f32", None), "bf16": ("bf16", None), "bf16_store": ("bf16", {"stage_fp8": 0}), "bf16_nostore": ("bf16", {"stage_fp8": 0, "bf16_store": 0}),
            "bf16_fp8stage": ("bf16", {"stage_fp8": 1}),
            # f32 again from initial weights moved by 1e-6 (relative): the run-to-run spread of the parity mode itself, i.e. the
            # resolution of a PSNR comparison at this batch size
            "f32_perturbed": ("f32", None)}
# --perturb R: the relative size of that move (default 1e-6; 2e-3 is the size of a bf16 rounding of every weight)


def run(variant, args, dev, data, log=None, seed=0):
    import nerfca_amd
    from nerfca_amd import _capi, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    kick = None
    batch_seed = seed + (7919 if variant == "f32_resample" else 0)
    if variant in ("f32_resample", "f32_bf16init"):
        prec, stage = "f32", None
    elif variant.startswith("f32_kick"):          # f32_kick<eps>: f32 from initial weights moved once by a relative eps (several per run, unlike f32_perturbed)
        kick, (prec, stage) = float(variant[len("f32_kick"):]), ("f32", None)
    else:
        prec, stage = VARIANTS[variant]
    torch.manual_seed(1 + 1000 * seed)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    if variant == "f32_perturbed" or kick is not None:
        with torch.no_grad():
            g = torch.Generator(device=dev).manual_seed(12345 + seed)
            for m in (s, t):
                for prm in m.parameters():
                    prm.mul_(1.0 + (kick if kick is not None else args.perturb) * torch.randn(prm.shape, generator=g, device=dev))
    if variant == "f32_bf16init":
        with torch.no_grad():
            for m in (s, t):
                for prm in m.parameters():
                    prm.copy_(prm.to(torch.bfloat16).to(prm.dtype))
    nerfca_amd.set_precision(prec, s, t)
    # --cross-eval: a second pair of models in the OTHER arithmetic that takes over the trained weights at every evaluation -- separates
    # what an arithmetic costs the training from what it costs the rendering of the held-out view
    shadow = None
    if args.cross_eval:
        s2, t2 = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        nerfca_amd.set_precision("f32" if prec == "bf16" else "bf16", s2, t2)
        shadow = (s2, t2)
    # schedules compressed to the length of the run (the reference anneals over 150 k steps of 1 024 rays)
    cfg = TrainConfig(depth_samples_per_ray_coarse=args.samples, img_sample_size=args.rays, static_pos_enc_window_decay_steps=args.steps,
                      temp_pos_enc_window_decay_steps=args.steps, lr_decay_steps=args.steps)
    tr = CompositeTrainer(cfg, s, t, data, dev, seed=batch_seed, plan_opts=stage)
    tr.update_windows(0)
    curve = []

    def point(it):
        tr.update_windows(it)
        e = tr.evaluate(it)
        curve.append({"step": it, "psnr_mse_db": float(e["test_psnr_mse"]), "test_psnr_reference_def_db": float(e["test_psnr"]), "test_loss": float(e["test_loss"])})
        if shadow is not None:
            with torch.no_grad():
                for dst, src in zip(shadow, (tr.s, tr.t)):
                    for pd, ps in zip(dst.parameters(), src.parameters()):
                        pd.copy_(ps)
            own = (tr.s, tr.t)
            tr.s, tr.t = shadow
            try:
                tr.update_windows(it)
                e2 = tr.evaluate(it)
            finally:
                tr.s, tr.t = own
            curve[-1]["other_arithmetic_psnr_mse_db"] = float(e2["test_psnr_mse"])
            curve[-1]["other_arithmetic_test_psnr_reference_def_db"] = float(e2["test_psnr"])
        if log:
            print(f"[psnr_run] {variant} step {it}: {curve[-1]['psnr_mse_db']:.3f} dB", file=log, flush=True)

    point(0)
    step = tr.step_graph if args.graph else tr.step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fp8_seen = [None]
    for it in range(args.steps):
        loss, _, _ = step(it)
        if it == 0:
            fp8_seen[0] = bool(tr.plan().get("stage_fp8"))       # what THIS trainer's planner did (8-bit staged store or none)
        if (it + 1) % args.every == 0:
            point(it + 1)
    torch.cuda.synchronize()
    if args.jsonl:          # every finished run is on disk at once: a run that is cut off keeps what it has
        with open(args.jsonl, "a") as f:
            f.write(json.dumps({"variant": variant, "seed": seed, "label": args.label, "library": _capi.build_info(), "rays": args.rays, "samples": args.samples,
                                "steps": args.steps, "perturb": kick if kick is not None else (args.perturb if variant == "f32_perturbed" else None), "curve": curve}) + "\n")
    return {"curve": curve, "final_train_loss": float(loss), "wall_s_incl_eval": time.perf_counter() - t0, "seed": seed,
            "stage_fp8_in_effect": None if prec == "f32" else fp8_seen[0]}


def gap_statistics(runs, names, tail=1):
    """Per variant: mean and standard deviation over the seeds of (variant - f32) in dB, at the final evaluation and averaged over
    the last `tail` evaluations, for both PSNR definitions; `f32_perturbed` (when run) is the resolution of the comparison."""
    import statistics as st
    out = {}
    for key in ("psnr_mse_db", "test_psnr_reference_def_db"):
        for v in names:
            if v == "f32" or "f32" not in runs:
                continue
            fin = [r["curve"][-1][key] - f["curve"][-1][key] for r, f in zip(runs[v], runs["f32"])]
            avg = [sum(c[key] for c in r["curve"][-tail:]) / tail - sum(c[key] for c in f["curve"][-tail:]) / tail for r, f in zip(runs[v], runs["f32"])]
            out.setdefault(v, {})[key] = {"final_gap_per_seed": fin, "final_gap_mean": st.mean(fin), "final_gap_sd": st.stdev(fin) if len(fin) > 1 else None,
                                          f"last{tail}_gap_per_seed": avg, f"last{tail}_gap_mean": st.mean(avg), f"last{tail}_gap_sd": st.stdev(avg) if len(avg) > 1 else None}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tail", type=int, default=3, help="evaluations averaged for the smoothed gap (multi-seed records)")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--variants", default="f32,bf16")
    ap.add_argument("--seeds", default="0", help="comma-separated trainer seeds (ray batches, depth jitter, initial weights); > 1: the record keeps every "
                    "run and the mean / standard deviation of the final gaps over the seeds")
    ap.add_argument("--perturb", type=float, default=1e-6, help="relative size of f32_perturbed's move of the initial weights")
    ap.add_argument("--cross-eval", action="store_true", help="also evaluate the trained weights in the other arithmetic (f32 <-> bf16) at every evaluation")
    ap.add_argument("--jsonl", default="", help="append one line per finished run to this file")
    ap.add_argument("--label", default="", help="free text kept in the record (e.g. which ablation library NERFCA_LIB points at)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a HIP graph (library Adam) instead of the eager fused step")
    args = ap.parse_args()
    from nerfca_amd import _capi, synthetic
    _capi.lib()
    dev = torch.device("cuda", 0)
    data = synthetic.make_dataset(args.det, args.samples, dev, views=synthetic.TRAIN_VIEWS)
    out = {"config": f"{args.det}^2 detector x {args.samples} samples/ray, {args.rays} rays/step, {args.steps} steps, 4 views x 10 phases + 1 held-out view, "
                     f"synthetic phantom, {'HIP-graph step' if args.graph else 'eager fused step'}",
           "library": _capi.build_info(), "label": args.label, "perturb": args.perturb}
    names = [v for v in args.variants.split(",") if v]
    seeds = [int(x) for x in args.seeds.split(",") if x != ""]
    if len(seeds) > 1:
        runs = {v: [run(v, args, dev, data, log=sys.stderr, seed=sd) for sd in seeds] for v in names}
        out["seeds"] = seeds
        out["runs"] = runs
        tail = max(1, min(args.tail, len(runs[names[0]][0]["curve"])))
        out["gaps"] = gap_statistics(runs, names, tail)
        print(json.dumps(out))
        return
    for v in names:
        out[v] = run(v, args, dev, data, log=sys.stderr, seed=seeds[0])
    if "f32" in out:
        ref = out["f32"]["curve"][-1]
        out["final_gap_vs_f32_db"] = {v: {"psnr_mse": out[v]["curve"][-1]["psnr_mse_db"] - ref["psnr_mse_db"],
                                          "test_psnr_reference_def": out[v]["curve"][-1]["test_psnr_reference_def_db"] - ref["test_psnr_reference_def_db"]}
                                      for v in names if v != "f32"}
        if "bf16" in out:      # (key of the round-1 / round-2 records)
            out["final_psnr_gap_db"] = ref["psnr_mse_db"] - out["bf16"]["curve"][-1]["psnr_mse_db"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
