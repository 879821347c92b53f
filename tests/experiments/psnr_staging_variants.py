#!/usr/bin/env python3
"""Which rounding of the fp8-staged backward costs held-out PSNR at SMALL batches (bench.py's `psnr` record: 64^2 detector, 256 rays
per step, 100 steps -- 250 x fewer samples per step than the bench configuration to average the staging noise over)?

(Lives under tests/: it drives the CPU oracle, which only tests, smoke() and the bench baselines may import.)  Runs ON the GPU box (the data set and the evaluation use the HIP f32 renderer, the training runs on the host cores): the CPU oracle
trains the same nets from the same weights, ray ids and jitter under several arithmetic models of the last F-wide layer + output
layer (oracle._StagedTail is replaced per variant; everything else is the emulation the GPU tests pin):

  f32          the reference's arithmetic
  bf16         bf16 operands, nothing staged in 8 bits
  sums         what the mode-5 kernels do: the block for the weight-gradient kernel is e5m2(bf16(g) relu'), dWo from the sums
  with_wo      the block carries Wo[f] (e5m2(bf16(Wo g) relu'), what modes 3 / 4 staged), dWo = g^T bf16(h) in f32 (needs the layer's
               output in the backward: the recompute this round removed)
  sums_exactg  as `sums`, but with g itself kept at bf16 precision in the block (no 8-bit rounding of the last block)
  sums_e4m3    as `sums`, the last block as e4m3 (3 mantissa bits instead of 2)
  sums_decor   as `sums`, the block multiplied by a per-feature constant c[f] in [1, 2) before the e5m2 rounding (divided out of the sums)

    python tests/experiments/psnr_staging_variants.py > profiles/r02_psnr_staging_variants.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerfca_amd                                                    # noqa: E402
from nerfca_amd import synthetic                                     # noqa: E402
from nerfca_amd.model.CPPN import CPPN                               # noqa: E402
from nerfca_amd.model.Temporal import Temporal                       # noqa: E402
from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig   # noqa: E402
from oracle import nerfca_oracle as O                                # noqa: E402


def make_tail(variant):
    class Tail(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, W, b, Wo, bo, state, d8, h8):
            xq, Wq = x.to(torch.bfloat16).to(x.dtype), W.to(torch.bfloat16).to(W.dtype)
            z = xq @ Wq.t() + b
            ctx.save_for_backward(xq, Wq, b, Wo, z)
            ctx.state, ctx.d8, ctx.h8 = state, d8, h8
            return torch.relu(z) @ Wo.t() + bo

        @staticmethod
        def backward(ctx, g):
            xq, Wq, b, Wo, z = ctx.saved_tensors
            q8, sc = O._q8, ctx.state["scale"]
            h4 = torch.relu(z)
            on = (h4.to(torch.bfloat16) > 0).to(g.dtype)
            chain = (g @ Wo).to(torch.bfloat16).to(g.dtype) * on
            hh = q8(xq * 2.0 ** O.H8_LOG2, ctx.h8) / 2.0 ** O.H8_LOG2
            wo = Wo.reshape(-1)
            if variant == "with_wo":
                dd = q8(chain * sc, ctx.d8) / sc
                return chain @ Wq, dd.t() @ hh, dd.sum(0), g.t() @ h4.to(torch.bfloat16).to(g.dtype), g.sum(0), None, None, None
            gq = g.to(torch.bfloat16).to(g.dtype) * on
            if variant == "sums":
                gq = q8(gq * sc, ctx.d8) / sc
            elif variant == "sums_e4m3":
                gq = q8(gq * sc, "e4m3") / sc
            elif variant == "sums_decor":
                c = 1.0 + torch.arange(gq.shape[1], dtype=g.dtype) * (0.9921875 / gq.shape[1])
                gq = q8((g * c).to(torch.bfloat16).to(g.dtype) * on * sc, ctx.d8) / sc / c
            S, s = gq.t() @ hh, gq.sum(0)
            dWo = ((Wq * S).sum(1) + b * s).reshape(Wo.shape)
            return chain @ Wq, wo[:, None] * S, wo * s, dWo, g.sum(0), None, None, None
    return Tail


def main():
    dev = torch.device("cuda:0")
    S, R, steps, det = 192, 256, 100, 64
    data = synthetic.make_dataset(det, S, dev, views=synthetic.TRAIN_VIEWS, n_phases=10, F=64)
    cfg = TrainConfig(depth_samples_per_ray_coarse=S, img_sample_size=R, static_pos_enc_window_decay_steps=steps,
                      temp_pos_enc_window_decay_steps=steps, lr_decay_steps=steps)

    def fresh():
        torch.manual_seed(1)
        sdef, tdef = synthetic.net_definitions(dev)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        nerfca_amd.set_precision("f32", s, t)
        return CompositeTrainer(cfg, s, t, data, dev, seed=0)

    table, phases = data.rays_train.cpu(), data.phases_train.cpu()
    I0 = torch.full((R,), float(data.geo["max_pixel_value"]))
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))          # (the box shows more CPUs than the job may use)
    out = {}
    orig = O._StagedTail
    for name in ("f32", "bf16", "sums", "with_wo", "sums_exactg", "sums_e4m3", "sums_decor"):
        tr = fresh()
        kw = {} if name == "f32" else dict(emulate_bf16=True)
        if name not in ("f32", "bf16"):
            kw.update(emulate_fp8_stage=S)
            O._StagedTail = make_tail(name)
        ss, sd = O.NetSpec(num_filters=128, **kw), O.NetSpec(num_filters=128, num_time_dim=8, **kw)
        ps = {k: v.detach().cpu().clone() for k, v in tr.s.state_dict().items()}
        pd = {k: v.detach().cpu().clone() for k, v in tr.t.state_dict().items()}
        ot = O.OracleTrainer(ps, ss, pd, sd, lr=cfg.lr, lr_end_factor=cfg.lr_end_factor, lr_decay_steps=steps, window_decay_steps=steps)
        z0 = tr.depth.cpu()
        t0 = time.perf_counter()
        for it in range(steps):
            ids = tr.draw_ray_ids_device(it).cpu()
            rays, ph = table.index_select(0, ids), phases.index_select(0, ids)
            zj = O.stratified_depths(z0, tr.draw_jitter(it).cpu())
            ot.step(it, rays[:, 0, :], rays[:, 1, :], ph[:, None].repeat(1, S), I0, zj, rays[:, 2, 0], rays[:, 3, 0])
            if it % 25 == 24:
                print(f"  {name}: step {it + 1}, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
        O._StagedTail = orig
        tr.s.load_state_dict({k: v.detach() for k, v in ot.ps.items()})
        tr.t.load_state_dict({k: v.detach() for k, v in ot.pd.items()})
        tr.update_windows(steps)
        e = tr.evaluate(steps)
        out[name] = {"psnr_mse_db": float(e["test_psnr_mse"]), "test_psnr_db": float(e["test_psnr"]), "wall_s": time.perf_counter() - t0}
        print(name, out[name], file=sys.stderr, flush=True)
    for k in list(out):
        out[k]["gap_vs_f32_db"] = out[k]["psnr_mse_db"] - out["f32"]["psnr_mse_db"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
