#!/usr/bin/env python3
"""CPU experiment behind DESIGN.md section 4.1 "fp8 staging": does training survive when the layer inputs (H) and the
output gradients (D) that cross HBM from the fused kernels to the weight-gradient kernel are staged as 8-bit floats?

It trains the composite model with CompositeTrainer on a small synthetic phantom (torch CPU ops, injected renderer) under
several arithmetic models of the MLP and reports the held-out PSNR of each:

  f32       plain f32 (the reference's arithmetic)
  bf16      what the bf16 HIP path does today: MFMA operands (layer inputs, weights, back-propagated deltas) rounded to
            bf16, f32 accumulation, f32 master weights
  fp8c+X/Y  (round 3: would fp8 CONTRACTIONS survive?)  like bf16+X/Y, but the FORWARD contractions of every layer take e4m3
            operands: the layer inputs x 2^h_log2 (the bytes the store already holds) and the weights scaled per layer by a power
            of two that brings max |W| into [128, 256); f32 accumulation; the backward as in bf16+X/Y
  fp8cd+X/Y the same with the dgrad contraction on 8-bit operands as well: deltas as e5m2 (scaled per tile), transposed weights e4m3
  bf16+X/Y  the same, but the weight gradient of every layer whose operands are staged through HBM is formed from
            D rounded to format X and H rounded to format Y (e4m3 / e5m2).  D is pre-scaled per 64-sample tile of a
            ray by a power of two taken from max |d loss / d raw| of the tile (the dgrad chain is linear in it, so the
            kernel scales once at the source); H uses a fixed power of two.  The last hidden layer's weight gradient
            stays bf16 (it is accumulated on chip, nothing is staged), as do bias gradients of that layer.

This file is a design experiment, not product code and not a test oracle; it shares no code with either.
    python tools/fp8_stage_emulation.py --steps 300 > profiles/r02_fp8_stage_emulation.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2
FMT = {"e4m3": (E4M3, 448.0), "e5m2": (E5M2, 57344.0)}


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def q8(x, fmt):
    dt, top = FMT[fmt]
    return x.clamp(-top, top).to(dt).to(torch.float32)


class Arith:
    """Arithmetic model of one run."""

    def __init__(self, name):
        self.name = name
        self.bf16 = name != "f32"
        self.chain8 = name.startswith("fp8c")            # forward contractions on e4m3 operands
        self.dgrad8 = name.startswith("fp8cd")           # dgrad contraction on e5m2 x e4m3 operands
        self.d_fmt = self.h_fmt = None
        self.d_log2, self.h_log2 = 4, 3
        if "+" in name:
            spec = name.split("+")[1]
            self.d_fmt, self.h_fmt = spec.split("/")
        self.tile_scale = None        # [N,1] power-of-two factor of each sample's tile, set by the output layer's backward
        self.S = None


class _Linear(torch.autograd.Function):
    """y = x W^T + b with the run's operand rounding; `staged`: the weight gradient of this layer is formed by the
    weight-gradient kernel from operands that crossed HBM (as opposed to the on-chip layer)."""

    @staticmethod
    def forward(ctx, x, W, b, ar: Arith, staged: bool, is_out: bool):
        if ar.chain8 and not is_out:
            sw = torch.exp2(7.0 - torch.floor(torch.log2(W.detach().abs().max().clamp(min=1e-30))))       # max |W| sw in [128, 256)
            hs = 2.0 ** ar.h_log2
            x, Wq = q8(bf(x) * hs, "e4m3") / hs, q8(W * sw, "e4m3") / sw
        elif ar.bf16 and not is_out:
            x, Wq = bf(x), bf(W)
        else:
            Wq = W
        ctx.save_for_backward(x, Wq)
        ctx.ar, ctx.staged, ctx.is_out = ar, staged, is_out
        return x @ Wq.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, Wq = ctx.saved_tensors
        ar = ctx.ar
        if not ar.bf16:
            return dy @ Wq, dy.t() @ x, dy.sum(0), None, None, None
        if ctx.is_out:
            # output layer: f32 VALU arithmetic in the kernel (dWo = sum g H, D_last = relu' Wo g)
            if ar.d_fmt:
                S = ar.S
                g = dy.reshape(-1, S, 1)
                nt = (S + 63) // 64
                pad = nt * 64 - S
                gp = torch.nn.functional.pad(g, (0, 0, 0, pad)).reshape(g.shape[0], nt, 64)
                amax = gp.abs().amax(dim=-1, keepdim=True).clamp(min=1e-30)
                # power of two that brings the tile's largest |g| into [2^d_log2, 2^(d_log2+1))
                sc = torch.exp2(ar.d_log2 - torch.floor(torch.log2(amax)))
                ar.tile_scale = sc.expand(-1, -1, 64).reshape(g.shape[0], nt * 64)[:, :S].reshape(-1, 1)
            return dy @ Wq, dy.t() @ x, dy.sum(0), None, None, None
        dq = bf(dy)                                  # the deltas are packed to bf16 as the next dgrad's B operand
        if ar.dgrad8 and ar.tile_scale is not None:
            s = ar.tile_scale
            dx = (q8(dy * s, "e5m2") / s) @ Wq       # (Wq is the e4m3 image of the forward)
        else:
            dx = dq @ Wq
        if ctx.staged and ar.d_fmt:
            s = ar.tile_scale
            d8 = q8(dy * s, ar.d_fmt) / s            # converted from the f32 accumulators, scaled by the tile's power of two
            h8 = q8(x * (2.0 ** ar.h_log2), ar.h_fmt) / (2.0 ** ar.h_log2)
            return dx, d8.t() @ h8, d8.sum(0), None, None, None
        return dx, dq.t() @ x, dq.sum(0), None, None, None


def encode(p, L, window):
    if L <= 0:
        return p
    scales = 2.0 ** torch.arange(0, L, device=p.device)
    xb = p[..., None, :] * scales[:, None]
    feat = torch.sin(torch.stack([xb, xb + 0.5 * torch.pi], dim=-2))
    if window is not None:
        feat = window.float().to(p.device)[..., None, None] * feat
    return torch.cat([p, feat.reshape(p.shape[0], -1)], dim=-1)


def make_render(ar: Arith, h_in_fp8: bool):
    def mlp(m, feats):
        prm = dict(m.named_parameters())
        n = m.num_early_layers + 1
        h = feats
        for i in range(n):
            # staged = every layer but the last hidden one (its dW is accumulated on chip); layer 0's H is the input block
            staged = i < n - 1
            h = torch.relu(_Linear.apply(h, prm[f"early_pts_layers.{2 * i}.weight"], prm[f"early_pts_layers.{2 * i}.bias"], ar, staged, False))
        return _Linear.apply(h, prm["output_linear.0.weight"], prm["output_linear.0.bias"], ar, False, True)

    def window(m):
        return m._band_window() if m.use_pos_enc in ("free_windowed", "nerfies_windowed") else None

    def render(s, t, o, d, ph, I0, z, dists, act="softplus", single=False, scale=1e-2):
        R, S = o.shape[0], z.shape[-1]
        ar.S = S
        pts = (o[:, None, :] + d[:, None, :] * z[:, None]).reshape(-1, 3).float()
        fs = encode(pts, s.pos_enc_basis, window(s))
        raw_s = mlp(s, fs).reshape(R, S)
        lat = t.time_latents[ph.reshape(R, -1)[:, :1].repeat(1, S).flatten().long()]
        raw_d = mlp(t, torch.cat([encode(pts, t.pos_enc_basis, window(t)), lat], -1)).reshape(R, S)
        a = torch.nn.functional.softplus(raw_s) * scale
        b = torch.nn.functional.softplus(raw_d) * scale
        return I0 - ((a + b) * dists).sum(-1), a, b

    return render


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--det", type=int, default=32)
    ap.add_argument("--samples", type=int, default=96)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--runs", default="f32,bf16,bf16+e4m3/e4m3,bf16+e5m2/e4m3,bf16+e5m2/e5m2")
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--device", default="cpu", help="cpu, or cuda: the same torch operations on the GPU (no library kernel is involved either way)")
    args = ap.parse_args()
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import TrainConfig
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from injected_trainer import InjectedTrainer as CompositeTrainer          # (the renderer is this file's torch model of the arithmetic)
    dev = torch.device(args.device)
    f32r = make_render(Arith("f32"), False)
    data = synthetic.make_dataset(args.det, args.samples, dev, views=synthetic.TRAIN_VIEWS, n_phases=4, F=64,
                                  render=lambda *a: f32r(*a)[0])
    out = {"config": vars(args), "runs": {}}
    for name in args.runs.split(","):
        res = []
        for seed in range(args.seeds):
            ar = Arith(name)
            torch.manual_seed(100 + seed)
            sdef, tdef = synthetic.net_definitions(dev, F=args.filters)
            s, t = CPPN(sdef), Temporal(tdef)
            cfg = TrainConfig(depth_samples_per_ray_coarse=args.samples, img_sample_size=args.rays, static_pos_enc_window_decay_steps=args.steps,
                              temp_pos_enc_window_decay_steps=args.steps, lr_decay_steps=args.steps)
            s, t = s.to(dev), t.to(dev)
            tr = CompositeTrainer(cfg, s, t, data, dev, seed=seed, render=make_render(ar, False), fused_adam=False)
            tr.update_windows(0)
            t0 = time.perf_counter()
            for it in range(args.steps):
                tr.step(it)
            tr._inj_render = f32r            # evaluate every run with the same (f32) renderer
            ev = tr.evaluate(args.steps)
            res.append({"seed": seed, "psnr_mse_db": float(ev["test_psnr_mse"]), "test_psnr_db": float(ev["test_psnr"]), "wall_s": time.perf_counter() - t0})
            print(name, res[-1], file=sys.stderr, flush=True)
        out["runs"][name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
