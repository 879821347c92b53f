#!/usr/bin/env python3
"""Generate golden fixtures by importing the real reference (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (``/root/reference``, pure Python/PyTorch) is imported here, fed seeded or
injected inputs, and its outputs are stored as small ``.npz`` files.  Only data travels:
no reference source, bytecode or text is written.  RNG draws that the reference makes
internally (``torch.rand`` in ``randomize_depth`` / ``sample_pdf``) are reproduced by
seeding the global generator and re-drawing the same sequence, and the draws themselves
are stored next to the outputs so that consumers never need the generator.

Fixture list follows SURVEY.md section 8(c) items (1)-(10).
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

REF = os.environ.get("NERFCA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.modules.setdefault("wandb", types.ModuleType("wandb"))  # data_helpers.py:4 imports it, never used here
sys.path.insert(0, os.path.join(REF, "train"))
sys.path.insert(0, REF)

from model.CPPN import CPPN  # noqa: E402
from model.Temporal import Temporal  # noqa: E402
import model_helpers as MH  # noqa: E402
import proj_helpers as PH  # noqa: E402
import data_helpers as DH  # noqa: E402

torch.set_num_threads(4)
DEV = torch.device("cpu")


def npz(name, **arrs):
    clean = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        clean[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **clean)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  keys={len(clean)}")


def static_def(F=128, early=4, late=0, pos_enc="free_windowed", L=12, start=1, gauss=None, sigma=2):
    return dict(num_early_layers=early, num_late_layers=late, num_filters=F, num_input_channels=3,
                num_output_channels=1, use_bias=True, pos_enc=pos_enc, pos_enc_window_start=start,
                pos_enc_basis=L, fourier_sigma=sigma, fourier_gaussian=gauss, act_func="relu", device=DEV)


def temporal_def(F=128, early=4, late=0, pos_enc="free_windowed", L=12, start=1, T=8, gauss=None, sigma=2):
    d = static_def(F, early, late, pos_enc, L, start, gauss, sigma)
    d.update(num_input_times=1, use_time_latents=True, num_time_dim=T)
    return d


def sd(model, prefix):
    return {f"{prefix}{k}": v.detach().clone() for k, v in model.state_dict().items()}


def grads(model, prefix):
    return {f"{prefix}{k}": p.grad.detach().clone() for k, p in model.named_parameters()}


# ----------------------------------------------------------------------------------- (1)
def gen_posenc():
    torch.manual_seed(101)
    x = (torch.rand(257, 3) * 2 - 1).float()
    out = {"x": x}
    m = CPPN(static_def(pos_enc="none", L=0))
    out["none"] = m.pos_enc(x, 0, "pts")
    m = CPPN(static_def(pos_enc="vanilla"))
    out["plain"] = m.pos_enc(x, 12, "pts")
    m = CPPN(static_def(pos_enc="nerfies_windowed"))
    for a in (0.0, 3.3, 12.0):
        m.windowed_alpha = a
        out[f"nerfies_a{a}"] = m.pos_enc(x, 12, "pts")
        out[f"nerfies_window_a{a}"] = m.windowed_pos_enc(12, "pts")
    m = CPPN(static_def(pos_enc="free_windowed"))
    t = Temporal(temporal_def(pos_enc="free_windowed"))
    for it in (0, 1000, 75000, 150000):
        m.update_freq_mask_alpha(it, 150000)
        t.update_freq_mask_alpha(it, 150000)
        out[f"free_it{it}"] = m.pos_enc(x, 12, "pts")
        out[f"free_temporal_it{it}"] = t.pos_enc(x, 12)
        out[f"free_mask_it{it}"] = m.freq_mask_alpha
    g = torch.Generator().manual_seed(7)
    gauss = torch.randn(3 * 12, generator=g)
    m = CPPN(static_def(pos_enc="fourier", gauss=gauss, sigma=2))
    out["fourier_gauss"] = gauss
    out["fourier_sigma"] = np.array(2)
    out["fourier"] = m.pos_enc(x, 12, "pts")
    # points at the magnitude the real geometry produces, f32 exactness of x*2^k matters there
    x2 = (torch.rand(64, 3) * 2.4 - 1.2).float()
    m = CPPN(static_def(pos_enc="vanilla"))
    out["x_wide"] = x2
    out["plain_wide"] = m.pos_enc(x2, 12, "pts")
    npz("posenc", **out)


# ----------------------------------------------------------------------------------- (2)
def gen_mlps():
    out = {}
    torch.manual_seed(202)
    x = (torch.rand(96, 3) * 2 - 1).float()
    ts = torch.randint(0, 10, (96,)).int()
    gout = torch.randn(96, 1)
    out["x"], out["ts"], out["gout"] = x, ts, gout
    cases = []
    for F in (32, 64, 128):
        for early in (0, 4):
            for late in (0, 2):
                cases.append((F, early, late))
    for (F, early, late) in cases:
        tag = f"F{F}_e{early}_l{late}"
        torch.manual_seed(1000 + F + early * 7 + late)
        m = CPPN(static_def(F=F, early=early, late=late))
        m.update_freq_mask_alpha(60000, 150000)
        y = m(x)
        (y * gout).sum().backward()
        out[f"s_{tag}_y"] = y
        out[f"s_{tag}_mask"] = m.freq_mask_alpha
        out.update(sd(m, f"s_{tag}_p_"))
        out.update(grads(m, f"s_{tag}_g_"))
        if late == 0:
            t = Temporal(temporal_def(F=F, early=early, late=0))
            t.update_freq_mask_alpha(60000, 150000)
            y = t.forward_composite(x, ts)
            (y * gout).sum().backward()
            out[f"d_{tag}_y"] = y
            out.update(sd(t, f"d_{tag}_p_"))
            out.update(grads(t, f"d_{tag}_g_"))
    # other encodings through the whole net (F=64)
    for enc in ("none", "vanilla", "nerfies_windowed", "fourier"):
        torch.manual_seed(77)
        g = torch.Generator().manual_seed(9)
        gauss = torch.randn(3 * 6, generator=g)
        L = 0 if enc == "none" else 6
        m = CPPN(static_def(F=64, early=2, pos_enc=enc, L=L, gauss=gauss, sigma=3))
        t = Temporal(temporal_def(F=64, early=2, pos_enc=enc, L=L, T=4, gauss=gauss, sigma=3))
        if enc == "nerfies_windowed":
            m.update_windowed_alpha(30000, 100000)
            t.update_windowed_alpha(30000, 100000)
            out["enc_nerfies_alpha"] = np.array(m.windowed_alpha)
        ys = m(x)
        yd = t.forward_composite(x, ts)
        ((ys + yd) * gout).sum().backward()
        out[f"enc_{enc}_ys"], out[f"enc_{enc}_yd"] = ys, yd
        out.update(sd(m, f"enc_{enc}_sp_"))
        out.update(sd(t, f"enc_{enc}_dp_"))
        out.update(grads(m, f"enc_{enc}_sg_"))
        out.update(grads(t, f"enc_{enc}_dg_"))
        if enc == "fourier":
            out["enc_fourier_gauss"] = gauss
    npz("mlps", **out)


# ----------------------------------------------------------------------------------- (3)
def gen_depth():
    z = DH.create_depth_values(3.4259, 5.5741, 192, DEV)
    torch.manual_seed(303)
    zj = MH.randomize_depth(z, DEV)
    torch.manual_seed(303)
    t_rand = torch.rand(z.shape)
    npz("depth", z=z, t_rand=t_rand, z_jit=zj, near=np.array(3.4259), far=np.array(5.5741))


# --------------------------------------------------------------------------- shared geometry
GEO16 = dict(DSD=25.0, DSO=4.5, nDetector=[16, 16], dDetector=[2.0 / 16, 2.0 / 16], offDetector=[0.0, 0.0])
VIEWS = [[-30, 30], [-30, -30], [60, -30], [60, 30]]
TEST_VIEW = [-5, 40]


def sample_rays(R, dtype, seed):
    """R rays drawn from the 4 training views of a 16x16 detector + phases."""
    rng = np.random.default_rng(seed)
    tabs = [np.stack(PH.get_ray_values_tigre(th, ph, 0, GEO16, DEV), 0) for th, ph in VIEWS]  # [2,W,H,3]
    rays = np.stack(tabs, 0).transpose(0, 2, 3, 1, 4).reshape(-1, 2, 3)
    ids = rng.integers(0, rays.shape[0], R)
    o = torch.from_numpy(rays[ids, 0].astype(dtype))
    d = torch.from_numpy(rays[ids, 1].astype(dtype))
    ph = torch.from_numpy(rng.integers(0, 10, R)).long()
    return o, d, ph


def build_models(seed, F=128, Ff=None):
    torch.manual_seed(seed)
    s = CPPN(static_def(F=F))
    t = Temporal(temporal_def(F=F))
    for m in (s, t):
        m.update_freq_mask_alpha(75000, 150000)
    sf = tf = None
    if Ff:
        sf = CPPN(static_def(F=Ff))
        tf = Temporal(temporal_def(F=Ff))
        for m in (sf, tf):
            m.update_freq_mask_alpha(75000, 150000)
    return s, t, sf, tf


# ----------------------------------------------------------------------------------- (4)
def gen_predict_iter():
    out = {}
    for (R, S) in ((8, 16), (64, 192)):
        for dt, dtn in ((np.float64, "f64"), (np.float32, "f32")):
            for nf in (0, 32):
                tag = f"R{R}_S{S}_{dtn}_fine{nf}"
                s, t, sf, tf = build_models(4000 + R + nf, F=128 if S == 192 else 64, Ff=32 if nf else None)
                o, d, ph = sample_rays(R, dt, 40 + R)
                z = DH.create_depth_values(3.4259, 5.5741, S, DEV)
                I0 = torch.full((R,), float(np.log(8.670397)))
                phs = ph[:, None].repeat(1, S)
                seed = 5000 + R + S + nf
                torch.manual_seed(seed)
                res = MH.obtain_train_predictions_iter(s, t, sf, tf, o, d, phs, I0, z, "softplus", 32768, nf, DEV)
                torch.manual_seed(seed)
                t_rand = torch.rand(z.shape)
                out[f"{tag}_t_rand"] = t_rand
                if nf:
                    out[f"{tag}_u"] = torch.rand([R, nf])
                out[f"{tag}_o"], out[f"{tag}_d"], out[f"{tag}_ph"], out[f"{tag}_z"], out[f"{tag}_I0"] = o, d, ph, z, I0
                names = ["pix_c", "sig_s_c", "sig_d_c", "dists_c", "pix_f", "sig_s_f", "sig_d_f", "dists_f"]
                for n, v in zip(names, res):
                    if v is not None:
                        out[f"{tag}_{n}"] = v
                out[f"{tag}_mask"] = s.freq_mask_alpha
                out.update(sd(s, f"{tag}_sp_"))
                out.update(sd(t, f"{tag}_dp_"))
                if nf:
                    out.update(sd(sf, f"{tag}_sfp_"))
                    out.update(sd(tf, f"{tag}_dfp_"))
    # static-only prediction (run_nerf.py path): R=16, S=64
    torch.manual_seed(4100)
    s = CPPN(static_def(F=128))
    s.update_freq_mask_alpha(40000, 80000)
    o, d, _ = sample_rays(16, np.float64, 41)
    z = DH.create_depth_values(3.4259, 5.5741, 64, DEV)
    I0 = torch.full((16,), float(np.log(8.670397)))
    torch.manual_seed(4101)
    pix, sig, dists = MH.obtain_train_predictions_static(s, o, d, I0, z, "softplus", 32768, DEV)
    torch.manual_seed(4101)
    out["static_t_rand"] = torch.rand(z.shape)
    out["static_o"], out["static_d"], out["static_z"], out["static_I0"] = o, d, z, I0
    out["static_pix"], out["static_sig"], out["static_dists"], out["static_mask"] = pix, sig, dists, s.freq_mask_alpha
    out.update(sd(s, "static_sp_"))
    npz("predict_iter", **out)


# ----------------------------------------------------------------------------------- (5)
def gen_render():
    torch.manual_seed(505)
    R, S = 12, 24
    raw_s, raw_d = torch.randn(R, S, 1) * 3, torch.randn(R, S, 1) * 3
    raw_s[0, 0, 0], raw_d[0, 1, 0] = 25.0, -30.0  # softplus threshold branch / tiny values
    I0 = torch.full((R,), 2.15991)
    z = MH.randomize_depth(DH.create_depth_values(3.4, 5.6, S, DEV), DEV)
    out = dict(raw_s=raw_s, raw_d=raw_d, I0=I0, z=z)
    for dt, dtn in ((torch.float64, "f64"), (torch.float32, "f32")):
        dirs = torch.randn(R, 3).to(dt)
        for act in ("softplus", "clamp", "Softplus"):
            p, a, b, dd = MH.render_volume_density_composite(raw_s, raw_d, I0, dirs, z, act)
            out[f"comp_{dtn}_{act}_pix"], out[f"comp_{dtn}_{act}_sig_s"], out[f"comp_{dtn}_{act}_sig_d"], out[f"comp_{dtn}_{act}_dists"] = p, a, b, dd
            p, a, dd = MH.render_volume_density(raw_s, I0, dirs, z, act)
            out[f"single_{dtn}_{act}_pix"], out[f"single_{dtn}_{act}_sig"], out[f"single_{dtn}_{act}_dists"] = p, a, dd
    npz("render", **out)


# ----------------------------------------------------------------------------------- (6)
def loss_args():
    return Namespace(favor_s_opt=None, skewness_val=1.0, entro_mask_thre=1e-4, entro_use_weighting=True,
                     entro_weighted_thresh=0.03, occl_reg_perc=0.2)


def gen_losses():
    out = {}
    torch.manual_seed(606)
    R, S = 20, 48
    for dtn, dt in (("f64", torch.float64), ("f32", torch.float32)):
        sig_s = (torch.rand(R, S) * 2e-2).requires_grad_(True)
        sig_d = (torch.rand(R, S) * 2e-2).requires_grad_(True)
        with torch.no_grad():
            sig_d[3] *= 1e-6  # a ray below the entropy mask threshold
            sig_s[5] *= 1e-6
        z = MH.randomize_depth(DH.create_depth_values(3.4, 5.6, S, DEV), DEV)
        dists = torch.cat((z[1:] - z[:-1], torch.tensor([1e-10], dtype=dt)))
        wpix = (1 + torch.rand(R)).to(dt)
        wpix[3] = 1.0
        res = MH.compute_losses(sig_s, sig_d, dists, wpix, loss_args())
        names = ["blendw", "sig_s_max", "sig_d_max", "favor", "s_ent", "s_sum", "d_ent", "d_sum", "occl", "l1", "l2"]
        for n, v in zip(names, res):
            out[f"{dtn}_{n}"] = v
        # gradient of a fixed mixture of the differentiable terms wrt sigma
        mix = 0.7 * res[3] + 1.3 * res[4] + 0.9 * res[6] + 0.5 * res[8] + 0.25 * res[9] + 2.0 * res[10]
        mix.backward()
        out[f"{dtn}_sig_s"], out[f"{dtn}_sig_d"], out[f"{dtn}_dists"], out[f"{dtn}_wpix"] = sig_s, sig_d, dists, wpix
        out[f"{dtn}_g_sig_s"], out[f"{dtn}_g_sig_d"] = sig_s.grad, sig_d.grad
        pred = torch.randn(R).to(dt)
        gt = torch.randn(R).to(dt)
        out[f"{dtn}_mse_pred"], out[f"{dtn}_mse_gt"] = pred, gt
        out[f"{dtn}_mse"] = MH.weighted_MSELoss()(pred, gt, wpix)
        out[f"{dtn}_occl_back"] = MH.compute_occl_loss(sig_d, dists, 0.2, use_back=True)
    npz("losses", **out)


# ----------------------------------------------------------------------------------- (7)
def gen_full_step():
    out = {}
    R, S = 64, 48
    s, t, _, _ = build_models(7000, F=64)
    out.update(sd(s, "init_sp_"))
    out.update(sd(t, "init_dp_"))
    plist = list(t.parameters()) + list(s.parameters())
    opt = torch.optim.Adam([{"params": plist, "lr": 1e-3}], lr=1e-3)
    sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1, end_factor=0.01, total_iters=150000)
    o, d, ph = sample_rays(R, np.float64, 70)
    rng = np.random.default_rng(71)
    gt = torch.from_numpy(rng.uniform(0.5, 2.0, R))
    wpix = torch.from_numpy(1 + rng.uniform(0, 1, R))
    z = DH.create_depth_values(3.4259, 5.5741, S, DEV)
    I0 = torch.full((R,), float(np.log(8.670397)))
    phs = ph[:, None].repeat(1, S)
    out["o"], out["d"], out["ph"], out["gt"], out["wpix"], out["z"], out["I0"] = o, d, ph, gt, wpix, z, I0
    args = loss_args()
    sch = dict(favor_s_weight_start=1e-12, favor_s_weight_end=1e-10, favor_s_weight_delay_steps=40000,
               dynamic_entro_weight_start=1e-10, dynamic_entro_weight_end=1e-8, occl_weight_start=1e-8,
               occl_weight_end=1e-4, l1_weight_start=1e-8, l1_weight_end=1e-15, hyperparam_decay_steps=100000)
    base_iter = 50000  # beyond the favor/occl delay so every term is live
    for k in range(3):
        n_iter = base_iter + k
        s.update_freq_mask_alpha(n_iter, 150000)
        t.update_freq_mask_alpha(n_iter, 150000)
        fw = MH.linear_param_decay(n_iter, sch["favor_s_weight_start"], sch["favor_s_weight_end"], sch["hyperparam_decay_steps"], delay_steps=sch["favor_s_weight_delay_steps"])
        ew = MH.linear_param_decay(n_iter, sch["dynamic_entro_weight_start"], sch["dynamic_entro_weight_end"], sch["hyperparam_decay_steps"])
        ow = MH.linear_param_decay(n_iter, sch["occl_weight_start"], sch["occl_weight_end"], sch["hyperparam_decay_steps"], delay_steps=sch["favor_s_weight_delay_steps"])
        lw = MH.linear_param_decay(n_iter, sch["l1_weight_start"], sch["l1_weight_end"], sch["hyperparam_decay_steps"])
        seed = 7100 + k
        torch.manual_seed(seed)
        res = MH.obtain_train_predictions_iter(s, t, None, None, o, d, phs, I0, z, "softplus", 32768, 0, DEV)
        torch.manual_seed(seed)
        out[f"step{k}_t_rand"] = torch.rand(z.shape)
        pix, sig_s, sig_d, dists = res[:4]
        pixel = MH.weighted_MSELoss()(pix, gt, wpix).mean()
        L = MH.compute_losses(sig_s, sig_d, dists, wpix, args)
        loss = pixel + fw * L[3] + ew * L[6] + ow * L[8] + lw * L[10] + lw * L[9]
        opt.zero_grad()
        loss.backward()
        out[f"step{k}_loss"], out[f"step{k}_pixel"], out[f"step{k}_pix"] = loss, pixel, pix
        out[f"step{k}_weights"] = np.array([fw, ew, ow, lw])
        if k == 0:
            out.update(grads(s, "step0_sg_"))
            out.update(grads(t, "step0_dg_"))
        opt.step()
        sched.step()
    out.update(sd(s, "final_sp_"))
    out.update(sd(t, "final_dp_"))
    out["base_iter"] = np.array(base_iter)
    npz("full_step", **out)



# ----------------------------------------------------------------------------------- (7-fine)
def gen_full_step_fine():
    """Three iterations of run_composite.py's loop body WITH the hierarchical pass (depth_samples_per_ray_fine > 0, :283-308):
    four nets in one optimiser (:192, :207), fine pixel loss with unit weights, fine regularisers with the pixel weights
    (:294-301).  The reference does not detach the sampled depths, so the coarse nets' gradients contain the through-depth term."""
    out = {}
    R, S, NF = 32, 40, 16
    s, t, sf, tf = build_models(7300, F=64, Ff=32)
    out.update(sd(s, "init_sp_")); out.update(sd(t, "init_dp_")); out.update(sd(sf, "init_sfp_")); out.update(sd(tf, "init_dfp_"))
    plist = list(t.parameters()) + list(s.parameters()) + list(tf.parameters()) + list(sf.parameters())
    opt = torch.optim.Adam([{"params": plist, "lr": 1e-3}], lr=1e-3)
    sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1, end_factor=0.01, total_iters=150000)
    o, d, ph = sample_rays(R, np.float64, 73)
    rng = np.random.default_rng(74)
    gt = torch.from_numpy(rng.uniform(0.5, 2.0, R))
    wpix = torch.from_numpy(1 + rng.uniform(0, 1, R))
    ones = torch.ones_like(wpix)
    z = DH.create_depth_values(3.4259, 5.5741, S, DEV)
    I0 = torch.full((R,), float(np.log(8.670397)))
    phs = ph[:, None].repeat(1, S)
    out["o"], out["d"], out["ph"], out["gt"], out["wpix"], out["z"], out["I0"] = o, d, ph, gt, wpix, z, I0
    args = loss_args()
    base_iter = 50000
    for k in range(3):
        n_iter = base_iter + k
        for m in (s, t, sf, tf):
            m.update_freq_mask_alpha(n_iter, 150000)
        fw = MH.linear_param_decay(n_iter, 1e-12, 1e-10, 100000, delay_steps=40000)
        ew = MH.linear_param_decay(n_iter, 1e-10, 1e-8, 100000)
        ow = MH.linear_param_decay(n_iter, 1e-8, 1e-4, 100000, delay_steps=40000)
        lw = MH.linear_param_decay(n_iter, 1e-8, 1e-15, 100000)
        seed = 7400 + k
        torch.manual_seed(seed)
        res = MH.obtain_train_predictions_iter(s, t, sf, tf, o, d, phs, I0, z, "softplus", 32768, NF, DEV)
        torch.manual_seed(seed)
        out[f"step{k}_t_rand"] = torch.rand(z.shape)
        out[f"step{k}_u"] = torch.rand([R, NF])
        pix, sig_s, sig_d, dists, pix_f, sig_sf, sig_df, dists_f = res
        pixel = MH.weighted_MSELoss()(pix, gt, wpix).mean()
        L = MH.compute_losses(sig_s, sig_d, dists, wpix, args)
        loss = pixel + fw * L[3] + ew * L[6] + ow * L[8] + lw * L[10] + lw * L[9]
        pixel_f = MH.weighted_MSELoss()(pix_f, gt, ones).mean()
        Lf = MH.compute_losses(sig_sf, sig_df, dists_f, wpix, args)
        loss = loss + pixel_f + fw * Lf[3] + ew * Lf[6] + ow * Lf[8] + lw * Lf[10] + lw * Lf[9]
        opt.zero_grad()
        loss.backward()
        out[f"step{k}_loss"], out[f"step{k}_pixel"], out[f"step{k}_pixel_f"] = loss, pixel, pixel_f
        out[f"step{k}_pix"], out[f"step{k}_pix_f"] = pix, pix_f
        if k == 0:
            out.update(grads(s, "step0_sg_")); out.update(grads(t, "step0_dg_"))
            out.update(grads(sf, "step0_sfg_")); out.update(grads(tf, "step0_dfg_"))
        opt.step()
        sched.step()
    out.update(sd(s, "final_sp_")); out.update(sd(t, "final_dp_")); out.update(sd(sf, "final_sfp_")); out.update(sd(tf, "final_dfp_"))
    out["base_iter"] = np.array(base_iter)
    out["n_fine"] = np.array(NF)
    npz("full_step_fine", **out)


# ----------------------------------------------------------------------------------- (7b)
def gen_static_step():
    """Three iterations of the body of train/run_nerf.py:186-231 (BASELINE configs[0]: static CPPN, 64 samples per
    ray): obtain_train_predictions_static, weighted MSE + occl_weight_start * occlusion, Adam + LinearLR."""
    out = {}
    R, S = 96, 64
    torch.manual_seed(8000)
    s = CPPN(static_def(F=64))
    out.update(sd(s, "init_sp_"))
    plist = [p for _, p in s.named_parameters()]
    opt = torch.optim.Adam([{"params": plist, "lr": 1e-3}], lr=1e-3)
    sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1, end_factor=0.01, total_iters=150000)
    o, d, _ = sample_rays(R, np.float64, 80)
    rng = np.random.default_rng(81)
    gt = torch.from_numpy(rng.uniform(0.5, 2.0, R))
    wpix = torch.from_numpy(1 + rng.uniform(0, 1, R))
    z = DH.create_depth_values(3.4259, 5.5741, S, DEV)
    I0 = torch.full((R,), float(np.log(8.670397)))
    out["o"], out["d"], out["gt"], out["wpix"], out["z"], out["I0"] = o, d, gt, wpix, z, I0
    occl_w, perc = 1e-2, 0.2          # larger than composite.txt's 1e-8 so the term is visible in the gradients
    out["occl_weight_start"], out["occl_reg_perc"] = np.array(occl_w), np.array(perc)
    base_iter = 60000
    for k in range(3):
        n_iter = base_iter + k
        s.update_freq_mask_alpha(n_iter, 150000)
        seed = 8100 + k
        torch.manual_seed(seed)
        pix, sig, dists = MH.obtain_train_predictions_static(s, o, d, I0, z, "softplus", 32768, DEV)
        torch.manual_seed(seed)
        out[f"step{k}_t_rand"] = torch.rand(z.shape)
        pixel = MH.weighted_MSELoss()(pix, gt, wpix).mean()
        occl = torch.sum(MH.compute_occl_loss(sig, dists, perc))
        loss = pixel + occl_w * occl
        opt.zero_grad()
        loss.backward()
        out[f"step{k}_loss"], out[f"step{k}_pixel"], out[f"step{k}_occl"], out[f"step{k}_pix"] = loss, pixel, occl, pix
        if k == 0:
            out.update(grads(s, "step0_sg_"))
            out["step0_sigma"] = sig
        opt.step()
        sched.step()
    out.update(sd(s, "final_sp_"))
    out["base_iter"] = np.array(base_iter)
    npz("static_step", **out)


# ----------------------------------------------------------------------------------- (8)
def gen_geometry():
    out = {}
    for N in (16, 200):
        geo = dict(DSD=25.0, DSO=4.5, nDetector=[N, N], dDetector=[2.0 / N, 2.0 / N], offDetector=[0.0, 0.0])
        for i, (th, ph) in enumerate(VIEWS + [TEST_VIEW]):
            ro, rd = PH.get_ray_values_tigre(th, ph, 0, geo, DEV)
            if N == 16:
                out[f"n16_v{i}_o"], out[f"n16_v{i}_d"] = ro, rd
            else:  # keep the big one small: corners + a strided sample
                out[f"n200_v{i}_o0"] = ro[0, 0]
                out[f"n200_v{i}_d_sub"] = rd[::25, ::25]
        out[f"n{N}_pose_v0"] = PH.source_matrix_tigre(np.array([0, 0, -4.5]), VIEWS[0][0], VIEWS[0][1])
    # non-square detector + offsets exercise the w/h bookkeeping
    geo = dict(DSD=25.0, DSO=4.5, nDetector=[6, 4], dDetector=[0.3, 0.5], offDetector=[0.05, -0.1])
    ro, rd = PH.get_ray_values_tigre(20.0, -10.0, 0, geo, DEV)
    out["rect_o"], out["rect_d"] = ro, rd
    # ray table on a synthetic 2-image set, written to a temp dir because the loader reads .npy files
    # (the reference's loader only works for square detectors: its image `.T` swaps W and H)
    import tempfile
    W, H = 5, 5
    geo = dict(DSD=25.0, DSO=4.5, nDetector=[W, H], dDetector=[0.3, 0.5], offDetector=[0.05, -0.1])
    rng = np.random.default_rng(8)
    with tempfile.TemporaryDirectory() as td:
        frames = []
        imgs, vars_ = [], []
        for k, (th, ph) in enumerate([(20.0, -10.0), (-35.0, 15.0)]):
            img = rng.uniform(0, 1, W * H)
            img[0], img[1] = 0.0, 1.0
            var = 1 + rng.uniform(0, 1, W * H)
            fp, wp = os.path.join(td, f"i{k}.npy"), os.path.join(td, f"v{k}.npy")
            np.save(fp, img)
            np.save(wp, var)
            frames.append(dict(theta=th, phi=ph, larm=0, file_path=fp, weighted_file_path=wp, img_min_max=[0.2, 1.7], heart_phase=3 + 4 * k))
            imgs.append(img)
            vars_.append(var)
        rays_train, phases_train = DH.prepare_data_for_loader_tigre(frames, geo, W, H, 8, 0.5, DEV)
    out["table_rays"], out["table_phases"] = rays_train, phases_train
    out["table_imgs"], out["table_vars"] = np.stack(imgs), np.stack(vars_)
    out["table_views"] = np.array([[20.0, -10.0], [-35.0, 15.0]])
    out["table_phase_in"] = np.array([3, 7])
    npz("geometry", **out)


# ----------------------------------------------------------------------------------- (9)
def gen_schedules():
    out = {}
    m = CPPN(static_def())
    its = [0, 1, 999, 12500, 75000, 137499, 137500, 149999, 150000, 200000]
    masks, alphas = [], []
    for it in its:
        m.update_freq_mask_alpha(it, 150000)
        masks.append(m.freq_mask_alpha.numpy().copy())
        alphas.append(float(m.windowed_alpha))
    out["free_its"], out["free_masks"], out["free_alphas"] = np.array(its), np.stack(masks), np.array(alphas)
    m0 = CPPN(static_def(start=0, L=10))
    masks0 = []
    for it in its:
        m0.update_freq_mask_alpha(it, 80000)
        masks0.append(m0.freq_mask_alpha.numpy().copy())
    out["free_masks_start0_L10_max80000"] = np.stack(masks0)
    iters = [0, 10, 39999, 40000, 40001, 90000, 140000, 200000]
    out["decay_iters"] = np.array(iters)
    out["decay_favor"] = np.array([MH.linear_param_decay(i, 1e-12, 1e-10, 100000, delay_steps=40000) for i in iters], dtype=np.float64)
    out["decay_l1"] = np.array([MH.linear_param_decay(i, 1e-8, 1e-15, 100000) for i in iters], dtype=np.float64)
    npz("schedules", **out)


# ---------------------------------------------------------------------------------- (10)
def gen_checkpoint_keys():
    import io
    s = CPPN(static_def(late=2))
    s.update_freq_mask_alpha(10, 100)
    t = Temporal(temporal_def())
    t.update_freq_mask_alpha(10, 100)
    out = {"static_late2_keys": np.array(list(s.state_dict().keys())), "temporal_keys": np.array(list(t.state_dict().keys())),
           "static_late2_shapes": np.array([str(tuple(v.shape)) for v in s.state_dict().values()]),
           "temporal_shapes": np.array([str(tuple(v.shape)) for v in t.state_dict().values()])}
    for name, m in (("static", s), ("temporal", t)):
        buf = io.BytesIO()
        m.save(buf, {"note": 1})
        buf.seek(0)
        ck = torch.load(buf, weights_only=False)
        out[f"{name}_save_keys"] = np.array(list(ck.keys()))
        out[f"{name}_save_version"] = np.array(ck["version"])
    npz("checkpoint_keys", **out)


# ----------------------------------------------------------------------------------- (11)
def gen_wide():
    """Nets outside the fused kernels' range (model/CPPN.py:40-65 takes any num_filters / channel counts): more than 128 units per layer,
    other channel counts than 3 -> 1 -- on points (values + every parameter gradient) and through the composite render of a small ray batch."""
    out = {}
    torch.manual_seed(1111)
    x = (torch.rand(200, 3) * 2 - 1).float()          # more than one 128-row tile, not a multiple of it
    ts = torch.randint(0, 10, (200,)).int()
    gout = torch.randn(200, 1)
    out["x"], out["ts"], out["gout"] = x, ts, gout
    for (F, early, late) in ((136, 1, 0), (136, 1, 2), (256, 1, 0)):
        tag = f"F{F}_e{early}_l{late}"
        torch.manual_seed(1100 + F + late)
        m = CPPN(static_def(F=F, early=early, late=late))
        m.update_freq_mask_alpha(60000, 150000)
        y = m(x)
        (y * gout).sum().backward()
        out[f"s_{tag}_y"] = y
        out.update(sd(m, f"s_{tag}_p_"))
        out.update(grads(m, f"s_{tag}_g_"))
    torch.manual_seed(1199)
    t = Temporal(temporal_def(F=136, early=1, late=0))
    t.update_freq_mask_alpha(60000, 150000)
    y = t.forward_composite(x, ts)
    (y * gout).sum().backward()
    out["d_F136_y"] = y
    out.update(sd(t, "d_F136_p_"))
    out.update(grads(t, "d_F136_g_"))
    # other channel counts: 2 -> 3 (plain bands) and 4 -> 2 (fourier)
    g = torch.Generator().manual_seed(19)
    for (cin, cout, enc, L) in ((2, 3, "vanilla", 5), (4, 2, "fourier", 3), (1, 1, "none", 0)):
        tag = f"c{cin}to{cout}"
        torch.manual_seed(1200 + cin)
        gauss = torch.randn(cin * max(L, 1), generator=g)
        d = static_def(F=48, early=2, late=1, pos_enc=enc, L=L, gauss=gauss, sigma=2)
        d.update(num_input_channels=cin, num_output_channels=cout)
        m = CPPN(d)
        xc = (torch.rand(150, cin, generator=g) * 2 - 1).float()
        gc = torch.randn(150, cout, generator=g)
        y = m(xc)
        (y * gc).sum().backward()
        out[f"{tag}_x"], out[f"{tag}_gout"], out[f"{tag}_y"], out[f"{tag}_gauss"] = xc, gc, y, gauss
        out.update(sd(m, f"{tag}_p_"))
        out.update(grads(m, f"{tag}_g_"))
    # composite render: static 136 + dynamic 136, and static 64 (a fused-kernel net) + dynamic 136
    for (Fs, Fd) in ((136, 136), (64, 136)):
        tag = f"rays_s{Fs}_d{Fd}"
        torch.manual_seed(1300 + Fs)
        s = CPPN(static_def(F=Fs, early=1))
        t = Temporal(temporal_def(F=Fd, early=1))
        for m in (s, t):
            m.update_freq_mask_alpha(75000, 150000)
        R, S = 7, 40
        o, dd, ph = sample_rays(R, np.float64, 77)
        z = DH.create_depth_values(3.4259, 5.5741, S, DEV)
        I0 = torch.full((R,), float(np.log(8.670397)))
        torch.manual_seed(1301)
        pix, sig_s, sig_d, dists, *_ = MH.obtain_train_predictions_iter(s, t, None, None, o, dd, ph[:, None].repeat(1, S), I0, z, "softplus", 32768, 0, DEV)
        torch.manual_seed(1301)
        out[f"{tag}_t_rand"] = torch.rand(z.shape)
        (pix.sum() + 30 * sig_s.sum() + 20 * sig_d.sum()).backward()
        out[f"{tag}_o"], out[f"{tag}_d"], out[f"{tag}_ph"], out[f"{tag}_z"], out[f"{tag}_I0"] = o, dd, ph, z, I0
        out[f"{tag}_pix"], out[f"{tag}_sig_s"], out[f"{tag}_sig_d"] = pix, sig_s, sig_d
        out.update(sd(s, f"{tag}_sp_"))
        out.update(sd(t, f"{tag}_dp_"))
        out.update(grads(s, f"{tag}_sg_"))
        out.update(grads(t, f"{tag}_dg_"))
    npz("wide", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["posenc", "mlps", "depth", "predict_iter", "render", "losses", "full_step", "full_step_fine", "static_step", "geometry", "schedules", "checkpoint_keys", "wide"]
    for w in which:
        globals()["gen_" + w]()
