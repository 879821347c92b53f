"""Pin the CPU oracle against fixtures captured from the real reference.

Everything here runs on CPU.  Tolerances: the oracle issues the same torch CPU ops as the
reference, so most comparisons are exact or within a few ulp; integer bookkeeping is exact.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import nerfca_oracle as O

torch.set_num_threads(4)


def spec_from(F, early, late, pos_enc="free_windowed", L=12, T=0, start=1, coef=None):
    return O.NetSpec(num_filters=F, num_early_layers=early, num_late_layers=late, pos_enc=pos_enc, pos_enc_basis=L,
                     pos_enc_window_start=start, num_time_dim=T, fourier_coefficients=coef)


# ------------------------------------------------------------------------------- (1) pos-enc
def test_posenc_modes(golden):
    g = golden("posenc")
    x = g["x"]
    assert torch.equal(O.encode(x, spec_from(8, 0, 0, "none", 0), None), g["none"])
    assert torch.equal(O.encode(x, spec_from(8, 0, 0, "vanilla"), None), g["plain"])
    assert torch.equal(O.encode(g["x_wide"], spec_from(8, 0, 0, "vanilla"), None), g["plain_wide"])
    for a in (0.0, 3.3, 12.0):
        w = O.nerfies_window(12, a)
        assert torch.equal(w, g[f"nerfies_window_a{a}"])
        assert torch.equal(O.encode(x, spec_from(8, 0, 0, "nerfies_windowed"), w), g[f"nerfies_a{a}"])
    for it in (0, 1000, 75000, 150000):
        w, _ = O.freq_mask_alpha(12, it, 150000, 1)
        assert torch.equal(w, g[f"free_mask_it{it}"])
        e = O.encode(x, spec_from(8, 0, 0, "free_windowed"), w)
        assert torch.equal(e, g[f"free_it{it}"])
        assert torch.equal(e, g[f"free_temporal_it{it}"])  # CPPN and Temporal encodings are the same function
    coef = g["fourier_gauss"] * float(g["fourier_sigma"])
    assert torch.equal(O.encode(x, spec_from(8, 0, 0, "fourier", coef=coef), None), g["fourier"])


def test_posenc_feature_order(golden):
    """[x, y, z, then per band: sin(x,y,z), cos(x,y,z)] -- SURVEY.md 8(a) a5."""
    g = golden("posenc")
    x, e = g["x"], g["plain"]
    assert e.shape == (257, 75)
    assert torch.equal(e[:, :3], x)
    k = 5
    assert torch.allclose(e[:, 3 + 6 * k: 6 + 6 * k], torch.sin(x * 2.0 ** k), atol=1e-6)
    assert torch.allclose(e[:, 6 + 6 * k: 9 + 6 * k], torch.cos(x * 2.0 ** k), atol=2e-5)


# ------------------------------------------------------------------------------- (9) schedules
def test_schedules(golden):
    g = golden("schedules")
    for it, m, a in zip(g.np("free_its"), g.np("free_masks"), g.np("free_alphas")):
        w, alpha = O.freq_mask_alpha(12, int(it), 150000, 1)
        assert np.array_equal(w.numpy(), m)
        assert float(alpha) == float(a)
    for it, m in zip(g.np("free_its"), g.np("free_masks_start0_L10_max80000")):
        assert np.array_equal(O.freq_mask_alpha(10, int(it), 80000, 0)[0].numpy(), m)
    its = g.np("decay_iters")
    fav = [O.linear_param_decay(int(i), 1e-12, 1e-10, 100000, delay_steps=40000) for i in its]
    l1 = [O.linear_param_decay(int(i), 1e-8, 1e-15, 100000) for i in its]
    assert np.array_equal(np.array(fav, dtype=np.float64), g.np("decay_favor"))
    assert np.array_equal(np.array(l1, dtype=np.float64), g.np("decay_l1"))
    # inactive bands are 1e-8, not zero; the top clip rounds to 1.0f
    w0 = O.freq_mask_alpha(12, 0, 150000, 1)[0]
    assert w0[0] == 1.0 and w0[1] == np.float32(1e-8) and w0[5] == np.float32(1e-8)


# ------------------------------------------------------------------------------- (2) MLPs
CASES = [(F, e, l) for F in (32, 64, 128) for e in (0, 4) for l in (0, 2)]


@pytest.mark.parametrize("F,early,late", CASES)
def test_static_mlp_forward_backward(golden, F, early, late):
    g = golden("mlps")
    tag = f"F{F}_e{early}_l{late}"
    spec = spec_from(F, early, late)
    params = {k: v.clone().requires_grad_(True) for k, v in g.prefixed(f"s_{tag}_p_").items()}
    assert list(params.keys()) == O.param_names(spec)
    assert {k: tuple(v.shape) for k, v in params.items()} == O.param_shapes(spec)
    y = O.static_forward(params, spec, g["x"], g[f"s_{tag}_mask"])
    assert rel_err(y, g[f"s_{tag}_y"]) < 1e-6
    (y * g["gout"]).sum().backward()
    for k, gr in g.prefixed(f"s_{tag}_g_").items():
        assert rel_err(params[k].grad, gr) < 1e-5, k


@pytest.mark.parametrize("F,early", [(F, e) for F in (32, 64, 128) for e in (0, 4)])
def test_dynamic_mlp_forward_backward(golden, F, early):
    g = golden("mlps")
    tag = f"F{F}_e{early}_l0"
    spec = spec_from(F, early, 0, T=8)
    params = {k: v.clone().requires_grad_(True) for k, v in g.prefixed(f"d_{tag}_p_").items()}
    assert list(params.keys()) == O.param_names(spec)
    w = O.freq_mask_alpha(12, 60000, 150000, 1)[0]
    y = O.dynamic_forward(params, spec, g["x"], g["ts"], w)
    assert rel_err(y, g[f"d_{tag}_y"]) < 1e-6
    (y * g["gout"]).sum().backward()
    for k, gr in g.prefixed(f"d_{tag}_g_").items():
        assert rel_err(params[k].grad, gr) < 1e-5, k


def test_dynamic_late_layers_is_unbound():
    """Temporal with num_late_layers > 0 never assigns its output (Temporal.py:128-135)."""
    spec = spec_from(32, 1, 2, T=4)
    p = O.init_params(spec, torch.Generator().manual_seed(0))
    with pytest.raises(UnboundLocalError):
        O.dynamic_forward(p, spec, torch.zeros(4, 3), torch.zeros(4, dtype=torch.int32), O.freq_mask_alpha(12, 0, 10, 1)[0])


@pytest.mark.parametrize("enc", ["none", "vanilla", "nerfies_windowed", "fourier"])
def test_other_encodings_through_net(golden, enc):
    g = golden("mlps")
    L = 0 if enc == "none" else 6
    coef = g["enc_fourier_gauss"] * 3 if enc == "fourier" else None
    win = O.nerfies_window(L, float(g["enc_nerfies_alpha"])) if enc == "nerfies_windowed" else None
    ss = spec_from(64, 2, 0, enc, L, coef=coef)
    sd_ = spec_from(64, 2, 0, enc, L, T=4, coef=coef)
    ps = {k: v.clone().requires_grad_(True) for k, v in g.prefixed(f"enc_{enc}_sp_").items()}
    pd = {k: v.clone().requires_grad_(True) for k, v in g.prefixed(f"enc_{enc}_dp_").items()}
    ys = O.static_forward(ps, ss, g["x"], win)
    yd = O.dynamic_forward(pd, sd_, g["x"], g["ts"], win)
    assert rel_err(ys, g[f"enc_{enc}_ys"]) < 1e-6 and rel_err(yd, g[f"enc_{enc}_yd"]) < 1e-6
    ((ys + yd) * g["gout"]).sum().backward()
    for k, gr in g.prefixed(f"enc_{enc}_sg_").items():
        assert rel_err(ps[k].grad, gr) < 1e-5, k
    for k, gr in g.prefixed(f"enc_{enc}_dg_").items():
        assert rel_err(pd[k].grad, gr) < 1e-5, k


# ------------------------------------------------------------------------------- (3) depth jitter
def test_depth_jitter(golden):
    g = golden("depth")
    z = O.depth_values(float(g["near"]), float(g["far"]), 192)
    assert torch.equal(z, g["z"])
    assert torch.equal(O.stratified_depths(z, g["t_rand"]), g["z_jit"])


# ------------------------------------------------------------------------------- (4) predict_iter
@pytest.mark.parametrize("R,S", [(8, 16), (64, 192)])
@pytest.mark.parametrize("dtn", ["f64", "f32"])
@pytest.mark.parametrize("nf", [0, 32])
def test_predict_iter(golden, R, S, dtn, nf):
    g = golden("predict_iter")
    tag = f"R{R}_S{S}_{dtn}_fine{nf}"
    F = 128 if S == 192 else 64
    ss, sdn = spec_from(F, 4, 0), spec_from(F, 4, 0, T=8)
    win = g[f"{tag}_mask"]
    zj = O.stratified_depths(g[f"{tag}_z"], g[f"{tag}_t_rand"])
    phs = g[f"{tag}_ph"][:, None].repeat(1, S)
    fine = None
    if nf:
        fine = dict(ps=g.prefixed(f"{tag}_sfp_"), spec_s=spec_from(32, 4, 0), win_s=win, pd=g.prefixed(f"{tag}_dfp_"),
                    spec_d=spec_from(32, 4, 0, T=8), win_d=win, n_fine=nf, u=g[f"{tag}_u"])
    res = O.predict_iter(g.prefixed(f"{tag}_sp_"), ss, win, g.prefixed(f"{tag}_dp_"), sdn, win, g[f"{tag}_o"], g[f"{tag}_d"],
                         phs, g[f"{tag}_I0"], zj, "softplus", fine)
    names = ["pix_c", "sig_s_c", "sig_d_c", "dists_c", "pix_f", "sig_s_f", "sig_d_f", "dists_f"]
    for n, v in zip(names, res):
        if v is None:
            assert f"{tag}_{n}" not in g
            continue
        ref = g[f"{tag}_{n}"]
        assert v.dtype == ref.dtype, n  # f64 accident of the real script is reproduced
        assert v.shape == ref.shape, n
        assert rel_err(v, ref) < 2e-6, n
    assert res[0].dtype == (torch.float64 if dtn == "f64" else torch.float32)


def test_predict_static(golden):
    g = golden("predict_iter")
    zj = O.stratified_depths(g["static_z"], g["static_t_rand"])
    pix, sig, dists = O.predict_static(g.prefixed("static_sp_"), spec_from(128, 4, 0), g["static_mask"], g["static_o"],
                                       g["static_d"], g["static_I0"], zj)
    assert rel_err(pix, g["static_pix"]) < 1e-6 and rel_err(sig, g["static_sig"]) < 1e-6
    assert torch.equal(dists, g["static_dists"])


# ------------------------------------------------------------------------------- (5) render
@pytest.mark.parametrize("dtn,dt", [("f64", torch.float64), ("f32", torch.float32)])
@pytest.mark.parametrize("act", ["softplus", "clamp", "Softplus"])
def test_render(golden, dtn, dt, act):
    g = golden("render")
    dirs = torch.zeros(12, 3, dtype=dt)
    p, a, b, dd = O.composite(g["raw_s"], g["raw_d"], g["I0"], dirs, g["z"], act)
    for v, n in ((p, "pix"), (a, "sig_s"), (b, "sig_d"), (dd, "dists")):
        ref = g[f"comp_{dtn}_{act}_{n}"]
        assert v.dtype == ref.dtype and torch.equal(v, ref), n
    p, a, dd = O.composite_single(g["raw_s"], g["I0"], dirs, g["z"], act)
    for v, n in ((p, "pix"), (a, "sig"), (dd, "dists")):
        assert torch.equal(v, g[f"single_{dtn}_{act}_{n}"]), n


def test_activation_default_is_sigmoid():
    """The parser default 'Softplus' (capital S) falls through to Sigmoid (data_helpers.py:58)."""
    x = torch.linspace(-3, 3, 7)
    assert torch.equal(O.activation("Softplus")(x), torch.sigmoid(x))
    assert torch.equal(O.activation("softplus")(x), torch.nn.functional.softplus(x))


# ------------------------------------------------------------------------------- (6) losses
@pytest.mark.parametrize("dtn", ["f64", "f32"])
def test_losses(golden, dtn):
    g = golden("losses")
    sig_s = g[f"{dtn}_sig_s"].clone().requires_grad_(True)
    sig_d = g[f"{dtn}_sig_d"].clone().requires_grad_(True)
    res = O.compute_losses(sig_s, sig_d, g[f"{dtn}_dists"], g[f"{dtn}_wpix"], O.LossArgs())
    names = ["blendw", "sig_s_max", "sig_d_max", "favor", "s_ent", "s_sum", "d_ent", "d_sum", "occl", "l1", "l2"]
    for n, v in zip(names, res):
        assert rel_err(v, g[f"{dtn}_{n}"]) < 1e-6, n
    mix = 0.7 * res[3] + 1.3 * res[4] + 0.9 * res[6] + 0.5 * res[8] + 0.25 * res[9] + 2.0 * res[10]
    mix.backward()
    assert rel_err(sig_s.grad, g[f"{dtn}_g_sig_s"]) < 1e-6
    assert rel_err(sig_d.grad, g[f"{dtn}_g_sig_d"]) < 1e-6
    assert torch.equal(O.weighted_mse(g[f"{dtn}_mse_pred"], g[f"{dtn}_mse_gt"], g[f"{dtn}_wpix"]), g[f"{dtn}_mse"])
    assert rel_err(O.occlusion(sig_d, g[f"{dtn}_dists"], 0.2, use_back=True), g[f"{dtn}_occl_back"]) < 1e-6
    # use_back=False: mask is all ones -> occlusion == mean ray sum
    assert rel_err(res[8], (sig_d * g[f"{dtn}_dists"]).sum(-1).mean()) < 1e-6


# ------------------------------------------------------------------------------- (7) full step
def test_full_training_steps(golden):
    g = golden("full_step")
    ss, sdn = spec_from(64, 4, 0), spec_from(64, 4, 0, T=8)
    tr = O.OracleTrainer(g.prefixed("init_sp_"), ss, g.prefixed("init_dp_"), sdn)
    S = g["z"].shape[0]
    phs = g["ph"][:, None].repeat(1, S)
    base = int(g["base_iter"])
    for k in range(3):
        n_iter = base + k
        assert np.allclose(np.array(O.loss_weights(n_iter, tr.sargs)), g.np(f"step{k}_weights"), rtol=0, atol=0)
        zj = O.stratified_depths(g["z"], g[f"step{k}_t_rand"])
        if k == 0:
            # gradients of step 0 before the optimiser touches anything
            win_s, win_d = tr.windows(n_iter)
            pix, a, b, dd = O.predict_iter(tr.ps, ss, win_s, tr.pd, sdn, win_d, g["o"], g["d"], phs, g["I0"], zj)[:4]
            loss, _, _ = O.composite_total_loss(pix, a, b, dd, g["gt"], g["wpix"], n_iter, tr.largs, tr.sargs)
            loss.backward()
            for name, gr in g.prefixed("step0_sg_").items():
                assert rel_err(tr.ps[name].grad, gr) < 1e-5, name
            for name, gr in g.prefixed("step0_dg_").items():
                assert rel_err(tr.pd[name].grad, gr) < 1e-5, name
        loss, pixel, _ = tr.step(n_iter, g["o"], g["d"], phs, g["I0"], zj, g["gt"], g["wpix"])
        assert rel_err(loss, g[f"step{k}_loss"]) < 1e-6
        assert rel_err(pixel, g[f"step{k}_pixel"]) < 1e-6
    for name, v in g.prefixed("final_sp_").items():
        assert rel_err(tr.ps[name], v) < 1e-5, name
    for name, v in g.prefixed("final_dp_").items():
        assert rel_err(tr.pd[name], v) < 1e-5, name


def test_static_training_steps(golden):
    """BASELINE configs[0]: the static-only loop of train/run_nerf.py on the reference's own trajectory."""
    g = golden("static_step")
    spec = spec_from(64, 4, 0)
    tr = O.OracleStaticTrainer(g.prefixed("init_sp_"), spec, occl_weight_start=float(g["occl_weight_start"]),
                               occl_reg_perc=float(g["occl_reg_perc"]))
    base = int(g["base_iter"])
    for k in range(3):
        n_iter = base + k
        zj = O.stratified_depths(g["z"], g[f"step{k}_t_rand"])
        if k == 0:
            loss, _, _, pix, sig = tr.loss(n_iter, g["o"], g["d"], g["I0"], zj, g["gt"], g["wpix"])
            loss.backward()
            assert rel_err(pix, g["step0_pix"]) < 1e-6 and pix.dtype == g["step0_pix"].dtype
            assert rel_err(sig, g["step0_sigma"]) < 1e-6
            for name, gr in g.prefixed("step0_sg_").items():
                assert rel_err(tr.ps[name].grad, gr) < 1e-5, name
        loss, pixel, occl = tr.step(n_iter, g["o"], g["d"], g["I0"], zj, g["gt"], g["wpix"])
        assert rel_err(loss, g[f"step{k}_loss"]) < 1e-6
        assert rel_err(pixel, g[f"step{k}_pixel"]) < 1e-6
        assert rel_err(occl, g[f"step{k}_occl"]) < 1e-6
    # after Adam: the normalised step m/sqrt(v) turns rounding noise of near-zero gradient entries (which depends on the
    # BLAS thread count of the run that made the fixture) into O(lr * 1e-3) parameter differences
    for name, v in g.prefixed("final_sp_").items():
        assert rel_err(tr.ps[name], v) < 1e-4, name


# ------------------------------------------------------------------------------- (8) geometry
VIEWS = [[-30, 30], [-30, -30], [60, -30], [60, 30], [-5, 40]]


def test_ray_geometry(golden):
    g = golden("geometry")
    for N in (16, 200):
        geo = dict(DSD=25.0, DSO=4.5, nDetector=[N, N], dDetector=[2.0 / N, 2.0 / N], offDetector=[0.0, 0.0])
        assert np.array_equal(O.pose_tigre(VIEWS[0][0], VIEWS[0][1], 4.5), g.np(f"n{N}_pose_v0"))
        for i, (th, ph) in enumerate(VIEWS):
            ro, rd = O.ray_values_tigre(th, ph, geo)
            if N == 16:
                assert np.array_equal(ro, g.np(f"n16_v{i}_o")) and np.array_equal(rd, g.np(f"n16_v{i}_d"))
            else:
                assert np.array_equal(ro[0, 0], g.np(f"n200_v{i}_o0"))
                assert np.array_equal(rd[::25, ::25], g.np(f"n200_v{i}_d_sub"))
    geo = dict(DSD=25.0, DSO=4.5, nDetector=[6, 4], dDetector=[0.3, 0.5], offDetector=[0.05, -0.1])
    ro, rd = O.ray_values_tigre(20.0, -10.0, geo)
    assert ro.shape == (6, 4, 3)
    assert np.array_equal(ro, g.np("rect_o")) and np.array_equal(rd, g.np("rect_d"))
    n = np.linalg.norm(rd, axis=-1)
    assert n.min() >= 1.0 - 1e-6  # directions are not normalised


def test_ray_table_bookkeeping(golden):
    """ray id = img*W*H + w*H + h, rows = (origin, direction, pixel x3, weight x3); exact."""
    g = golden("geometry")
    W = H = 5
    geo = dict(DSD=25.0, DSO=4.5, nDetector=[W, H], dDetector=[0.3, 0.5], offDetector=[0.05, -0.1])
    views, ph_in = g.np("table_views"), g.np("table_phase_in")
    frames = [dict(theta=float(v[0]), phi=float(v[1]), img_min_max=[0.2, 1.7], heart_phase=int(p)) for v, p in zip(views, ph_in)]
    table, phases = O.build_ray_table(frames, list(g.np("table_imgs")), list(g.np("table_vars")), geo, 0.5)
    ref_t, ref_p = g.np("table_rays"), g.np("table_phases")
    assert table.dtype == ref_t.dtype == np.float64 and phases.dtype == ref_p.dtype
    assert np.array_equal(table, ref_t) and np.array_equal(phases, ref_p)
    # explicit id -> (img, w, h) map
    img, w, h = 1, 3, 2
    rid = img * W * H + w * H + h
    ro, rd = O.ray_values_tigre(float(views[img][0]), float(views[img][1]), geo)
    assert np.array_equal(table[rid, 1], rd[w, h].astype(np.float64))
    pix = O.denormalize_image(g.np("table_imgs")[img], W, H, [0.2, 1.7])
    assert table[rid, 2, 0] == pix[w, h] and phases[rid] == ph_in[img]


# ------------------------------------------------------------------------------- (10) checkpoint keys
def test_param_names_match_state_dict(golden):
    g = golden("checkpoint_keys")
    assert O.param_names(spec_from(128, 4, 2)) == list(g.np("static_late2_keys"))
    assert O.param_names(spec_from(128, 4, 0, T=8)) == list(g.np("temporal_keys"))
    shp = O.param_shapes(spec_from(128, 4, 2))
    assert [str(shp[k]) for k in O.param_names(spec_from(128, 4, 2))] == list(g.np("static_late2_shapes"))
    shp = O.param_shapes(spec_from(128, 4, 0, T=8))
    assert [str(shp[k]) for k in O.param_names(spec_from(128, 4, 0, T=8))] == list(g.np("temporal_shapes"))


def test_fine_training_step_loss_and_gradients(golden):
    """Step 0 of the reference's hierarchical loop (tests/golden/full_step_fine.npz: coarse + fine nets, fine pixel loss with
    unit weights, fine regularisers with the pixel weights, run_composite.py:283-306): the oracle's loss and the gradients
    of all four nets -- the coarse nets' contain the term through the sampled depths, which the reference does not detach."""
    g = golden("full_step_fine")
    nf = int(g["n_fine"])
    ss, sd = O.NetSpec(num_filters=64), O.NetSpec(num_filters=64, num_time_dim=8)
    sfs, sfd = O.NetSpec(num_filters=32), O.NetSpec(num_filters=32, num_time_dim=8)
    P = {k: {n: v.clone().requires_grad_(True) for n, v in g.prefixed(f"init_{k}_").items()} for k in ("sp", "dp", "sfp", "dfp")}
    n_iter = int(g["base_iter"])
    win = O.freq_mask_alpha(12, n_iter, 150000, 1)[0]
    S = g["z"].shape[0]
    zj = O.stratified_depths(g["z"], g["step0_t_rand"])
    fine = dict(ps=P["sfp"], spec_s=sfs, win_s=win, pd=P["dfp"], spec_d=sfd, win_d=win, n_fine=nf, u=g["step0_u"])
    pix, a, b, dists, pix_f, af, bf, dists_f = O.predict_iter(P["sp"], ss, win, P["dp"], sd, win, g["o"], g["d"], g["ph"][:, None].repeat(1, S),
                                                              g["I0"], zj, fine=fine)
    la, sa = O.LossArgs(), O.ScheduleArgs()
    loss_c, pixel, _ = O.composite_total_loss(pix, a, b, dists, g["gt"], g["wpix"], n_iter, la, sa)
    pixel_f = O.weighted_mse(pix_f, g["gt"], torch.ones_like(g["wpix"])).mean()
    tf = O.compute_losses(af, bf, dists_f, g["wpix"], la)
    fw, ew, ow, lw = O.loss_weights(n_iter, sa)
    loss = loss_c + pixel_f + fw * tf[3] + ew * tf[6] + ow * tf[8] + lw * tf[10] + lw * tf[9]
    loss.backward()
    assert rel_err(loss, g["step0_loss"]) < 1e-6 and rel_err(pixel, g["step0_pixel"]) < 1e-6 and rel_err(pixel_f, g["step0_pixel_f"]) < 1e-6
    for key, pre in (("sp", "step0_sg_"), ("dp", "step0_dg_"), ("sfp", "step0_sfg_"), ("dfp", "step0_dfg_")):
        for n, ref in g.prefixed(pre).items():
            assert rel_err(P[key][n].grad, ref) < 2e-4, (key, n, rel_err(P[key][n].grad, ref))
