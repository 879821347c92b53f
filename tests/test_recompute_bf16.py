"""GPU parity of the bf16 mode WITHOUT a forward store (`NCA_OPT_STAGE_FP8 = 0`: "BASELINE configs[1] as written" -- bf16 operands
everywhere, nothing staged in 8 bits): the forward writes no store and `nca_fused_bf16<F, NCA_KM_BWD>` ("mode 1") recomputes the
layers.  What `loss.backward()` yields in the reference: train/run_composite.py:306.

Until round 3 that option selected a bf16-STAGED store (kernel modes 3 / 4, one layer's weight gradient on chip); it was retired in
round 4 (never better in held-out PSNR than the 8-bit staged store, 1.75 x slower).  These tests also cover the
per-caller planner options that replaced "set a process-wide option, run, set it back": `fused.PlanScope` (NcaRays.plan_opts /
plan_out, ABI 9).
"""
import contextlib
import dataclasses

import pytest
import torch

from conftest import rel_err
from oracle import nerfca_oracle as O
from test_hip_parity import BF_GRAD, BF_OUT, make_dynamic, make_static

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@contextlib.contextmanager
def count_dgrad_launches(out):
    from nerfca_amd import _capi
    _capi.timing_reset()
    _capi.timing_enable(True)
    try:
        yield
    finally:
        out.append(_capi.timing_read("bwd_dgrad")[1])
        _capi.timing_enable(False)
        _capi.timing_reset()


def _inputs(R, S, gen):
    o = (torch.rand(R, 3, generator=gen) * 0.2 + torch.tensor([3.0, -2.0, 2.5])).double()
    d = (torch.rand(R, 3, generator=gen) - 0.5).double()
    d = d / d.norm(dim=-1, keepdim=True) * 1.001
    ph = torch.randint(0, 10, (R,), generator=gen)
    z = O.stratified_depths(O.depth_values(3.4259, 5.5741, S), torch.rand(S, generator=gen))
    I0 = torch.full((R,), 2.15991)
    cp, cs, cd = torch.randn(R, generator=gen).double(), torch.randn(R, S, generator=gen), torch.randn(R, S, generator=gen)
    return o, d, ph, z, I0, cp, cs, cd


def _oracle_grads_bf16(ps, ss, pd, sd, win, win_d, o, d, ph, I0, z, cp, cs, cd, ray_chunk=None):
    """Outputs and parameter gradients of the bf16-emulating oracle; rays are independent, so the backward may run
    over ray chunks (bounded memory) and add up."""
    R, S = o.shape[0], z.shape[0]
    sse, sde = dataclasses.replace(ss, emulate_bf16=True), dataclasses.replace(sd, emulate_bf16=True)
    pso = {k: v.clone().requires_grad_(True) for k, v in ps.items()}
    pdo = {k: v.clone().requires_grad_(True) for k, v in pd.items()}
    outs = []
    step = ray_chunk or R
    for r0 in range(0, R, step):
        sl = slice(r0, min(R, r0 + step))
        n = sl.stop - sl.start
        pts = O.query_points(o[sl], d[sl], z)
        raw_s = O.static_forward(pso, sse, pts, win).reshape(n, S, -1)
        raw_d = O.dynamic_forward(pdo, sde, pts, ph[sl][:, None].repeat(1, S).flatten(), win_d).reshape(n, S, -1)
        pix, a, b, dists = O.composite(raw_s, raw_d, I0[sl], d[sl], z)
        ((pix * cp[sl]).sum() + (a * cs[sl]).sum() * 50 + (b * cd[sl]).sum() * 50).backward()
        outs.append((pix.detach(), a.detach(), b.detach()))
    pix, a, b = (torch.cat([x[i] for x in outs]) for i in range(3))
    return pix, a, b, dists, pso, pdo


def _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd, want_depth=False):
    from nerfca_amd import render_rays
    for m in (s, t):
        m.zero_grad()
    zz = z.to(dev)
    if want_depth:
        zz = zz[None, :].repeat(o.shape[0], 1).clone().requires_grad_(True)
    pix, a, b = render_rays(s, t, o.to(dev), d.to(dev), ph.to(dev), I0.to(dev), zz, dists.to(dev))
    ((pix * cp.to(dev)).sum() + (a * cs.to(dev)).sum() * 50 + (b * cd.to(dev)).sum() * 50).backward()
    g = {"s." + k: p.grad.detach().clone() for k, p in s.named_parameters()}
    g.update({"t." + k: p.grad.detach().clone() for k, p in t.named_parameters()})
    if want_depth:
        g["depth"] = zz.grad.detach().clone()
    return pix.detach(), a.detach(), b.detach(), g


@pytest.mark.parametrize("R,S,F,early", [(8, 16, 32, 1), (33, 50, 64, 3), (64, 192, 128, 4), (7, 500, 128, 4), (300, 70, 128, 2)])
@pytest.mark.parametrize("it_d", [75000, 30000])
def test_no_store_recompute_backward_vs_emulating_oracle(dev, R, S, F, early, it_d):
    """Golden-sized batches with `stage_fp8 = 0` given as a PER-CALL option (PlanScope): the plan says "no store, mode 1", every
    gradient is within the bf16 tolerance of the oracle that rounds what the kernels round, with one ray chunk and with several;
    the forward's outputs are bit-identical to those of the default plan (the 8-bit staged store changes nothing the forward
    returns).  it_d == 75000: one band window for both nets; 30000: one per net."""
    from nerfca_amd import _capi, fused, set_precision
    gen = torch.Generator().manual_seed(900 + R + S)
    ss = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=0)
    sd = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    win, win_d = O.freq_mask_alpha(12, 75000, 150000, 1)[0], O.freq_mask_alpha(12, it_d, 150000, 1)[0]
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    pix, a, b, dists, pse, pde = _oracle_grads_bf16(ps, ss, pd, sd, win, win_d, o, d, ph, I0, z, cp, cs, cd)
    s = make_static(ps, dev, F=F, early=early, late=0)
    t = make_dynamic(pd, dev, F=F, early=early, late=0, T=8)
    set_precision("bf16", s, t)
    s.update_freq_mask_alpha(75000, 150000)
    t.update_freq_mask_alpha(it_d, 150000)
    saved = fused.BWD_WORKSPACE_BYTES
    res, plans = {}, {}
    try:
        for name, ws, opts in (("default", 6 << 30, {}), ("no_store", 6 << 30, {"stage_fp8": 0, "bf16_store": 0}), ("no_store_chunks", 24 << 20, {"stage_fp8": 0, "bf16_store": 0})):
            fused.BWD_WORKSPACE_BYTES = ws
            with fused.PlanScope(**opts) as sc:
                res[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
            plans[name] = sc.decided()
    finally:
        fused.BWD_WORKSPACE_BYTES = saved
    assert _capi.get_option(_capi.OPT_STAGE_FP8) == -1          # the process-wide option was never touched
    assert plans["default"]["fwd_store_format"] & _capi.STORE_KIND_MASK == _capi.STORE_FP8 and plans["default"]["bwd_kernel_mode"] == 5, plans["default"]
    for name in ("no_store", "no_store_chunks"):
        assert plans[name]["fwd_store_format"] == 0 and plans[name]["bwd_kernel_mode"] == 1 and plans[name]["stage_fp8"] == 0, plans[name]
    assert plans["no_store_chunks"]["chunks"] > 1 or R * S < 4096, plans["no_store_chunks"]
    p1, a1, b1, g1 = res["no_store"]
    assert rel_err(a1.cpu(), a) < BF_OUT and rel_err(b1.cpu(), b) < BF_OUT
    for k, pe in list(("s." + k, v) for k, v in pse.items()) + list(("t." + k, v) for k, v in pde.items()):
        assert rel_err(g1[k].cpu(), pe.grad) < BF_GRAD, k
    pc, ac, bc, gc = res["no_store_chunks"]
    assert torch.equal(pc, p1) and torch.equal(ac, a1) and torch.equal(bc, b1)
    for k in g1:
        assert rel_err(gc[k], g1[k]) < 2e-6, k          # same products, another (fixed) summation order
    p0, a0, b0, _ = res["default"]
    assert torch.equal(p0, p1) and torch.equal(a0, a1) and torch.equal(b0, b1)


@pytest.mark.parametrize("F,R,S", [(128, 9, 130), (32, 5, 33)])
def test_no_store_with_depth_gradients(dev, F, R, S):
    """d loss / d depth from the recompute backward (the depth-gradient kernel reads D_0 of both nets from the one scratch) equals
    the one from the default plan (mode 5 with bf16 output-gradient blocks) up to summation order."""
    from nerfca_amd import fused, set_precision
    gen = torch.Generator().manual_seed(77 + F)
    ss, sd = O.NetSpec(num_filters=F, num_early_layers=2), O.NetSpec(num_filters=F, num_early_layers=2, num_time_dim=8)
    s = make_static(O.init_params(ss, gen), dev, F=F, early=2, late=0)
    t = make_dynamic(O.init_params(sd, gen), dev, F=F, early=2, late=0, T=8)
    set_precision("bf16", s, t)
    for m in (s, t):
        m.update_freq_mask_alpha(10000, 150000)
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    dists = O.ray_dists(z, torch.float64)
    res = {}
    for name, opts in (("default", {}), ("no_store", {"stage_fp8": 0, "bf16_store": 0})):
        with fused.PlanScope(**opts) as sc:
            res[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd, want_depth=True)
        assert sc.decided()["bwd_kernel_mode"] == (5 if name == "default" else 1), sc.decided()
    g1, g0 = res["no_store"][3], res["default"][3]
    assert float(g1["depth"].abs().max()) > 0
    assert rel_err(g0["depth"], g1["depth"]) < 2e-6


def test_two_trainers_keep_their_own_planner_options(dev):
    """Per-trainer planner options: two trainers of one process, one with the default plan and one with
    `plan_opts={"stage_fp8": 0, "bf16_store": 0}`, stepped alternately -- each runs ITS plan every time (its own record says so), the process-wide
    option is never written, and each trainer's trajectory is bit-identical to the one it takes when it runs alone."""
    import nerfca_amd
    from nerfca_amd import _capi, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    data = synthetic.make_dataset(32, 64, dev, views=synthetic.TRAIN_VIEWS[:2], n_phases=3, F=32)

    def trainer(opts):
        torch.manual_seed(9)
        sdef, tdef = synthetic.net_definitions(dev)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        nerfca_amd.set_precision("bf16", s, t)
        return CompositeTrainer(TrainConfig(depth_samples_per_ray_coarse=64, img_sample_size=1024), s, t, data, dev, seed=2, plan_opts=opts)

    def run(trs, n=4):
        out = [[] for _ in trs]
        for it in range(n):
            for k, tr in enumerate(trs):
                loss, _, _ = tr.step_fused(75000 + it)
                out[k].append(loss.detach().clone())
                want = (5, 1)[k] if len(trs) == 2 else None
                if want is not None:
                    assert tr.plan()["bwd_kernel_mode"] == want, (k, tr.plan())
        return [torch.stack(o) for o in out], [torch.cat([p.detach().flatten() for p in tr.params]).clone() for tr in trs]

    both_l, both_p = run([trainer(None), trainer({"stage_fp8": 0, "bf16_store": 0})])
    a_l, a_p = run([trainer(None)])
    b_l, b_p = run([trainer({"stage_fp8": 0, "bf16_store": 0})])
    assert _capi.get_option(_capi.OPT_STAGE_FP8) == -1
    assert torch.equal(both_l[0], a_l[0]) and torch.equal(both_p[0], a_p[0])
    assert torch.equal(both_l[1], b_l[0]) and torch.equal(both_p[1], b_p[0])
    assert not torch.equal(a_p[0], b_p[0])          # (they ARE different arithmetics: 8-bit staged vs recomputed bf16 operands of the weight gradient)


def test_plan_options_are_validated_and_the_retired_store_format_is_refused(dev):
    from nerfca_amd import _capi, fused, set_precision
    with pytest.raises(_capi.NcaError):
        _capi.NcaPlanOpts(onchip_min_tiles=0)          # the retired option has no per-call field either
    with pytest.raises(_capi.NcaError):
        _capi.get_option(0)                            # NCA_OPT_RESERVED0
    gen = torch.Generator().manual_seed(3)
    ss, sd = O.NetSpec(num_filters=32, num_early_layers=1), O.NetSpec(num_filters=32, num_early_layers=1, num_time_dim=8)
    s = make_static(O.init_params(ss, gen), dev, F=32, early=1, late=0)
    t = make_dynamic(O.init_params(sd, gen), dev, F=32, early=1, late=0, T=8)
    set_precision("bf16", s, t)
    for m in (s, t):
        m.update_freq_mask_alpha(75000, 150000)
    o, d, ph, z, I0, cp, cs, cd = _inputs(8, 16, gen)
    dists = O.ray_dists(z, torch.float64)
    with fused.PlanScope(stage_fp8=7):          # not 0 / 1 / -1
        with pytest.raises(_capi.NcaError, match="NCA_OPT_STAGE_FP8"):
            _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    # a backward that is told "format 2" (the bf16-staged store of ABI <= 8) is refused
    batch = fused._RayBatch(o.to(dev), d.to(dev), ph.to(dev), I0.to(dev), z.to(dev), dists.to(dev), "softplus", False, 1e-2)
    pix, a, b, keep = fused.render_forward_raw(batch, s._binding, t._binding, for_backward=True)
    assert keep[-1] & _capi.STORE_KIND_MASK == _capi.STORE_FP8
    bad = keep[:-1] + (2,)
    with pytest.raises(_capi.NcaError, match="store_format"):
        fused.render_backward_raw(batch, s._binding, t._binding, bad, torch.ones_like(pix), torch.zeros_like(a), torch.zeros_like(b))
