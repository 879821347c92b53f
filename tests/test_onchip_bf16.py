"""GPU parity of the on-chip weight gradient of the bench's backward: the bf16 backward from the forward's store with the last
hidden layer's weight gradient accumulated on chip -- `nca_fused_bf16<F, NCA_KM_BWD_ONCHIP>` ("mode 4": one launch per net,
per-workgroup dW slabs) plus the reduced weight-gradient job set.  The planner selects it from 8 * 8 * CUs wave tiles
(~1.05 M samples); `nca_set_option(NCA_OPT_ONCHIP_MIN_TILES, 0)` forces it at sizes the oracle finishes in seconds, -1
switches it off (mode 3).  What `loss.backward()` yields in the reference: train/run_composite.py:306.

Every test asserts that mode 4 really ran (two dgrad launches per backward, one per net).  The comparisons with the
recompute backward and with mode 3 are exact up to summation order, which holds for bf16 staging (NCA_OPT_STAGE_FP8 = 0);
the default fp8 staging of the same kernels is tested in tests/test_fp8_stage.py; the PSNR gates (tests/test_psnr_gates.py) train
with the library's defaults.
"""
import contextlib
import dataclasses

import pytest
import torch

from conftest import nca_option, rel_err
from oracle import nerfca_oracle as O
from test_hip_parity import BF_GRAD, BF_OUT, make_dynamic, make_static

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@contextlib.contextmanager
def onchip_min_tiles(value):
    """Set the planner's threshold for the duration of a block (None: leave the default)."""
    from nerfca_amd import _capi
    old = _capi.get_option(_capi.OPT_ONCHIP_MIN_TILES)
    if value is not None:
        _capi.set_option(_capi.OPT_ONCHIP_MIN_TILES, value)
    try:
        yield
    finally:
        _capi.set_option(_capi.OPT_ONCHIP_MIN_TILES, old)


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
def test_mode4_forced_vs_emulating_oracle_recompute_and_mode3(dev, R, S, F, early, it_d):
    """Golden-sized batches pushed through mode 4 (threshold forced to 0): every gradient against (i) the bf16-emulating
    oracle, (ii) the recompute backward, (iii) mode 3 (threshold -1), with one ray chunk and with several (small workspace:
    the per-workgroup dW slabs then accumulate over launches).  it_d == 75000: one band window for both nets (the encoded
    input is stored once); 30000: one window per net."""
    from nerfca_amd import fused, set_precision
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
    saved = fused.STORE_FORWARD_LIMIT_BYTES, fused.BWD_WORKSPACE_BYTES
    res, launches = {}, []
    try:
        for name, limit, ws, thr in (("recompute", 0, 6 << 30, None), ("mode3", 96 << 30, 6 << 30, -1), ("mode4", 96 << 30, 6 << 30, 0),
                                     ("mode4_chunks", 96 << 30, 24 << 20, 0)):
            fused.STORE_FORWARD_LIMIT_BYTES, fused.BWD_WORKSPACE_BYTES = limit, ws
            with onchip_min_tiles(thr), nca_option("STAGE_FP8", 0), count_dgrad_launches(launches):
                res[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    finally:
        fused.STORE_FORWARD_LIMIT_BYTES, fused.BWD_WORKSPACE_BYTES = saved
    assert launches[1] >= 1 and launches[2] == 2 * launches[1], launches          # mode 4: one dgrad launch per net and ray chunk
    assert launches[3] % 2 == 0 and launches[3] >= launches[2], launches
    p4, a4, b4, g4 = res["mode4"]
    # (i) the oracle that rounds what the kernel rounds
    assert rel_err(a4.cpu(), a) < BF_OUT and rel_err(b4.cpu(), b) < BF_OUT
    for k, pe in list(("s." + k, v) for k, v in pse.items()) + list(("t." + k, v) for k, v in pde.items()):
        assert rel_err(g4[k].cpu(), pe.grad) < BF_GRAD, k
    # (ii), (iii): same products, other (fixed) summation orders; outputs bit for bit
    for other in ("recompute", "mode3", "mode4_chunks"):
        po, ao, bo, go = res[other]
        assert torch.equal(po, p4) and torch.equal(ao, a4) and torch.equal(bo, b4), other
        for k in g4:
            assert rel_err(go[k], g4[k]) < 2e-6, (other, k)


@pytest.mark.parametrize("F,R,S", [(128, 9, 130), (32, 5, 33)])
def test_mode4_forced_with_depth_gradients(dev, F, R, S):
    """The on-chip path combined with d loss / d depth (per-net dgrad launches, then the depth-gradient kernel reads D_0 of
    both nets from the chunk scratch): parameter and depth gradients equal those of mode 3 and of the recompute backward."""
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
    saved = fused.STORE_FORWARD_LIMIT_BYTES
    res, launches = {}, []
    try:
        for name, limit, thr in (("recompute", 0, None), ("mode3", 96 << 30, -1), ("mode4", 96 << 30, 0)):
            fused.STORE_FORWARD_LIMIT_BYTES = limit
            with onchip_min_tiles(thr), nca_option("STAGE_FP8", 0), count_dgrad_launches(launches):
                res[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd, want_depth=True)
    finally:
        fused.STORE_FORWARD_LIMIT_BYTES = saved
    assert launches == [1, 1, 2], launches
    g4 = res["mode4"][3]
    assert float(g4["depth"].abs().max()) > 0
    for other in ("recompute", "mode3"):
        go = res[other][3]
        for k in g4:
            assert rel_err(go[k], g4[k]) < 2e-6, (other, k)


def test_mode4_natural_threshold_vs_oracle(dev):
    """The planner's own choice at the smallest batch that selects mode 4 on a 256-CU part (5 504 rays x 192 samples = 16 512
    wave tiles >= 8 * 8 * 256), default nets (F=128, 4 hidden layers): every gradient against the bf16-emulating oracle (run
    over ray chunks on the host cores) and against mode 3."""
    from nerfca_amd import _capi, set_precision
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    S = 192
    R = (8 * 8 * cus + 2) // 3 + 40
    assert _capi.get_option(_capi.OPT_ONCHIP_MIN_TILES) == 8 * 8 * cus
    gen = torch.Generator().manual_seed(4242)
    ss, sd = O.NetSpec(num_filters=128), O.NetSpec(num_filters=128, num_time_dim=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    win = O.freq_mask_alpha(12, 75000, 150000, 1)[0]
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    pix, a, b, dists, pse, pde = _oracle_grads_bf16(ps, ss, pd, sd, win, win, o, d, ph, I0, z, cp, cs, cd, ray_chunk=512)
    s = make_static(ps, dev, F=128, early=4, late=0)
    t = make_dynamic(pd, dev, F=128, early=4, late=0, T=8)
    set_precision("bf16", s, t)
    for m in (s, t):
        m.update_freq_mask_alpha(75000, 150000)
    launches = []
    with nca_option("STAGE_FP8", 0), count_dgrad_launches(launches):
        p4, a4, b4, g4 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    with nca_option("STAGE_FP8", 0), onchip_min_tiles(-1), count_dgrad_launches(launches):
        p3, a3, b3, g3 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    assert launches == [2, 1], launches
    assert rel_err(a4.cpu(), a) < BF_OUT and rel_err(b4.cpu(), b) < BF_OUT
    worst = 0.0
    for k, pe in list(("s." + k, v) for k, v in pse.items()) + list(("t." + k, v) for k, v in pde.items()):
        e = rel_err(g4[k].cpu(), pe.grad)
        worst = max(worst, e)
        assert e < BF_GRAD, (k, e)
        assert rel_err(g3[k], g4[k]) < 2e-6, k
    print(f"mode 4 at {R} x {S}: worst gradient distance from the bf16-emulating oracle {worst:.2e}")


def test_mode4_full_size_step_equals_mode3(dev):
    """One `step_fused` at the bench configuration (65 536 rays x 192 samples, default nets, bf16): the loss terms and the flat
    gradient with the on-chip layer (the planner's default there) equal those of mode 3."""
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    data = synthetic.make_dataset(256, 192, dev, views=synthetic.TRAIN_VIEWS[:1], n_phases=2)
    res, launches = [], []
    for thr in (None, -1):
        torch.manual_seed(1)
        sdef, tdef = synthetic.net_definitions(dev)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        nerfca_amd.set_precision("bf16", s, t)
        cfg = TrainConfig(depth_samples_per_ray_coarse=192, img_sample_size=65536)
        tr = CompositeTrainer(cfg, s, t, data, dev, seed=0)
        with onchip_min_tiles(thr), nca_option("STAGE_FP8", 0), count_dgrad_launches(launches):
            _, _, terms = tr.step_fused(75000)
        grads = torch.cat([p.grad.flatten() for p in tr.params]).clone()
        res.append((terms.clone(), grads))
    assert launches == [2, 1], launches
    assert torch.equal(res[0][0], res[1][0])                    # forward and losses do not depend on the backward's mode
    e = rel_err(res[0][1], res[1][1])
    print(f"flat gradient, mode 4 vs mode 3 at 65 536 x 192: {e:.2e}")
    assert e < 5e-6
