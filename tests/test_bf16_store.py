"""GPU parity of the bf16 mode's BF16 forward store (round 5: NCA_STORE_BF16, NCA_OPT_BF16_STORE; include/nerfca_hip.h) -- what runs
when nothing may be staged in 8 bits (NCA_OPT_STAGE_FP8 = 0: BASELINE configs[1] "bf16" as written).  The storing forward leaves the
layer inputs as bf16 fragments, the ReLU masks of every layer and the raw outputs; the backward from it (mode 5) recomputes nothing,
writes bf16 output gradients, and the weight gradient contracts bf16 x bf16 with f32 accumulation; the output layer's weight
gradient comes from the last layer's sums (nca_layout.hpp).  What the reference's `loss.backward()` yields: train/run_composite.py:306.

The oracle emulates exactly these roundings: NetSpec(emulate_bf16, emulate_fp8_stage = S, emulate_stage_formats = ("bf16", "bf16")) --
the mode-5 arithmetic (_StagedLinear / _StagedTail) with nothing rounded to 8 bits.
"""
import pytest
import torch

from conftest import rel_err
from oracle import nerfca_oracle as O
from test_fp8_stage import _oracle_grads
from test_hip_parity import BF_GRAD, BF_OUT, make_dynamic, make_static
from test_recompute_bf16 import _hip_grads, _inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _models(dev, ps, pd, F, early, it_d):
    from nerfca_amd import set_precision
    s = make_static(ps, dev, F=F, early=early, late=0)
    t = make_dynamic(pd, dev, F=F, early=early, late=0, T=8)
    set_precision("bf16", s, t)
    s.update_freq_mask_alpha(75000, 150000)
    t.update_freq_mask_alpha(it_d, 150000)
    return s, t


@pytest.mark.parametrize("R,S,F,early", [(8, 16, 32, 1), (33, 50, 64, 3), (64, 192, 128, 4), (7, 500, 128, 4), (300, 70, 128, 2)])
@pytest.mark.parametrize("it_d", [75000, 30000])
def test_bf16_store_vs_emulating_oracle(dev, R, S, F, early, it_d):
    """Every parameter gradient of the backward from the bf16 store against the oracle that rounds what the kernels round; ragged
    tiles; one band window for both nets (one shared input block in the store) or one each; resident and streaming kernels bit-identical;
    several ray chunks equal one; everything the forward returns bit-identical to the default (8-bit staged) plan; the plan's record
    says "bf16 store, mode 5, nothing in 8 bits"."""
    from nerfca_amd import _capi, fused
    gen = torch.Generator().manual_seed(7300 + R + S)
    ss = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=0)
    sd = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    win, win_d = O.freq_mask_alpha(12, 75000, 150000, 1)[0], O.freq_mask_alpha(12, it_d, 150000, 1)[0]
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    cp[: R // 4] = 0; cs[: R // 4] = 0; cd[: R // 4] = 0           # tiles whose upstream gradient is all zero
    pix, a, b, dists, go = _oracle_grads(ps, ss, pd, sd, win, win_d, o, d, ph, I0, z, cp, cs, cd, False, fp8=True, formats=("bf16", "bf16"))
    s, t = _models(dev, ps, pd, F, early, it_d)
    saved = fused.BWD_WORKSPACE_BYTES
    res, plans = {}, {}
    try:
        for name, ws, opts in (("default", 6 << 30, {}), ("store_streaming", 6 << 30, {"stage_fp8": 0, "resident_min_tiles": -1}),
                               ("store_resident", 6 << 30, {"stage_fp8": 0, "resident_min_tiles": 0}), ("store_chunks", 24 << 20, {"stage_fp8": 0, "resident_min_tiles": -1})):
            fused.BWD_WORKSPACE_BYTES = ws
            with fused.PlanScope(**opts) as sc:
                res[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
            plans[name] = sc.decided()
    finally:
        fused.BWD_WORKSPACE_BYTES = saved
    for name in ("store_streaming", "store_resident", "store_chunks"):
        pl = plans[name]
        assert pl["fwd_store_format"] & _capi.STORE_KIND_MASK == _capi.STORE_BF16 and pl["bwd_kernel_mode"] == 5 and pl["stage_fp8"] == 0, (name, pl)
    assert plans["store_resident"]["fwd_resident"] == 1 and plans["store_streaming"]["fwd_resident"] == 0
    assert plans["store_chunks"]["chunks"] > 1 or R * S < 4096, plans["store_chunks"]
    p1, a1, b1, g1 = res["store_streaming"]
    assert rel_err(a1.cpu(), a) < BF_OUT and rel_err(b1.cpu(), b) < BF_OUT
    for k in go:
        assert bool(torch.isfinite(g1[k]).all()), k
        assert rel_err(g1[k].cpu(), go[k]) < BF_GRAD, (k, rel_err(g1[k].cpu(), go[k]))
    for name in ("default", "store_resident", "store_chunks"):          # the forward's arithmetic does not depend on what it stores
        for i in range(3):
            assert torch.equal(res[name][i], res["store_streaming"][i]), (name, i)
    for k in g1:
        assert torch.equal(res["store_resident"][3][k], g1[k]), k                      # resident = streaming, bit for bit
        assert rel_err(res["store_chunks"][3][k], g1[k]) < 5e-6, k                     # same products, another (fixed) summation order


@pytest.mark.parametrize("F,R,S", [(128, 9, 130), (32, 5, 33)])
def test_bf16_store_with_depth_gradients(dev, F, R, S):
    """d loss / d depth from the backward out of the bf16 store equals the one of the store-less recompute backward (both read bf16 D_0
    fragments) up to summation order; the parameter gradients agree to the bf16 tolerance."""
    from nerfca_amd import fused
    gen = torch.Generator().manual_seed(7400 + F)
    ss, sd = O.NetSpec(num_filters=F, num_early_layers=2), O.NetSpec(num_filters=F, num_early_layers=2, num_time_dim=8)
    s, t = _models(dev, O.init_params(ss, gen), O.init_params(sd, gen), F, 2, 10000)
    for m in (s, t):
        m.update_freq_mask_alpha(10000, 150000)
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    dists = O.ray_dists(z, torch.float64)
    res = {}
    for name, opts, mode in (("store", {"stage_fp8": 0}, 5), ("no_store", {"stage_fp8": 0, "bf16_store": 0}, 1)):
        with fused.PlanScope(**opts) as sc:
            res[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd, want_depth=True)
        assert sc.decided()["bwd_kernel_mode"] == mode, sc.decided()
    g1, g0 = res["store"][3], res["no_store"][3]
    assert float(g1["depth"].abs().max()) > 0 and rel_err(g1["depth"], g0["depth"]) < 2e-6
    for k in g0:
        if k != "depth":
            assert rel_err(g1[k], g0[k]) < BF_GRAD, k


def test_bf16_store_full_size_step_and_graph(dev):
    """The bench configuration (65 536 rays x 192 samples, default nets) with `stage_fp8 = 0`: the plan takes the bf16 store; loss terms
    bit-identical to the recompute plan's (same forward), the flat gradient within the summation-order / last-layer-identity noise of
    it (both are bf16 x bf16 contractions of the same operands); the graph-replayed step follows the host-launched one."""
    import nerfca_amd
    from nerfca_amd import _capi, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    data = synthetic.make_dataset(256, 192, dev, views=synthetic.TRAIN_VIEWS[:1], n_phases=2)
    res = []

    def trainer(opts):
        torch.manual_seed(1)
        sdef, tdef = synthetic.net_definitions(dev)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        nerfca_amd.set_precision("bf16", s, t)
        return CompositeTrainer(TrainConfig(depth_samples_per_ray_coarse=192, img_sample_size=65536), s, t, data, dev, seed=0, plan_opts=opts)

    for opts, fmt, mode in (({"stage_fp8": 0}, _capi.STORE_BF16, 5), ({"stage_fp8": 0, "bf16_store": 0}, 0, 1)):
        tr = trainer(opts)
        _, _, terms = tr.step_fused(75000)
        pl = tr.plan()
        assert pl["fwd_store_format"] & _capi.STORE_KIND_MASK == fmt and pl["bwd_kernel_mode"] == mode, pl
        res.append((terms.clone(), torch.cat([p.grad.flatten() for p in tr.params]).clone()))
        del tr
        torch.cuda.empty_cache()
    assert torch.equal(res[0][0], res[1][0])
    gs, gr = res[0][1].double(), res[1][1].double()
    cos = float((gs * gr).sum() / (gs.norm() * gr.norm()))
    print(f"flat gradient at 65 536 x 192, bf16 store (mode 5) vs recompute (mode 1): max-norm distance {rel_err(gs, gr):.2e}, cosine {cos:.8f}")
    assert bool(torch.isfinite(gs).all()) and rel_err(gs, gr) < 5e-3 and cos > 0.99999
    eager, graph = trainer({"stage_fp8": 0}), trainer({"stage_fp8": 0})
    for it in range(4):
        le = float(eager.step_fused(75000 + it)[0].detach())
        lg = float(graph.step_graph(75000 + it)[0].detach())
        assert abs(le - lg) <= 1e-3 * abs(le), (it, le, lg)
    assert graph.plan()["fwd_store_format"] & _capi.STORE_KIND_MASK == _capi.STORE_BF16
