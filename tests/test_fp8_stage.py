"""GPU parity of the bf16 mode's 8-bit staged forward store (NCA_OPT_STAGE_FP8, the default whenever a backward follows a forward):
the layer inputs and output gradients that only the weight-gradient kernel reads cross HBM as 8-bit floats -- inputs of
layers 1..NL-2 as e4m3 (x 4), every stored output gradient as e5m2 scaled by a power of two per 64-sample tile of a ray --
while the MLP contractions of the forward and of the dgrad chain stay bf16 with f32 accumulation (nca_layout.hpp).  What the
reference's `loss.backward()` yields: train/run_composite.py:306.

The oracle emulates the same roundings (NetSpec.emulate_fp8_stage / emulate_onchip_last / emulate_stage_formats).  Two
measured distances per gradient: the kernels with fp8 staging from the oracle that stages in fp8, and the kernels WITHOUT a store
(NCA_OPT_STAGE_FP8 = 0: the recompute backward, bf16 operands everywhere -- tests/test_recompute_bf16.py) from the oracle that
does not stage; the first must be within the bound of the other bf16 tests (5e-2 of the max-norm) or
within 1e-2 of the second -- although the staging itself moves a gradient of a few thousand random-signed samples by
5 .. 25 % of its max-norm.  Outputs must be BIT-identical with and without the store (the forward's arithmetic does not change).  The training-quality gate (held-out PSNR within 0.1 dB of f32 at
the bench configuration) is tests/test_psnr_gates.py, which runs the defaults (and both stagings at the reference's default batch).
"""
import dataclasses

import pytest
import torch

from conftest import nca_option, rel_err
from oracle import nerfca_oracle as O
from test_hip_parity import BF_GRAD, BF_OUT, make_dynamic, make_static
from test_recompute_bf16 import _hip_grads, _inputs, count_dgrad_launches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _oracle_grads(ps, ss, pd, sd, win, win_d, o, d, ph, I0, z, cp, cs, cd, onchip, fp8=True, ray_chunk=None, formats=("e5m2", "e4m3")):
    R, S = o.shape[0], z.shape[0]
    kw = dict(emulate_bf16=True, emulate_fp8_stage=S if fp8 else 0, emulate_onchip_last=onchip, emulate_stage_formats=formats)
    sse, sde = dataclasses.replace(ss, **kw), dataclasses.replace(sd, **kw)
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
    g = {"s." + k: v.grad for k, v in pso.items()}
    g.update({"t." + k: v.grad for k, v in pdo.items()})
    return pix, a, b, dists, g


@pytest.mark.parametrize("R,S,F,early", [(8, 16, 32, 1), (33, 50, 64, 3), (64, 192, 128, 4), (7, 500, 128, 4), (300, 70, 128, 2), (40, 130, 64, 0)])
@pytest.mark.parametrize("it_d", [75000, 30000])
def test_fp8_stage_vs_emulating_oracle(dev, R, S, F, early, it_d):
    """Every parameter gradient of the backward from an fp8-staged store (mode 5: nothing recomputed -- raw outputs and the masks
    of all layers come from the store, every output gradient is staged as e5m2, the output layer's weight gradient is a job of
    the weight-gradient kernel) against the oracle that rounds what the kernels round; ragged tiles (S not a multiple of 64),
    nets without a hidden layer to stage (early = 0: no store, the recompute backward), one band window for both nets or one
    each; outputs bit-identical to the store-less plan; several ray chunks equal one."""
    from nerfca_amd import fused, set_precision
    onchip = False
    gen = torch.Generator().manual_seed(300 + R + S)
    ss = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=0)
    sd = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    win, win_d = O.freq_mask_alpha(12, 75000, 150000, 1)[0], O.freq_mask_alpha(12, it_d, 150000, 1)[0]
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    cp[: R // 4] = 0; cs[: R // 4] = 0; cd[: R // 4] = 0           # tiles whose upstream gradient is all zero
    pix, a, b, dists, go = _oracle_grads(ps, ss, pd, sd, win, win_d, o, d, ph, I0, z, cp, cs, cd, onchip, fp8=early > 0)
    go16 = _oracle_grads(ps, ss, pd, sd, win, win_d, o, d, ph, I0, z, cp, cs, cd, onchip, fp8=early > 0, formats=None)[4]
    s = make_static(ps, dev, F=F, early=early, late=0)
    t = make_dynamic(pd, dev, F=F, early=early, late=0, T=8)
    set_precision("bf16", s, t)
    s.update_freq_mask_alpha(75000, 150000)
    t.update_freq_mask_alpha(it_d, 150000)
    saved = fused.BWD_WORKSPACE_BYTES
    launches = []
    try:
        with count_dgrad_launches(launches):
            p8, a8, b8, g8 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
        with nca_option("STAGE_FP8", 0), nca_option("BF16_STORE", 0):          # no store: the recompute backward, nothing in 8 bits
            p16, a16, b16, g16 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
        fused.BWD_WORKSPACE_BYTES = 24 << 20
        pc, ac, bc, gc = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    finally:
        fused.BWD_WORKSPACE_BYTES = saved
    if early > 0:
        from nerfca_amd import _capi
        per_net = 0 <= _capi.get_option(_capi.OPT_RESIDENT_MIN_TILES) <= R * ((S + 63) // 64)     # resident weights: one launch per net
        assert launches == [2 if per_net else 1], launches
    assert torch.equal(p8, p16) and torch.equal(a8, a16) and torch.equal(b8, b16)
    assert torch.equal(pc, p8)
    assert rel_err(a8.cpu(), a) < BF_OUT and rel_err(b8.cpu(), b) < BF_OUT
    worst = base = shift = 0.0
    for k in go:
        assert bool(torch.isfinite(g8[k]).all()), k
        e8, e16 = rel_err(g8[k].cpu(), go[k]), rel_err(g16[k].cpu(), go16[k])
        worst, base, shift = max(worst, e8), max(base, e16), max(shift, rel_err(g8[k], g16[k]))
        # (e16 itself is the business of the bf16 tests; a handful of ReLU mask flips -- pre-activations within rounding of zero,
        # summed in another order -- move a max-norm of ~1e4 random-signed samples by up to ~5e-2 in either staging)
        assert e8 < max(BF_GRAD, e16 + 1e-2), (k, e8, e16)
        assert rel_err(gc[k], g8[k]) < 2e-6, k
    print(f"fp8 staging {R}x{S} F={F} early={early}: worst distance from the oracle that stages in fp8 {worst:.2e} (bf16 staging from its oracle: "
          f"{base:.2e}); the staging itself moves the gradient by {shift:.2e}")


def test_fp8_stage_saturates_instead_of_overflowing(dev):
    """Upstream gradients and weights far outside any sane range: the conversions saturate (MODE.FP16_OVFL), nothing turns
    into inf / NaN on its way through the 8-bit blocks."""
    from nerfca_amd import set_precision
    gen = torch.Generator().manual_seed(7)
    F, early, R, S = 64, 2, 12, 70
    ss, sd = O.NetSpec(num_filters=F, num_early_layers=early), O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    for p in (ps, pd):
        for k in p:
            if k.endswith("weight") and "early_pts_layers" in k and not k.startswith("early_pts_layers.0."):
                p[k] = p[k] * 1000.0           # activations ~1e5 (e4m3 x 4 tops out at 112), deltas amplified ~400x per layer (e5m2 at 57 344)
    s = make_static(ps, dev, F=F, early=early, late=0)
    t = make_dynamic(pd, dev, F=F, early=early, late=0, T=8)
    set_precision("bf16", s, t)
    for m in (s, t):
        m.update_freq_mask_alpha(75000, 150000)
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    dists = O.ray_dists(z, torch.float64)
    _, _, _, g = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp * 1e20, cs * 1e20, cd * 1e20)
    for k, v in g.items():
        assert bool(torch.isfinite(v).all()), k
    assert max(float(v.abs().max()) for v in g.values()) > 0


def test_fp8_stage_with_depth_gradients(dev):
    """d loss / d depth is formed from bf16 D_0 fragments, so a backward that wants it writes its output gradients as bf16
    even though the store's hidden blocks are e4m3: the depth gradient equals the one of the store-less (recompute) backward, the
    parameter gradients agree with the fully fp8-staged ones to the staging noise."""
    from nerfca_amd import set_precision
    gen = torch.Generator().manual_seed(70)
    F, early, R, S = 128, 3, 9, 130
    ss, sd = O.NetSpec(num_filters=F, num_early_layers=early), O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=8)
    s = make_static(O.init_params(ss, gen), dev, F=F, early=early, late=0)
    t = make_dynamic(O.init_params(sd, gen), dev, F=F, early=early, late=0, T=8)
    set_precision("bf16", s, t)
    for m in (s, t):
        m.update_freq_mask_alpha(10000, 150000)
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    dists = O.ray_dists(z, torch.float64)
    _, _, _, gz = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd, want_depth=True)
    with nca_option("STAGE_FP8", 0), nca_option("BF16_STORE", 0):
        _, _, _, gz16 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd, want_depth=True)
    _, _, _, g8 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    assert rel_err(gz["depth"], gz16["depth"]) < 2e-6
    for k in g8:                                            # ~1 200 random-signed samples: the staging noise is 5 .. 25 % of a max-norm
        # (the output layer's weights: 1 170 terms g h that largely cancel, h as e4m3 -- 3 mantissa bits -- instead of bf16)
        tol = 0.5 if "output_linear" in k else 0.3
        assert bool(torch.isfinite(gz[k]).all()) and rel_err(gz[k], gz16[k]) < tol, k           # e4m3 layer inputs vs bf16 ones
        assert rel_err(gz[k], g8[k]) < 0.3, k                                                     # bf16 vs e5m2 output gradients


def test_fp8_stage_full_size_step(dev):
    """One `step_fused` at the bench configuration (65 536 rays x 192 samples, default nets) with the 8-bit staged store against
    the store-less recompute backward: same forward and loss terms bit for bit; the flat gradient moves by the staging noise only
    (the 8-bit rounding errors are zero-mean and average over 12.6 M samples)."""
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    data = synthetic.make_dataset(256, 192, dev, views=synthetic.TRAIN_VIEWS[:1], n_phases=2)
    res = []
    for fp8 in (1, 0):
        torch.manual_seed(1)
        sdef, tdef = synthetic.net_definitions(dev)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        nerfca_amd.set_precision("bf16", s, t)
        tr = CompositeTrainer(TrainConfig(depth_samples_per_ray_coarse=192, img_sample_size=65536), s, t, data, dev, seed=0)
        with nca_option("STAGE_FP8", fp8), nca_option("BF16_STORE", 0):
            _, _, terms = tr.step_fused(75000)
        res.append((terms.clone(), torch.cat([p.grad.flatten() for p in tr.params]).clone()))
    assert torch.equal(res[0][0], res[1][0])
    g8, g16 = res[0][1].double(), res[1][1].double()
    e = rel_err(g8, g16)
    cos = float((g8 * g16).sum() / (g8.norm() * g16.norm()))
    print(f"flat gradient at 65 536 x 192, 8-bit staged store vs recompute in bf16: max-norm distance {e:.2e}, cosine {cos:.7f}")
    assert bool(torch.isfinite(g8).all()) and e < 2e-2 and cos > 0.9999
