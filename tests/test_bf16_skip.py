"""GPU parity of the bf16 mode for a CPPN WITH a skip connection (num_late_layers > 0: h = relu(W_skip cat[encoded input, h]), then
num_late_layers - 1 further layers; model/CPPN.py:53-58, 102-106) -- round 6; until then NCA_E_UNSUPPORTED in the bf16 mode.  The skip layer
streams as two LDS stages (encoded part from a re-formed layer-0 operand, hidden part from the previous layer's registers); its weight
gradient is two jobs over the two stored blocks; a last layer that is the skip layer keeps [Wo | bo] behind its second image and its
gradient falls out of sums over K0 + F columns.  Checked in all three backward plans -- from the 8-bit staged store (mode 5), from the
bf16 store, recompute (mode 1) -- against the oracle emulating each plan's roundings, for rays (a static net with a skip layer beside a
plain dynamic net) and for points (CPPN.forward)."""
import dataclasses

import pytest
import torch

from conftest import nca_option, rel_err
from oracle import nerfca_oracle as O
from test_fp8_stage import _oracle_grads
from test_hip_parity import BF_GRAD, BF_OUT, grads_of, make_dynamic, make_static
from test_recompute_bf16 import _hip_grads, _inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("R,S,F,early,late,it", [(120, 50, 64, 2, 1, 150000), (64, 192, 128, 4, 2, 75000), (7, 500, 128, 1, 1, 150000), (40, 70, 32, 0, 3, 75000), (45, 64, 32, 1, 1, 150000),
                                                 (300, 70, 128, 2, 2, 30000)])
def test_bf16_rays_with_a_skip_layer_vs_emulating_oracle(dev, R, S, F, early, late, it):
    """`it` = the iteration of the FreeNeRF window: 150 000 opens every band (all 75 encoded features carry weight: the encoded-part jobs of a narrow
    net must cover all four 32-slot tiles of the input block, also where they rebuild their output-gradient block from mask bits -- late = 1)."""
    from nerfca_amd import fused, set_precision
    gen = torch.Generator().manual_seed(1300 + R + S + late)
    ss = O.NetSpec(num_filters=F, num_early_layers=early, num_late_layers=late, num_time_dim=0)
    sd = O.NetSpec(num_filters=F, num_early_layers=max(early, 1), num_time_dim=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    win = O.freq_mask_alpha(12, it, 150000, 1)[0]
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    cp[: R // 4] = 0; cs[: R // 4] = 0; cd[: R // 4] = 0           # tiles whose upstream gradient is all zero
    pix, a, b, dists, g8o = _oracle_grads(ps, ss, pd, sd, win, win, o, d, ph, I0, z, cp, cs, cd, False)                              # e5m2 / e4m3 staging
    g16so = _oracle_grads(ps, ss, pd, sd, win, win, o, d, ph, I0, z, cp, cs, cd, False, formats=("bf16", "bf16"))[4]                # the bf16 store
    g16o = _oracle_grads(ps, ss, pd, sd, win, win, o, d, ph, I0, z, cp, cs, cd, False, fp8=False, formats=None)[4]                   # no store: recompute
    s = make_static(ps, dev, F=F, early=early, late=late)
    t = make_dynamic(pd, dev, F=F, early=max(early, 1), late=0, T=8)
    set_precision("bf16", s, t)
    for m in (s, t):
        m.update_freq_mask_alpha(it, 150000)
    saved = fused.BWD_WORKSPACE_BYTES
    try:
        p8, a8, b8, g8 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
        with nca_option("STAGE_FP8", 0):
            p16s, a16s, b16s, g16s = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
        with nca_option("STAGE_FP8", 0), nca_option("BF16_STORE", 0):
            p16, a16, b16, g16 = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
        fused.BWD_WORKSPACE_BYTES = 24 << 20
        pc, ac, bc, gc = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    finally:
        fused.BWD_WORKSPACE_BYTES = saved
    # the forward's arithmetic does not depend on the plan
    assert torch.equal(p8, p16) and torch.equal(a8, a16) and torch.equal(b8, b16) and torch.equal(p8, p16s) and torch.equal(a8, a16s) and torch.equal(pc, p8)
    assert rel_err(a8.cpu(), a) < BF_OUT and rel_err(b8.cpu(), b) < BF_OUT and rel_err(p8.cpu(), pix) < BF_OUT
    worst = {}
    for k in g8o:
        # the recompute plan against the oracle that stages nothing: the bound of every bf16 test; the two stores against the oracles that stage as
        # they do: that bound, or the recompute plan's own distance + 1e-2 (a handful of ReLU mask flips -- pre-activations within rounding of zero,
        # summed in another order -- move a max-norm of ~1e3 random-signed samples by a few percent in either staging: tests/test_fp8_stage.py)
        e16 = rel_err(g16[k].cpu(), g16o[k])
        for name, got, want, bound in (("recompute", g16, g16o, 1.5 * BF_GRAD), ("8-bit store", g8, g8o, max(BF_GRAD, e16 + 1e-2)), ("bf16 store", g16s, g16so, max(BF_GRAD, e16 + 1e-2))):
            assert bool(torch.isfinite(got[k]).all()), (name, k)
            e = rel_err(got[k].cpu(), want[k])
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < bound, (name, k, e, e16)
        assert rel_err(gc[k], g8[k]) < 2e-6, k          # several ray chunks = one
    print(f"bf16 skip layer {R}x{S} F={F} early={early} late={late}: worst gradient distance from the emulating oracle " + ", ".join(f"{n} {v:.2e}" for n, v in worst.items()))


@pytest.mark.parametrize("F,early,late", [(32, 0, 2), (64, 4, 2), (128, 4, 2), (128, 2, 1)])
def test_bf16_points_with_a_skip_layer_vs_emulating_oracle(dev, F, early, late):
    """CPPN.forward on points in the bf16 mode (nca_mlp_fwd / nca_mlp_bwd): the net's output and parameter gradients against the oracle
    that rounds the MFMA operands to bf16 (the point backward recomputes: no store)."""
    from nerfca_amd import set_precision
    gen = torch.Generator().manual_seed(77 + F + early + late)
    ss = O.NetSpec(num_filters=F, num_early_layers=early, num_late_layers=late, num_time_dim=0)
    ps = O.init_params(ss, gen)
    win = O.freq_mask_alpha(12, 60000, 150000, 1)[0]
    N = 777
    x = (torch.rand(N, 3, generator=gen) * 2 - 1)
    gout = torch.randn(N, 1, generator=gen)
    pse = {k: v.clone().requires_grad_(True) for k, v in ps.items()}
    y = O.static_forward(pse, dataclasses.replace(ss, emulate_bf16=True), x, win)
    (y * gout).sum().backward()
    m = make_static(ps, dev, F=F, early=early, late=late)
    set_precision("bf16", m)
    m.update_freq_mask_alpha(60000, 150000)
    y2 = m(x.to(dev))
    assert rel_err(y2.cpu(), y.detach()) < BF_OUT
    (y2 * gout.to(dev)).sum().backward()
    got = grads_of(m)
    for k, v in pse.items():
        assert rel_err(got[k], v.grad) < BF_GRAD, k
