"""GPU parity of the bf16 kernels with RESIDENT weight images (NCA_OPT_RESIDENT_MIN_TILES): one net per launch, all of that
net's images loaded into LDS once per workgroup, no weight DMA / barrier in the tile loop; a two-net render takes two forward
launches, the second compositing with the sigma the first one wrote.  The arithmetic is that of the streaming kernels, so
every output, every byte the backward reads from the forward's store (seen through the gradients) and every gradient must be
BIT-identical with the option forced on (0) and off (-1) -- at sizes where the oracle tests of test_hip_parity.py /
test_fp8_stage.py pin the streaming kernels.  What the reference computes here: model/CPPN.py:127-166, model/Temporal.py:108-135
(the MLPs), train/model_helpers.py:131-154 (compositing)."""
import contextlib

import pytest
import torch

from conftest import nca_option
from oracle import nerfca_oracle as O
from test_hip_parity import make_dynamic, make_static
from test_recompute_bf16 import _hip_grads, _inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@contextlib.contextmanager
def count_launches(out):
    from nerfca_amd import _capi
    _capi.timing_reset()
    _capi.timing_enable(True)
    try:
        yield
    finally:
        out.append((_capi.timing_read("fwd")[1], _capi.timing_read("bwd_dgrad")[1]))
        _capi.timing_enable(False)
        _capi.timing_reset()


def _nets(dev, F, early, it_d, gen):
    from nerfca_amd import set_precision
    ss = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=0)
    sd = O.NetSpec(num_filters=F, num_early_layers=early, num_time_dim=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    s = make_static(ps, dev, F=F, early=early, late=0)
    t = make_dynamic(pd, dev, F=F, early=early, late=0, T=8)
    set_precision("bf16", s, t)
    s.update_freq_mask_alpha(75000, 150000)
    t.update_freq_mask_alpha(it_d, 150000)
    return s, t


# (F, early): 128 x 4 is the bench's net -- its five forward images (155 KiB) fit, and so do the four transposed images of the backward
# from the store (128 KiB).  fp8 = 0: no store -- the recompute backward carries both nets' forward AND transposed images: streaming, one launch
@pytest.mark.parametrize("R,S,F,early", [(8, 16, 32, 1), (33, 50, 64, 3), (64, 192, 128, 4), (7, 500, 128, 4), (300, 70, 128, 2), (40, 130, 64, 0)])
@pytest.mark.parametrize("it_d", [75000, 30000])
@pytest.mark.parametrize("fp8", [1, 0])
def test_resident_equals_streaming_render_and_gradients(dev, R, S, F, early, it_d, fp8):
    """Training path (forward with a store and backward from it; or, fp8 = 0, no store and the recompute backward) and the plain
    forward, resident forced vs never: bit-identical outputs and gradients; the forward really took two launches."""
    from nerfca_amd import render_rays
    gen = torch.Generator().manual_seed(4100 + R + S)
    s, t = _nets(dev, F, early, it_d, gen)
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    dists = O.composite(torch.zeros(R, S, 1), torch.zeros(R, S, 1), I0, d, z)[3]
    got, launches, plain = {}, [], {}
    for name, thr in (("streaming", -1), ("resident", 0)):
        with nca_option("RESIDENT_MIN_TILES", thr), nca_option("STAGE_FP8", fp8), nca_option("BF16_STORE", 0), count_launches(launches):
            got[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
        with nca_option("RESIDENT_MIN_TILES", thr), torch.no_grad():
            plain[name] = render_rays(s, t, o.to(dev), d.to(dev), ph.to(dev), I0.to(dev), z.to(dev), dists.to(dev))
    if early > 0:                               # (a net without hidden layers has no store: recompute backward, one launch)
        assert launches == [(1, 1), (2, 2 if fp8 else 1)], launches          # (the recompute backward is one streaming launch for both nets)
    for i in range(3):
        assert torch.equal(got["resident"][i], got["streaming"][i]), i
        assert torch.equal(plain["resident"][i], plain["streaming"][i]), i
    for k, v in got["streaming"][3].items():
        assert torch.equal(got["resident"][3][k], v), k


@pytest.mark.parametrize("R,S,F,early", [(16, 100, 128, 4), (9, 64, 64, 2)])
def test_resident_equals_streaming_with_depth_gradients(dev, R, S, F, early):
    """Resident forward and backward with the depth-gradient path (bf16 output-gradient blocks from an 8-bit staged store)."""
    gen = torch.Generator().manual_seed(4200 + R + S)
    s, t = _nets(dev, F, early, 75000, gen)
    o, d, ph, z, I0, cp, cs, cd = _inputs(R, S, gen)
    dists = O.composite(torch.zeros(R, S, 1), torch.zeros(R, S, 1), I0, d, z)[3]
    for want_depth in (False, True):
        got = {}
        for name, thr in (("streaming", -1), ("resident", 0)):
            with nca_option("RESIDENT_MIN_TILES", thr):
                got[name] = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd, want_depth=want_depth)
        for i in range(3):
            assert torch.equal(got["resident"][i], got["streaming"][i]), i
        for k, v in got["streaming"][3].items():
            assert torch.equal(got["resident"][3][k], v), k


@pytest.mark.parametrize("N,F,early", [(1000, 128, 4), (77, 64, 1)])
def test_resident_equals_streaming_point_queries(dev, N, F, early):
    """One net per call anyway: point queries of either net (CPPN.forward, Temporal.forward_composite)."""
    gen = torch.Generator().manual_seed(4300 + N)
    s, t = _nets(dev, F, early, 75000, gen)
    pts = (torch.rand(N, 3, generator=gen) * 2 - 1).to(dev)
    ph = torch.randint(0, 10, (N,), generator=gen).to(dev)
    out = {}
    for name, thr in (("streaming", -1), ("resident", 0)):
        with nca_option("RESIDENT_MIN_TILES", thr), torch.no_grad():
            out[name] = (s(pts), t.forward_composite(pts, ph))
    assert torch.equal(out["resident"][0], out["streaming"][0]) and torch.equal(out["resident"][1], out["streaming"][1])


def test_resident_default_threshold_and_option_validation(dev):
    from nerfca_amd import _capi
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert _capi.get_option(_capi.OPT_RESIDENT_MIN_TILES) in (8 * 8 * cus, -1, 0)      # (-1 / 0 when NCA_RESIDENT is set in the environment)
    with pytest.raises(RuntimeError):
        _capi.set_option(_capi.OPT_RESIDENT_MIN_TILES, -2)
