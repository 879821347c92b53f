"""GPU tests of the overlapped backward (NCA_OPT_OVERLAP_CUS, include/nerfca_hip.h): in the bf16 mode's backward from the 8-bit staged
store the static net's weight-gradient launch runs BESIDE the dynamic net's dgrad launch on a second stream of the library, the
dynamic net's weight gradient after the join.  What must hold:

* everything the forward returns is untouched (bit-identical to the plain plan);
* the gradients are those of the plain plan up to the order of the sample sums (the value of the option fixes the split of each
  job's samples over waves) -- and bit-identical run to run for a given value, forked or not;
* a captured HIP graph carries the fork and the join: the graph-replayed step follows the host-launched one.

The serial chain this replaces: loss.backward() of train/run_composite.py:283-308 (the reference's autograd runs the two nets'
backward passes one after the other on one stream)."""
import pytest
import torch

from conftest import nca_option
from oracle import nerfca_oracle as O
from test_recompute_bf16 import _hip_grads, _inputs
from test_resident_bf16 import _nets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _run(dev, s, t, inputs, dists, ovl, scope_opts=None):
    from nerfca_amd import fused
    o, d, ph, z, I0, cp, cs, cd = inputs
    with fused.PlanScope(resident_min_tiles=0, overlap_cus=ovl, **(scope_opts or {})) as sc:
        out = _hip_grads(s, t, dev, o, d, ph, I0, z, dists, cp, cs, cd)
    torch.cuda.synchronize()
    return out, sc.decided()


@pytest.mark.parametrize("R,S", [(1024, 192), (700, 500)])
def test_overlapped_backward_equals_plain_backward(dev, R, S):
    gen = torch.Generator().manual_seed(5100 + R)
    s, t = _nets(dev, 128, 4, 75000, gen)
    inputs = _inputs(R, S, gen)
    o, d, ph, z, I0 = inputs[:5]
    dists = O.composite(torch.zeros(R, S, 1), torch.zeros(R, S, 1), I0, d, z)[3]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    base, plan0 = _run(dev, s, t, inputs, dists, 0)
    assert plan0["overlap_cus"] == 0 and plan0["bwd_kernel_mode"] == 5 and plan0["bwd_resident"] == 1
    for ovl in (cus // 4, cus // 2, cus - 64):
        got, plan = _run(dev, s, t, inputs, dists, ovl)
        assert plan["overlap_cus"] == ovl and plan["overlap_forked"] == 1, plan
        again, _ = _run(dev, s, t, inputs, dists, ovl)
        for i in range(3):
            assert torch.equal(got[i], base[i]), i            # pix, sigma_s, sigma_d: the forward is untouched
        for k, v in base[3].items():
            ref = float(v.abs().max())
            err = float((got[3][k] - v).abs().max())
            assert err <= 2e-5 * ref + 1e-30, (ovl, k, err, ref)       # the same products, another summation order
            assert torch.equal(again[3][k], got[3][k]), (ovl, k)       # bit-identical run to run


def test_overlapped_backward_over_several_ray_chunks(dev):
    """A workspace budget that cuts the batch into ray chunks: every chunk forks and joins, the slabs accumulate."""
    from nerfca_amd import fused
    R, S = 2048, 192
    gen = torch.Generator().manual_seed(5200)
    s, t = _nets(dev, 128, 4, 75000, gen)
    inputs = _inputs(R, S, gen)
    o, d, ph, z, I0 = inputs[:5]
    dists = O.composite(torch.zeros(R, S, 1), torch.zeros(R, S, 1), I0, d, z)[3]
    saved = fused.BWD_WORKSPACE_BYTES
    try:
        fused.BWD_WORKSPACE_BYTES = 1 << 30
        base, p0 = _run(dev, s, t, inputs, dists, 0)
        assert p0["chunks"] == 1
        fused.BWD_WORKSPACE_BYTES = 420 << 20
        got, p1 = _run(dev, s, t, inputs, dists, 96)
        assert p1["chunks"] >= 2 and p1["overlap_cus"] == 96 and p1["overlap_forked"] == 1, p1
    finally:
        fused.BWD_WORKSPACE_BYTES = saved
    for k, v in base[3].items():
        ref = float(v.abs().max())
        assert float((got[3][k] - v).abs().max()) <= 2e-5 * ref + 1e-30, k


def test_small_batches_and_other_paths_do_not_fork(dev):
    """Below the size the overlapped plan is for (and on every path but mode 5 with resident images) the option changes nothing."""
    R, S = 64, 192
    gen = torch.Generator().manual_seed(5300)
    s, t = _nets(dev, 128, 4, 75000, gen)
    inputs = _inputs(R, S, gen)
    o, d, ph, z, I0 = inputs[:5]
    dists = O.composite(torch.zeros(R, S, 1), torch.zeros(R, S, 1), I0, d, z)[3]
    base, p0 = _run(dev, s, t, inputs, dists, 0)
    got, p1 = _run(dev, s, t, inputs, dists, 96)
    assert p1["overlap_cus"] == 0, p1
    for k, v in base[3].items():
        assert torch.equal(got[3][k], v), k
    got, p2 = _run(dev, s, t, inputs, dists, 96, {"stage_fp8": 0, "bf16_store": 0})           # the recompute backward: one launch, nothing to overlap
    assert p2["overlap_cus"] == 0 and p2["bwd_kernel_mode"] == 1, p2


def test_graph_replayed_step_with_overlap_follows_the_host_launched_step(dev):
    """The fork and the join are events on the capture stream: the captured graph holds both branches.  Same trajectory as the
    host-launched step (whose Adam is torch's: 1e-3 as in test_graph_step_matches_eager_step), and two graph-replayed trainers
    agree bit for bit."""
    from nerfca_amd import set_precision, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    S, R = 96, 4096
    data = synthetic.make_dataset(32, S, dev, views=synthetic.TRAIN_VIEWS[:2], n_phases=3, F=32)
    outs = []
    for graph in (False, True, True):
        torch.manual_seed(9)
        sdef, tdef = synthetic.net_definitions(dev)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        set_precision("bf16", s, t)
        cfg = TrainConfig(depth_samples_per_ray_coarse=S, img_sample_size=R, lr_decay_steps=6, lr_end_factor=0.1,
                          static_pos_enc_window_decay_steps=40, temp_pos_enc_window_decay_steps=40)
        tr = CompositeTrainer(cfg, s, t, data, dev, seed=5, fused_loss=True, plan_opts={"resident_min_tiles": 0, "overlap_cus": 96})
        losses = []
        for it in range(6):
            out = tr.step_graph(3 * it) if graph else tr.step_fused(3 * it)
            losses.append(float(out[0].detach()))
        plan = tr.plan()
        assert plan["overlap_cus"] == 96 and plan["overlap_forked"] == 1, plan
        outs.append((losses, torch.cat([p.detach().flatten() for p in tr.params]).cpu()))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert abs(a - b) <= 1e-3 * abs(a), (outs[0][0], outs[1][0])
    assert float((outs[1][1] - outs[0][1]).abs().max() / outs[0][1].abs().max()) < 1e-3
    assert outs[1][0] == outs[2][0] and torch.equal(outs[1][1], outs[2][1])
