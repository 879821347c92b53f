"""GPU tests of the device-side step preparation (ABI 11): the Philox batch sampler (nca_draw_ray_ids / nca_draw_uniform) against its
NumPy restatement bit for bit, nca_begin_step against nca_prepare_batch and the host schedules, the launch-fusing entry points
(nca_pack_weights2, the loss kernel forming pix from the forward's ray sums, Adam's in-kernel tick), and the graph-replayed step in
DEVICE mode against the host-launched step.  Reference: train/run_composite.py:250-281 (sampling, schedules), model/CPPN.py:144-159,
train/model_helpers.py:3-12, 264-269."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import philox_ref as P
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _tables(n_rays=5000, n_var=700, seed=0):
    rng = np.random.default_rng(seed)
    var = np.sort(rng.choice(n_rays, n_var, replace=False))
    return var, np.setdiff1d(np.arange(n_rays), var)


@pytest.mark.parametrize("R,n_var", [(1024, 512), (1000, 125), (65536, 32768), (7, 3), (4096, 0)])
def test_sampler_matches_numpy_restatement(dev, R, n_var):
    """ids and jitter of the library's sampler == tests/philox_ref.py, bit for bit; a slot range drawn alone == that range of the whole
    batch (what a rank of a sharded step draws); exactly n_var variance ids; deterministic; another iteration differs."""
    from nerfca_amd.fused import BatchSampler
    var, non = _tables()
    smp = BatchSampler(11, R, n_var, var if n_var else None, non if n_var else None, 5000, dev)
    for it in (0, 17, 75000, (1 << 33) + 5):
        ids = smp.ray_ids(it)
        want = P.ray_ids(11, it, R, n_var, var if n_var else None, non if n_var else None, 5000)
        assert ids.dtype == torch.int64 and np.array_equal(ids.cpu().numpy(), want), it
        if n_var:
            assert int(np.isin(want, var).sum()) == n_var
        lo, hi = R // 3, R // 3 + max(1, R // 4)
        assert torch.equal(smp.ray_ids(it, lo, hi - lo), ids[lo:hi])
        assert torch.equal(ids, smp.ray_ids(it))
        t = smp.uniform(it, 500)
        assert t.dtype == torch.float32 and np.array_equal(t.cpu().numpy(), P.uniform(11, it, 500))
    assert not torch.equal(smp.ray_ids(1), smp.ray_ids(2))
    # the iteration from a device counter: n_iter + *iter_dev
    from nerfca_amd import _capi
    ctr = torch.tensor([40], dtype=torch.int64, device=dev)
    out = torch.empty(R, dtype=torch.int64, device=dev)
    d = smp.desc(2, ctr)
    _capi.check(_capi.lib().nca_draw_ray_ids(C.byref(d), 0, R, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    assert torch.equal(out, smp.ray_ids(42))


def test_sampler_argument_errors(dev):
    from nerfca_amd import _capi
    from nerfca_amd.fused import BatchSampler
    var, non = _tables()
    smp = BatchSampler(1, 64, 32, var, non, 5000, dev)
    with pytest.raises(_capi.NcaError, match="outside the global batch"):
        smp.ray_ids(0, 60, 8)
    bad = BatchSampler(1, 64, 0, None, None, 0, dev)
    with pytest.raises(_capi.NcaError, match="n_rows"):
        bad.ray_ids(0)


def _host_schedule(cfg, n_iter, L=12, start=1):
    from nerfca_amd.schedules import freq_mask, linear_param_decay
    c = cfg
    win = freq_mask(L, n_iter, c.static_pos_enc_window_decay_steps, start)[0]
    w = [linear_param_decay(n_iter, c.favor_s_weight_start, c.favor_s_weight_end, c.hyperparam_decay_steps, c.favor_s_weight_delay_steps),
         linear_param_decay(n_iter, c.dynamic_entro_weight_start, c.dynamic_entro_weight_end, c.hyperparam_decay_steps),
         linear_param_decay(n_iter, c.occl_weight_start, c.occl_weight_end, c.hyperparam_decay_steps, c.favor_s_weight_delay_steps),
         linear_param_decay(n_iter, c.l1_weight_start, c.l1_weight_end, c.hyperparam_decay_steps)]
    return win, torch.tensor([float(x) for x in w], dtype=torch.float64)


@pytest.mark.parametrize("R,S", [(1, 2), (37, 5), (1024, 500), (65536, 192), (100, 1000)])
def test_begin_step_equals_prepare_batch_and_host_schedules(dev, R, S):
    """nca_begin_step == nca_draw_ray_ids + nca_draw_uniform + nca_prepare_batch (bit for bit: o, d, gt, w, phases, z, dists), and its band
    windows / loss weights == the host schedules (schedules.freq_mask, linear_param_decay) bit for bit, at iterations around every
    breakpoint of the schedules (delay, decay end, whole pointer values, 0)."""
    from nerfca_amd import _capi, fused
    from nerfca_amd.train.trainer import TrainConfig
    g = torch.Generator().manual_seed(R * 31 + S)
    N = 4 * R + 3
    table = torch.randn((N, 4, 3), generator=g, dtype=torch.float64).to(dev)
    phases = torch.randint(0, 10, (N,), generator=g).to(dev)
    depth = torch.linspace(2.0, 6.0, S).to(dev)
    var = np.arange(0, N, 3)
    non = np.setdiff1d(np.arange(N), var)
    smp = fused.BatchSampler(5, R, R // 2, var, non, N, dev)
    cfg = TrainConfig(static_pos_enc_window_decay_steps=1500, hyperparam_decay_steps=1000, favor_s_weight_delay_steps=400)
    L = 12
    win = torch.zeros(L, dtype=torch.float32, device=dev)
    wts = torch.zeros(4, dtype=torch.float64, device=dev)
    sch = _capi.NcaSchedules()
    sch.n_windows = 1
    sch.window[0].kind, sch.window[0].L, sch.window[0].window_start, sch.window[0].decay_steps, sch.window[0].out = _capi.WINDOW_FREE, L, 1, 1500, win.data_ptr()
    for k, (a, b, delay) in enumerate(((cfg.favor_s_weight_start, cfg.favor_s_weight_end, 400), (cfg.dynamic_entro_weight_start, cfg.dynamic_entro_weight_end, 0),
                                       (cfg.occl_weight_start, cfg.occl_weight_end, 400), (cfg.l1_weight_start, cfg.l1_weight_end, 0))):
        sch.weight[k].start, sch.weight[k].end, sch.weight[k].steps, sch.weight[k].delay = a, b, 1000, delay
    sch.weights_out = wts.data_ptr()
    its = [0, 1, 124, 125, 126, 399, 400, 401, 999, 1000, 1001, 1399, 1400, 1499, 1500, 1501, 75000] if R == 37 else [0, 125, 777, 1500]
    for it in its:
        (o, d, gt, w, ph, z, dists), ids, t = fused.begin_step(smp, it, 0, R, table, phases, depth, schedules=sch, want_draws=True)
        assert torch.equal(ids, smp.ray_ids(it)) and torch.equal(t, smp.uniform(it, S))
        ref = fused.prepare_batch(ids, table, phases, depth, t)
        for x, y in zip((o, d, gt, w, ph, z, dists), ref):
            assert x.dtype == y.dtype and torch.equal(x, y)
        hw, hwts = _host_schedule(cfg, it, L, 1)
        assert torch.equal(win.cpu(), hw), (it, win.cpu(), hw)
        assert torch.equal(wts.cpu(), hwts), (it, wts.cpu(), hwts)
    # a rank's slice of the slots; injected ids and jitter pass through untouched; the iteration from a device counter
    lo, n = R // 2, R - R // 2
    part = fused.begin_step(smp, 9, lo, n, table, phases, depth)
    whole = fused.begin_step(smp, 9, 0, R, table, phases, depth)
    assert torch.equal(part[0], whole[0][lo:]) and torch.equal(part[4], whole[4][lo:]) and torch.equal(part[5], whole[5])
    ids_in = torch.randint(0, N, (R,), generator=g).to(dev)
    t_in = torch.rand(S, generator=g)
    inj = fused.begin_step(smp, 9, 0, R, table, phases, depth, ids_in=ids_in, t_rand_in=t_in)
    for x, y in zip(inj, fused.prepare_batch(ids_in, table, phases, depth, t_in)):
        assert torch.equal(x, y)
    ctr = torch.tensor([7], dtype=torch.int64, device=dev)
    viac = fused.begin_step(smp, 2, 0, R, table, phases, depth, iter_dev=ctr)
    for x, y in zip(viac, whole):
        assert torch.equal(x, y)
    # an id outside the table is clamped and counted, never read
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    ids_bad = ids_in.clone()
    ids_bad[0] = N + 5
    fused.begin_step(smp, 9, 0, R, table, phases, depth, ids_in=ids_bad, t_rand_in=t_in, bad_ids=bad)
    assert int(bad.item()) == 1


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_pack_weights2_equals_two_packs(dev, prec):
    from nerfca_amd import set_precision, synthetic
    from nerfca_amd.fused import FieldBinding
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    torch.manual_seed(4)
    sdef, tdef = synthetic.net_definitions(dev, F=64)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    set_precision(prec, s, t)
    one_s, one_d = s._binding.ensure_packed().clone(), t._binding.ensure_packed().clone()
    s._binding.packed = t._binding.packed = None
    two_s, two_d = FieldBinding.ensure_packed_pair(s._binding, t._binding)
    assert torch.equal(one_s, two_s) and torch.equal(one_d, two_d)


def test_adam_ticks_in_kernel(dev):
    """The last workgroup of nca_adam_step increments the step count (and the training iteration), leaves its arrival counter at zero."""
    from nerfca_amd import synthetic
    from nerfca_amd.fused import FusedAdam
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    torch.manual_seed(3)
    sdef, tdef = synthetic.net_definitions(dev, F=128)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    it = torch.tensor([75000], dtype=torch.int64, device=dev)
    adam = FusedAdam([t, s], lr=1e-3, iter_counter=it)
    for k in range(5):
        adam.step([torch.ones_like(b.flat) for b in adam.bindings])
        assert adam._step.cpu().tolist() == [k + 1, 0] and int(it.item()) == 75001 + k


def test_loss_kernel_forms_pix_from_ray_sums(dev):
    """A forward called with want_pix=False leaves its per-tile ray sums to the loss kernel: terms, gradients and pix bit-identical to the
    forward's own pix kernel followed by the loss kernel; terms_f32 = the terms rounded to f32."""
    from nerfca_amd import fused, set_precision, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import TrainConfig
    for prec, R, S in (("f32", 50, 70), ("bf16", 300, 500)):
        torch.manual_seed(6)
        sdef, tdef = synthetic.net_definitions(dev, F=64)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        set_precision(prec, s, t)
        for m in (s, t):
            m.update_freq_mask_alpha(75000, 150000)
        g = torch.Generator().manual_seed(1)
        o = (torch.rand(R, 3, generator=g) * 0.2 + torch.tensor([3.0, -2.0, 2.5])).double().to(dev)
        d = (torch.rand(R, 3, generator=g) - 0.5).double().to(dev)
        ph = torch.randint(0, 10, (R,), generator=g).to(dev)
        z = torch.linspace(3.4, 5.6, S).to(dev)
        dists = torch.cat([z[1:] - z[:-1], torch.full((1,), 1e-10, device=dev)]).double()
        I0 = torch.full((R,), 2.16, device=dev)
        gt, w = torch.rand(R, generator=g).double().to(dev) + 1.5, torch.rand(R, generator=g).double().to(dev) + 0.5
        batch = fused._RayBatch(o, d, ph, I0, z, dists, "softplus", False, 1e-2)
        cfg = TrainConfig()
        wts = (1e-3, 1e-3, 1e-2, 1e-3)
        pix, a, b, _ = fused.render_forward_raw(batch, s._binding, t._binding)
        one = fused.fused_losses(pix, gt, w, a, b, dists, cfg, wts)
        sums, a2, b2, _ = fused.render_forward_raw(batch, s._binding, t._binding, want_pix=False)
        assert isinstance(sums, fused.RaySums) and torch.equal(a, a2) and torch.equal(b, b2)
        pix2 = torch.empty(R, dtype=torch.float64, device=dev)
        t32 = torch.empty(13, dtype=torch.float32, device=dev)
        two = fused.fused_losses(sums, gt, w, a2, b2, dists, cfg, wts, pix_out=pix2, terms_f32=t32)
        assert torch.equal(pix, pix2)
        for x, y in zip(one, two):
            assert torch.equal(x, y)
        assert torch.equal(t32, one[0].to(torch.float32))


def _trainer(dev, prec, R, S, seed=5, world=1, rank=0, **cfgkw):
    from nerfca_amd import set_precision, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    data = _trainer.data.get((S,))
    if data is None:
        data = _trainer.data[(S,)] = synthetic.make_dataset(16, S, dev, views=synthetic.TRAIN_VIEWS[:2], n_phases=3, F=32)
    torch.manual_seed(9)
    sdef, tdef = synthetic.net_definitions(dev, F=64)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    set_precision(prec, s, t)
    kw = dict(depth_samples_per_ray_coarse=S, img_sample_size=R, favor_s_weight_delay_steps=4, l1_weight_start=1e-3, l1_weight_end=1e-5,
              occl_weight_start=1e-2, occl_weight_end=1e-4, dynamic_entro_weight_start=1e-3, favor_s_weight_start=1e-3, entro_mask_thre=1e-6,
              hyperparam_decay_steps=40, lr=5e-3, lr_decay_steps=6, lr_end_factor=0.1, static_pos_enc_window_decay_steps=40, temp_pos_enc_window_decay_steps=40)
    kw.update(cfgkw)
    return CompositeTrainer(TrainConfig(**kw), s, t, data, dev, rank=rank, world=world, seed=seed, fused_loss=True)


_trainer.data = {}


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_device_mode_graph_step_is_bit_identical_to_the_host_launched_step(dev, prec):
    """The graph-replayed step in DEVICE mode (ids, jitter, windows, loss weights made by nca_begin_step from a device counter; nothing
    copied per step) against the host-launched fused step with the library's Adam: the flat gradient of EVERY step and the parameters after
    ten steps are bit-identical -- consecutive iterations, then a jump (the counter is re-set), across the schedules' breakpoints."""
    from nerfca_amd.fused import FusedAdam
    its = list(range(0, 8)) + [38, 39, 40, 41]
    # host-launched: fused_gradients + FusedAdam (the same optimiser arithmetic as the graph's)
    tr = _trainer(dev, prec, 512, 48)
    adam = FusedAdam([tr.t, tr.s], lr=tr.cfg.lr, end_factor=tr.cfg.lr_end_factor, total_iters=tr.cfg.lr_decay_steps)
    host = []
    for it in its:
        terms, gs, gd = tr.fused_gradients(it)
        host.append((terms.clone(), gd.clone(), gs.clone()))
        adam.step([gd, gs])
    p_host = torch.cat([b.flat for b in adam.bindings]).clone()
    tg = _trainer(dev, prec, 512, 48)
    for k, it in enumerate(its):
        loss, pixel, terms = tg.step_graph(it)
        assert tg._device_mode
        nd, ns = host[k][1].numel(), host[k][2].numel()
        assert torch.equal(terms, host[k][0]), (it, terms, host[k][0])
        assert torch.equal(tg._flat[:nd], host[k][1]) and torch.equal(tg._flat[nd:nd + ns], host[k][2]), it
        assert torch.equal(tg._flat[nd + ns:], host[k][0].to(torch.float32))
    assert torch.equal(torch.cat([b.flat for b in tg.adam.bindings]), p_host)
    assert int(tg.adam.step_count.item()) == len(its) and int(tg._iter_dev.item()) == its[-1] + 1
    tg.check_ray_ids()


def test_record_mode_still_serves_injected_draws(dev):
    """A trainer whose draws are replaced (a test replaying the reference's own ids / jitter) replays its graph in RECORD mode -- ids and
    the pinned record copied per step -- and equals the host-launched step on the same draws."""
    outs = []
    for graph in (False, True):
        tr = _trainer(dev, "f32", 256, 48)
        g = torch.Generator().manual_seed(3)
        ids = [torch.randint(0, tr.data.rays_train.shape[0], (256,), generator=g) for _ in range(4)]
        jit = [torch.rand(48, generator=g) for _ in range(4)]
        tr.draw_ray_ids_device = lambda n, ids=ids: ids[n % 4].to(dev)
        tr.draw_jitter = lambda n, jit=jit: jit[n % 4]
        losses = [float((tr.step_graph(it) if graph else tr.step_fused(it))[0]) for it in range(4)]
        if graph:
            assert not tr._device_mode
        outs.append((losses, torch.cat([p.detach().flatten() for p in tr.params]).cpu()))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert abs(a - b) <= 1e-5 * abs(a)
    assert rel_err(outs[1][1], outs[0][1]) < 1e-5


def test_host_launched_step_raises_on_a_ray_id_outside_the_table(dev):
    """run_composite.py:262: NumPy raises IndexError on an id outside the ray table.  The library clamps and counts; the host-launched
    steps read the counter every step and raise (NERFCA_STRICT=0 defers to the syncing calls)."""
    from nerfca_amd import _capi
    tr = _trainer(dev, "f32", 64, 48)
    n = tr.data.rays_train.shape[0]
    tr.draw_ray_ids_device = lambda it: torch.full((64,), n + 3, dtype=torch.int64, device=dev)
    with pytest.raises(_capi.NcaError, match="outside the ray table"):
        tr.step(0)
    tr2 = _trainer(dev, "f32", 64, 48)
    tr2.strict_ids = False
    tr2.draw_ray_ids_device = lambda it: torch.full((64,), -1, dtype=torch.int64, device=dev)
    tr2.step(0)                                   # deferred ...
    with pytest.raises(_capi.NcaError, match="outside the ray table"):
        tr2.early_stop()                          # ... to the next syncing call


def test_dropin_losses_accept_what_the_reference_accepts(dev):
    """compute_losses with FEWER pixel weights than rays (the reference fills weighted_mask[:len] and leaves the rest 0,
    train/model_helpers.py:216-219) and weighted_MSELoss with broadcastable shapes (its expression broadcasts, :287): values and gradients
    against the reference's torch expressions (tests/injected_trainer.py restates compute_losses from the part functions)."""
    from injected_trainer import all_terms
    from nerfca_amd.train import model_helpers as MH
    from nerfca_amd.train.trainer import TrainConfig
    g = torch.Generator().manual_seed(5)
    R, S = 40, 30
    cfg = TrainConfig(entro_mask_thre=0.5, entro_weighted_thresh=0.03)
    a = (torch.rand(R, S, generator=g) * 0.2).to(dev).requires_grad_(True)
    b = (torch.rand(R, S, generator=g) * 0.2).to(dev).requires_grad_(True)
    dists = (torch.rand(S, generator=g) * 0.01 + 0.01).double().to(dev)
    w_short = (torch.rand(R - 13, generator=g) * 0.2 + 0.95).double().to(dev)          # some above 1.03, some below
    got = MH.compute_losses(a, b, dists, w_short, cfg)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    want = all_terms(a2, b2, dists, w_short, cfg)
    for x, y in zip(got, want):
        assert rel_err(x.detach().cpu(), y.detach().cpu()) < 1e-6
    (got[6] * 3.0 + got[3]).backward()
    (want[6] * 3.0 + want[3]).backward()
    assert rel_err(b.grad.cpu(), b2.grad.cpu()) < 1e-5 and rel_err(a.grad.cpu(), a2.grad.cpu()) < 1e-5
    p = torch.rand(R, generator=g).double().to(dev).requires_grad_(True)
    gt = torch.rand(R, generator=g).double().to(dev)
    wscalar = torch.tensor([1.7], dtype=torch.float64, device=dev)
    out = MH.weighted_MSELoss()(p, gt, wscalar)
    assert out.shape == (R,) and torch.allclose(out, (p - gt) ** 2 * wscalar, rtol=1e-14, atol=0)
    out.mean().backward()
    assert torch.allclose(p.grad, (2 * (p - gt) * wscalar / R).detach(), rtol=1e-12, atol=0)
