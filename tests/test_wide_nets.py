"""The GENERAL kernels (nerf-ca_amd/csrc/nca_wide.hpp): nets the fused kernels do not hold -- more than 128 units per layer, other channel counts
than 3 -> 1 (model/CPPN.py:40-65 takes any num_filters / num_input_channels / num_output_channels).

Reference values: tests/golden/wide.npz, written by tests/golden/make_golden.py (gen_wide) from the reference's own CPPN / Temporal /
obtain_train_predictions_iter.  CPU tests pin the oracle on them; GPU tests hold the library to them at 1e-5 (f32 parity mode) through the
drop-in model classes, and compare the general kernels with the fused kernels on nets both can run."""
import ctypes as C

import pytest
import torch

from conftest import rel_err
from oracle import nerfca_oracle as O

TOL = 1e-5
STATIC_CASES = [(136, 1, 0), (136, 1, 2), (256, 1, 0)]
CHANNEL_CASES = [(2, 3, "vanilla", 5), (4, 2, "fourier", 3), (1, 1, "none", 0)]


def spec_from(F, early, late, pos_enc="free_windowed", L=12, T=0, cin=3, cout=1, coef=None):
    return O.NetSpec(num_filters=F, num_early_layers=early, num_late_layers=late, num_input_channels=cin, num_output_channels=cout,
                     pos_enc=pos_enc, pos_enc_basis=L, pos_enc_window_start=1, num_time_dim=T, fourier_coefficients=coef)


# ------------------------------------------------------------------------------------------ the oracle on the reference's values (CPU)
@pytest.mark.parametrize("F,early,late", STATIC_CASES)
def test_oracle_wide_static(golden, F, early, late):
    g = golden("wide")
    tag = f"F{F}_e{early}_l{late}"
    spec = spec_from(F, early, late)
    params = {k: v.clone().requires_grad_(True) for k, v in g.prefixed(f"s_{tag}_p_").items()}
    assert list(params.keys()) == O.param_names(spec)
    y = O.static_forward(params, spec, g["x"], O.freq_mask_alpha(12, 60000, 150000, 1)[0])
    assert rel_err(y, g[f"s_{tag}_y"]) < 2e-6
    (y * g["gout"]).sum().backward()
    for k, gr in g.prefixed(f"s_{tag}_g_").items():
        assert rel_err(params[k].grad, gr) < TOL, k


def test_oracle_wide_dynamic(golden):
    g = golden("wide")
    spec = spec_from(136, 1, 0, T=8)
    params = {k: v.clone().requires_grad_(True) for k, v in g.prefixed("d_F136_p_").items()}
    y = O.dynamic_forward(params, spec, g["x"], g["ts"], O.freq_mask_alpha(12, 60000, 150000, 1)[0])
    assert rel_err(y, g["d_F136_y"]) < 2e-6
    (y * g["gout"]).sum().backward()
    for k, gr in g.prefixed("d_F136_g_").items():
        assert rel_err(params[k].grad, gr) < TOL, k


@pytest.mark.parametrize("cin,cout,enc,L", CHANNEL_CASES)
def test_oracle_other_channel_counts(golden, cin, cout, enc, L):
    g = golden("wide")
    tag = f"c{cin}to{cout}"
    spec = spec_from(48, 2, 1, enc, L, cin=cin, cout=cout, coef=g[f"{tag}_gauss"] * 2 if enc == "fourier" else None)
    params = {k: v.clone().requires_grad_(True) for k, v in g.prefixed(f"{tag}_p_").items()}
    y = O.static_forward(params, spec, g[f"{tag}_x"], None)
    assert y.shape == (150, cout) and rel_err(y, g[f"{tag}_y"]) < 2e-6
    (y * g[f"{tag}_gout"]).sum().backward()
    for k, gr in g.prefixed(f"{tag}_g_").items():
        assert rel_err(params[k].grad, gr) < TOL, k


def test_general_layout_header_compiles_and_counts_parameters(tmp_path):
    """nca_wide.hpp on the host compiler: parameter counts equal the module's, the packed image holds every padded weight, the biases and
    the output layer, and the natural -> padded column map of a skip layer puts the hidden part behind the padded encoded part."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    hdr = open(os.path.join(root, "nerf-ca_amd", "csrc", "nca_wide.hpp")).read()
    hdr = hdr[:hdr.index("// where a chunk's samples come from")]          # the layout half: no HIP types
    hdr = "\n".join(l for l in hdr.splitlines() if not l.startswith("#include") and not l.startswith("#pragma"))
    src.write_text(r'''
#include <cstdio>
#include "nca_layout.hpp"
''' + hdr + r'''
int main() {
    NcaNet n{}; n.F = 144; n.n_hidden = 1; n.n_late = 2; n.enc_mode = NCA_ENC_BANDS; n.L = 12;
    NcaWideLayout y; const char* why = "";
    if (nca_build_layout_wide(n, &y, &why)) { std::printf("ERR %s\n", why); return 1; }
    std::printf("%d %d %d %d %lld %d\n", y.n_params, y.K0, y.K0p, y.layer[2].Kp, (long long)y.packed_floats, nca_wide_col(y, y.layer[2], 75));
    n.reserved = NCA_NET_CHANNELS(2, 3); n.F = 48; n.n_late = 0; n.L = 5;
    if (nca_build_layout_wide(n, &y, &why)) { std::printf("ERR %s\n", why); return 1; }
    std::printf("%d %d %d %d\n", y.n_params, y.Kenc, nca_net_is_wide(n) ? 1 : 0, y.Cout);
    n.F = 130; std::printf("%d\n", nca_build_layout_wide(n, &y, &why));
    return 0;
}''')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "nerf-ca_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    K0, K0p = 75, 80          # 3 (1 + 2 * 12) encoded inputs, padded to 16
    n_params = (144 * K0 + 144) + (144 * 144 + 144) + (144 * (K0 + 144) + 144) + (144 * 144 + 144) + 144 + 1
    packed = 144 * K0p + 144 * 144 + 144 * (K0p + 144) + 144 * 144 + 4 * 144 + 144 + 1
    assert lines[0].split() == [str(n_params), str(K0), str(K0p), str(K0p + 144), str((packed + 3) // 4 * 4), str(K0p)]
    assert lines[1].split() == [str((48 * 22 + 48) + (48 * 48 + 48) + 3 * 48 + 3), "22", "1", "3"]
    assert int(lines[2]) == -2          # NCA_E_UNSUPPORTED: the library takes multiples of 16 (the host pads)


# ------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def model_def(F, early, late, pos_enc="free_windowed", L=12, T=0, gauss=None, sigma=2, cin=3, cout=1, device="cpu"):
    d = dict(num_early_layers=early, num_late_layers=late, num_filters=F, num_input_channels=cin, num_output_channels=cout,
             use_bias=True, pos_enc=pos_enc, pos_enc_window_start=1, pos_enc_basis=L, fourier_sigma=sigma,
             fourier_gaussian=gauss, act_func="relu", device=device)
    if T:
        d.update(num_input_times=1, use_time_latents=True, num_time_dim=T)
    return d


def make_static(params, dev, **kw):
    from nerfca_amd.model.CPPN import CPPN
    m = CPPN(model_def(device=dev, **kw))
    m.load_state_dict(params)
    return m.to(dev)


def make_dynamic(params, dev, **kw):
    from nerfca_amd.model.Temporal import Temporal
    m = Temporal(model_def(device=dev, **kw))
    m.load_state_dict(params)
    return m.to(dev)


def grads_of(model):
    return {k: p.grad.detach().cpu() for k, p in model.named_parameters()}


@pytest.mark.gpu
@pytest.mark.parametrize("F,early,late", STATIC_CASES)
def test_points_wide_static_vs_reference(golden, dev, F, early, late):
    from nerfca_amd import _capi
    g = golden("wide")
    tag = f"F{F}_e{early}_l{late}"
    m = make_static(g.prefixed(f"s_{tag}_p_"), dev, F=F, early=early, late=late)
    assert _capi.net_is_general(m._binding.net) and m._binding.net.F == (F + 15) // 16 * 16
    m.update_freq_mask_alpha(60000, 150000)
    y = m(g["x"].to(dev))
    assert y.shape == (200, 1) and rel_err(y.cpu(), g[f"s_{tag}_y"]) < TOL
    (y * g["gout"].to(dev)).sum().backward()
    got = grads_of(m)
    for k, ref in g.prefixed(f"s_{tag}_g_").items():
        assert got[k].shape == ref.shape and rel_err(got[k], ref) < TOL, k


@pytest.mark.gpu
def test_points_wide_dynamic_vs_reference(golden, dev):
    g = golden("wide")
    m = make_dynamic(g.prefixed("d_F136_p_"), dev, F=136, early=1, late=0, T=8)
    m.update_freq_mask_alpha(60000, 150000)
    y = m.forward_composite(g["x"].to(dev), g["ts"].to(dev))
    assert rel_err(y.cpu(), g["d_F136_y"]) < TOL
    (y * g["gout"].to(dev)).sum().backward()
    got = grads_of(m)
    for k, ref in g.prefixed("d_F136_g_").items():
        assert rel_err(got[k], ref) < TOL, k          # (time_latents: the per-phase sums through the one-hot columns of the input block)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,enc,L", CHANNEL_CASES)
def test_points_other_channel_counts_vs_reference(golden, dev, cin, cout, enc, L):
    g = golden("wide")
    tag = f"c{cin}to{cout}"
    m = make_static(g.prefixed(f"{tag}_p_"), dev, F=48, early=2, late=1, pos_enc=enc, L=L, gauss=g[f"{tag}_gauss"], sigma=2, cin=cin, cout=cout)
    y = m(g[f"{tag}_x"].to(dev))
    assert y.shape == (150, cout) and rel_err(y.cpu(), g[f"{tag}_y"]) < TOL
    (y * g[f"{tag}_gout"].to(dev)).sum().backward()
    got = grads_of(m)
    for k, ref in g.prefixed(f"{tag}_g_").items():
        assert rel_err(got[k], ref) < TOL, k


@pytest.mark.gpu
@pytest.mark.parametrize("Fs,Fd", [(136, 136), (64, 136)])
def test_composite_render_with_wide_nets_vs_reference(golden, dev, Fs, Fd):
    """obtain_train_predictions_iter with nets beyond 128 units: both on the general kernels, or a fused-kernel static net beside a general
    dynamic net -- each leaves its raw field, one compositing kernel follows; the backward goes the same way back."""
    from nerfca_amd.train import model_helpers as MH
    g = golden("wide")
    tag = f"rays_s{Fs}_d{Fd}"
    s = make_static(g.prefixed(f"{tag}_sp_"), dev, F=Fs, early=1, late=0)
    t = make_dynamic(g.prefixed(f"{tag}_dp_"), dev, F=Fd, early=1, late=0, T=8)
    for m in (s, t):
        m.update_freq_mask_alpha(75000, 150000)
    S = g[f"{tag}_z"].shape[0]
    phs = g[f"{tag}_ph"][:, None].repeat(1, S).to(dev)
    res = MH.obtain_train_predictions_iter(s, t, None, None, g[f"{tag}_o"].to(dev), g[f"{tag}_d"].to(dev), phs, g[f"{tag}_I0"].to(dev),
                                           g[f"{tag}_z"].to(dev), "softplus", 32768, 0, dev, t_rand=g[f"{tag}_t_rand"])
    pix, sig_s, sig_d = res[0], res[1], res[2]
    for v, n in ((pix, "pix"), (sig_s, "sig_s"), (sig_d, "sig_d")):
        ref = g[f"{tag}_{n}"]
        assert v.dtype == ref.dtype and tuple(v.shape) == tuple(ref.shape) and rel_err(v.cpu(), ref) < TOL, n
    (pix.sum() + 30 * sig_s.sum() + 20 * sig_d.sum()).backward()
    for m, pre in ((s, "sg_"), (t, "dg_")):
        got = grads_of(m)
        for k, ref in g.prefixed(f"{tag}_{pre}").items():
            assert rel_err(got[k], ref) < TOL, (pre, k)


def _force_general(model):
    from nerfca_amd import _capi
    b = model._binding
    b.net.reserved |= _capi.NET_GENERAL
    b.packed = None
    return model


@pytest.mark.gpu
@pytest.mark.parametrize("F,late,T", [(32, 0, 0), (64, 2, 0), (128, 0, 8)])
def test_general_kernels_equal_fused_kernels_on_nets_both_run(dev, F, late, T):
    """NCA_NET_GENERAL sends a net of the fused kernels' range through the general ones: same values and gradients to f32 rounding, many chunks
    or one (a workspace cap of 3 MB leaves 128-row chunks: the chunked sums only re-associate)."""
    from nerfca_amd import fused as FZ
    gen = torch.Generator().manual_seed(5)
    spec = spec_from(F, 2, late, T=T)
    p = O.init_params(spec, gen)
    N = 1000
    x = (torch.rand(N, 3, generator=gen) * 2 - 1).to(dev)
    ts = torch.randint(0, 10, (N,), generator=gen).to(dev)
    go = torch.randn(N, 1, generator=gen).to(dev)

    def run(general, cap=None):
        mk = make_dynamic if T else make_static
        m = mk(p, dev, F=F, early=2, late=late, **({"T": T} if T else {}))
        m.update_freq_mask_alpha(60000, 150000)
        if general:
            _force_general(m)
        old = FZ.BWD_WORKSPACE_BYTES
        if cap:
            FZ.BWD_WORKSPACE_BYTES = cap
        try:
            y = m.forward_composite(x, ts) if T else m(x)
            (y * go).sum().backward()
        finally:
            FZ.BWD_WORKSPACE_BYTES = old
        return y.detach().cpu(), grads_of(m)

    y0, g0 = run(False)
    for cap in (None, 3 << 20):
        y1, g1 = run(True, cap)
        assert rel_err(y1, y0) < TOL
        for k in g0:
            assert rel_err(g1[k], g0[k]) < 2 * TOL, (cap, k)
    ya, ga = run(True)
    yb, gb = run(True)
    assert torch.equal(ya, yb) and all(torch.equal(ga[k], gb[k]) for k in ga)          # run to run: the same bits


@pytest.mark.gpu
def test_general_kernels_refusals(golden, dev):
    """What the general kernels do not do is refused by name, never approximated: bf16 mode, a width that is not a multiple of 16 at the C ABI, a forward
    without a workspace (depth gradients and per-point latent gradients likewise: nca_api_wide.inc)."""
    from nerfca_amd import _capi, fused as FZ
    g = golden("wide")
    m = make_static(g.prefixed("s_F256_e1_l0_p_"), dev, F=256, early=1, late=0)
    m.update_freq_mask_alpha(60000, 150000)
    FZ.set_precision("bf16", m)
    with pytest.raises(_capi.NcaError, match="bf16 mode runs nets of up to 128 units"):
        m(g["x"].to(dev))
    FZ.set_precision("f32", m)
    lib = _capi.lib()
    net = _capi.NcaNet(F=130, n_hidden=1, n_late=0, enc_mode=_capi.ENC_BANDS, L=12, T=0, P=0, reserved=0)
    assert lib.nca_param_count(C.byref(net)) == -2 and b"multiple of 16" in lib.nca_last_error()
    # nca_mlp_fwd has no workspace argument: a general net is sent to nca_mlp_fwd_ws
    b = m._binding
    packed = b.ensure_packed()
    x = g["x"].to(dev).contiguous()
    raw = torch.empty(200, device=dev)
    win, _ = m._enc_buffers()
    rc = lib.nca_mlp_fwd(C.byref(b.net), b.prec, packed.data_ptr(), win.data_ptr(), None, b.flat.data_ptr(), 200, x.data_ptr(), None, raw.data_ptr(), None)
    assert rc == -4 and b"nca_mlp_fwd_ws" in lib.nca_last_error()


@pytest.mark.gpu
def test_trainer_steps_with_wide_nets(dev):
    """CompositeTrainer over nets of 136 units (general kernels, width 144): the autograd step, the fused step and the graph-replayed step all
    train, agree on the first step's loss, the loss falls, and the padding units stay at zero."""
    from nerfca_amd import _capi, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    data = synthetic.make_dataset(16, 48, dev, views=synthetic.TRAIN_VIEWS[:2], n_phases=3, F=32)
    first, last = {}, {}
    for mode in ("autograd", "fused", "graph"):
        torch.manual_seed(5)
        sdef, tdef = synthetic.net_definitions(dev, F=136)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        cfg = TrainConfig(depth_samples_per_ray_coarse=48, img_sample_size=160)
        tr = CompositeTrainer(cfg, s, t, data, dev, seed=3, fused_loss=(mode != "autograd"))
        step = tr.step_graph if mode == "graph" else tr.step
        for it in range(6):
            loss, _, _ = step(1000 + it)
            if it == 0:
                first[mode] = float(loss.detach())
        assert torch.isfinite(loss)
        last[mode] = float(loss.detach())
        for m in (s, t):
            bnd = m._binding
            assert bnd.net.F == 144 and _capi.net_is_general(bnd.net) and bnd._is_flat()
            mask = torch.ones_like(bnd.flat, dtype=torch.bool)
            for g in bnd.split_grads(mask):
                g.fill_(False)
            assert bool(mask.any()) and float(bnd.flat[mask].abs().max()) == 0.0
    assert abs(first["fused"] - first["autograd"]) <= 1e-5 * abs(first["autograd"]), first
    assert abs(first["graph"] - first["fused"]) <= 1e-6 * abs(first["fused"]), first
    assert all(last[m] < first[m] for m in first), (first, last)


@pytest.mark.gpu
@pytest.mark.parametrize("R,S", [(7, 40), (1500, 192)])
def test_general_forward_store_equals_recompute(dev, R, S):
    """Both nets on the general kernels: the forward leaves every layer's output in a store (NCA_STORE_GENERAL: what autograd keeps in the reference) and
    the backward recomputes nothing; without a store (fused.STORE_FORWARD_LIMIT_BYTES = 0) it recomputes per run of whole rays.  Same runs, same split
    sums: the gradients are the same BITS either way -- one run (7 x 40) or two (1 500 x 192 > 2^18 samples)."""
    from nerfca_amd import _capi, fused as FZ, render_rays
    gen = torch.Generator().manual_seed(11)
    ps = O.init_params(spec_from(144, 1, 0), gen)
    pd = O.init_params(spec_from(144, 1, 0, T=8), gen)
    o = (torch.rand(R, 3, generator=gen) * 0.1 + torch.tensor([3.0, -2.0, 2.5])).double().to(dev)
    d = (torch.rand(R, 3, generator=gen) - 0.5).double().to(dev)
    ph = torch.randint(0, 10, (R,), generator=gen).to(dev)
    z = torch.linspace(3.4, 5.6, S).to(dev)
    dists = torch.cat([z[1:] - z[:-1], torch.tensor([1e-10], device=dev)]).double()
    I0 = torch.full((R,), 2.16, device=dev)

    def run(limit):
        s = make_static(ps, dev, F=144, early=1, late=0)
        t = make_dynamic(pd, dev, F=144, early=1, late=0, T=8)
        for m in (s, t):
            m.update_freq_mask_alpha(75000, 150000)
        old = FZ.STORE_FORWARD_LIMIT_BYTES
        FZ.STORE_FORWARD_LIMIT_BYTES = limit
        try:
            pix, a, b = render_rays(s, t, o, d, ph, I0, z, dists)
            fmt = _capi.last_plan()["fwd_store_format"]
            (pix.sum() + 30 * a.sum() + 20 * b.sum()).backward()
        finally:
            FZ.STORE_FORWARD_LIMIT_BYTES = old
        return fmt, pix.detach().cpu(), {**{"s." + k: v for k, v in grads_of(s).items()}, **{"d." + k: v for k, v in grads_of(t).items()}}

    f1, p1, g1 = run(96 << 30)
    f0, p0, g0 = run(0)
    assert f1 == _capi.STORE_GENERAL and f0 == _capi.STORE_NONE
    assert torch.equal(p1, p0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k


EDGE_CASES = [
    # (F, early, late, enc, L, T, N, bias)
    (16, 0, 0, "none", 0, 0, 1, True),               # one layer, one point, the smallest width
    (144, 0, 1, "free_windowed", 4, 0, 129, True),   # the skip layer is the only hidden-width layer and the last one
    (160, 2, 3, "fourier", 5, 0, 300, True),         # skip in the middle, fourier features
    (1024, 1, 0, "free_windowed", 12, 0, 200, True), # the widest net
    (208, 1, 0, "vanilla", 10, 16, 257, True),       # latents: 16 of them, 63 + 16 + 10 one-hot columns -> two 16-column pads
    (136, 2, 0, "free_windowed", 12, 8, 500, False), # no biases (zero gaps in the flat buffer), a width the host pads (144)
]


@pytest.mark.gpu
@pytest.mark.parametrize("F,early,late,enc,L,T,N,bias", EDGE_CASES)
def test_general_kernels_edge_shapes_vs_oracle(dev, F, early, late, enc, L, T, N, bias):
    """Shapes at the edges of the general kernels -- one layer, one point, 1 024 units, a skip layer that is the last layer, 16 latents, no biases, widths the
    host pads -- against the oracle in f32 (values 1e-5; gradients 1e-5 or three times the f32 oracle's own distance from the f64 oracle, as everywhere)."""
    from nerfca_amd import _capi
    gen = torch.Generator().manual_seed(100 + F)
    coef = torch.randn(3 * max(L, 1), generator=gen) if enc == "fourier" else None
    spec = spec_from(F, early, late, enc, L, T=T, coef=coef * 2 if coef is not None else None)
    p = O.init_params(spec, gen)
    if not bias:
        p = {k: v for k, v in p.items() if not k.endswith(".bias")}
    x = (torch.rand(N, 3, generator=gen) * 2 - 1)
    ts = torch.randint(0, 10, (N,), generator=gen).int()
    go = torch.randn(N, 1, generator=gen)
    win = O.freq_mask_alpha(L, 60000, 150000, 1)[0] if enc == "free_windowed" else None

    def oracle(dt):
        pp = {k: v.clone().to(dt).requires_grad_(True) for k, v in p.items()}
        full = dict(pp)
        if not bias:          # the oracle's layers take a bias: zeros
            for k, shp in O.param_shapes(spec).items():
                if k not in full:
                    full[k] = torch.zeros(shp, dtype=dt)
        w = win.to(dt) if win is not None else None
        y = O.dynamic_forward(full, spec, x.to(dt), ts, w) if T else O.static_forward(full, spec, x.to(dt), w)
        (y * go.to(dt)).sum().backward()
        return y.detach(), {k: v.grad for k, v in pp.items()}

    y32, g32 = oracle(torch.float32)
    y64, g64 = oracle(torch.float64)
    kw = dict(F=F, early=early, late=late, pos_enc=enc, L=L, gauss=coef, sigma=2)
    d = model_def(device=dev, **kw, **({"T": T} if T else {}))
    d["use_bias"] = bias
    if T:
        from nerfca_amd.model.Temporal import Temporal as M
    else:
        from nerfca_amd.model.CPPN import CPPN as M
    m = M(d)
    m.load_state_dict(p)
    m = m.to(dev)
    if F <= 128:
        _force_general(m)
    assert _capi.net_is_general(m._binding.net)
    if enc == "free_windowed":
        m.update_freq_mask_alpha(60000, 150000)
    y = m.forward_composite(x.to(dev), ts.to(dev)) if T else m(x.to(dev))
    assert rel_err(y.cpu(), y64) < max(TOL, 3 * rel_err(y32, y64))
    (y * go.to(dev)).sum().backward()
    got = grads_of(m)
    for k, ref in g64.items():
        assert rel_err(got[k], ref) < max(TOL, 3 * rel_err(g32[k], ref)), k


@pytest.mark.gpu
def test_single_field_and_two_wide_widths_vs_oracle(dev):
    """render_volume_density's single-field render through a 256-unit net, and a composite render of a 144-unit static net with a 256-unit dynamic net (both on
    the general kernels, different widths: the store holds one block per net), against the f64 oracle."""
    from nerfca_amd import _capi, render_rays
    gen = torch.Generator().manual_seed(77)
    R, S = 9, 48
    ss, sd = spec_from(144, 1, 0), spec_from(256, 1, 0, T=8)
    ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
    win = O.freq_mask_alpha(12, 75000, 150000, 1)[0]
    o = (torch.rand(R, 3, generator=gen) * 0.1 + torch.tensor([3.0, -2.0, 2.5])).double()
    d = (torch.rand(R, 3, generator=gen) - 0.5).double()
    ph = torch.randint(0, 10, (R,), generator=gen)
    z = O.stratified_depths(O.depth_values(3.4259, 5.5741, S), torch.rand(S, generator=gen))
    I0 = torch.full((R,), 2.15991)

    def oracle(dt, single):
        pso = {k: v.clone().to(dt).requires_grad_(True) for k, v in ps.items()}
        pdo = {k: v.clone().to(dt).requires_grad_(True) for k, v in pd.items()}
        pts = O.query_points(o, d, z).to(dt)
        raw_s = O.static_forward(pso, ss, pts, win.to(dt)).reshape(R, S, -1)
        if single:
            out = O.composite_single(raw_s, I0.to(dt), d, z.to(dt))
            (out[0].sum() + 30 * out[1].sum()).backward()
            return out, pso, None
        raw_d = O.dynamic_forward(pdo, sd, pts, ph[:, None].repeat(1, S).flatten(), win.to(dt)).reshape(R, S, -1)
        out = O.composite(raw_s, raw_d, I0.to(dt), d, z.to(dt))
        (out[0].sum() + 30 * out[1].sum() + 20 * out[2].sum()).backward()
        return out, pso, pdo

    for single in (True, False):
        o32, s32, d32 = oracle(torch.float32, single)
        o64, s64, d64 = oracle(torch.float64, single)
        s = make_static(ps, dev, F=144, early=1, late=0)
        t = make_dynamic(pd, dev, F=256, early=1, late=0, T=8)
        for m in (s, t):
            m.update_freq_mask_alpha(75000, 150000)
        dists = o64[-1].to(dev)
        if single:
            pix, a = render_rays(s, None, o.to(dev), d.to(dev), ph.to(dev), I0.to(dev), z.to(dev), dists, single=True)
            (pix.sum() + 30 * a.sum()).backward()
            outs = (pix, a)
        else:
            pix, a, b = render_rays(s, t, o.to(dev), d.to(dev), ph.to(dev), I0.to(dev), z.to(dev), dists)
            assert _capi.last_plan()["fwd_store_format"] == _capi.STORE_GENERAL
            (pix.sum() + 30 * a.sum() + 20 * b.sum()).backward()
            outs = (pix, a, b)
        for v, r64, r32 in zip(outs, o64, o32):
            assert rel_err(v.cpu(), r64) < max(TOL, 3 * rel_err(r32, r64))
        for m, g64, g32 in ((s, s64, s32),) + (() if single else ((t, d64, d32),)):
            for k, prm in m.named_parameters():
                assert rel_err(prm.grad.cpu(), g64[k].grad) < max(TOL, 3 * rel_err(g32[k].grad, g64[k].grad)), (single, k)
