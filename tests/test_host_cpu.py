"""CPU tests of the host-side mirror of the reference interface and of the C-ABI library surface.

No compute call goes to the library here (there is no GPU): we check that it loads, exports every
symbol of include/nerfca_hip.h, answers the pure-host queries, and that unsupported configurations
are explicit errors.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_err


def model_def(F=128, early=4, late=0, pos_enc="free_windowed", L=12, T=0, gauss=None):
    d = dict(num_early_layers=early, num_late_layers=late, num_filters=F, num_input_channels=3, num_output_channels=1,
             use_bias=True, pos_enc=pos_enc, pos_enc_window_start=1, pos_enc_basis=L, fourier_sigma=2,
             fourier_gaussian=gauss, act_func="relu", device="cpu")
    if T:
        d.update(num_input_times=1, use_time_latents=True, num_time_dim=T)
    return d


# ----------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    from nerfca_amd import _capi
    header = open(os.path.join(ROOT, "include", "nerfca_hip.h")).read()
    declared = set(re.findall(r"\b(nca_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = C.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in nerfca_hip.h but not exported"
    assert declared == set(_capi.SYMBOLS), "ctypes table and header disagree"
    import re as _re
    assert _capi.lib().nca_abi_version() == int(_re.search(r"#define NCA_ABI_VERSION (\d+)", header).group(1))


def test_param_count_and_packed_size_match_reference_nets():
    from nerfca_amd import _capi
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    s, t = CPPN(model_def()), Temporal(model_def(T=8))
    assert sum(p.numel() for p in s.parameters()) == 75905
    assert sum(p.numel() for p in t.parameters()) == 77009      # a7
    lib = _capi.lib()
    assert lib.nca_param_count(C.byref(s._binding.net)) == 75905
    assert lib.nca_param_count(C.byref(t._binding.net)) == 77009
    assert lib.nca_packed_bytes(C.byref(s._binding.net), _capi.PREC_F32) > 4 * 75905


def test_unsupported_configs_are_explicit_errors():
    from nerfca_amd import _capi
    lib = _capi.lib()
    bad = _capi.NcaNet(F=48, n_hidden=4, n_late=0, enc_mode=1, L=12, T=0, P=0, reserved=0)
    assert lib.nca_param_count(C.byref(bad)) == -2
    assert b"num_filters" in lib.nca_last_error()
    late_dyn = _capi.NcaNet(F=64, n_hidden=1, n_late=2, enc_mode=1, L=4, T=4, P=10, reserved=0)
    assert lib.nca_param_count(C.byref(late_dyn)) == -2
    with pytest.raises(_capi.NcaError):
        _capi.check(lib.nca_packed_bytes(C.byref(bad), 0))


def test_fused_path_refuses_cpu_tensors():
    from nerfca_amd import _capi
    from nerfca_amd.model.CPPN import CPPN
    m = CPPN(model_def(F=32, early=1))
    m.update_freq_mask_alpha(1, 10)
    with pytest.raises(_capi.NcaError, match="GPU"):
        m(torch.zeros(4, 3))


# ----------------------------------------------------------------------------- drop-in module surface
def test_state_dict_keys_and_save_blob(golden, tmp_path):
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    g = golden("checkpoint_keys")
    s, t = CPPN(model_def(late=2)), Temporal(model_def(T=8))
    assert list(s.state_dict().keys()) == list(g.np("static_late2_keys"))
    assert list(t.state_dict().keys()) == list(g.np("temporal_keys"))
    assert [str(tuple(v.shape)) for v in t.state_dict().values()] == list(g.np("temporal_shapes"))
    for name, m in (("static", s), ("temporal", t)):
        m.update_freq_mask_alpha(10, 100)
        f = tmp_path / f"{name}.pth"
        m.save(str(f), {"note": 1})
        blob = torch.load(str(f), weights_only=False)
        assert list(blob.keys()) == list(g.np(f"{name}_save_keys"))
        assert blob["version"] == str(g.np(f"{name}_save_version"))


def test_same_seed_same_init_as_reference(golden):
    """Layers are created in the reference's order, so a torch seed reproduces its initial weights."""
    from nerfca_amd.model.CPPN import CPPN
    g = golden("mlps")
    torch.manual_seed(1000 + 128 + 4 * 7 + 0)
    m = CPPN(model_def(F=128, early=4, late=0))
    for k, v in g.prefixed("s_F128_e4_l0_p_").items():
        assert torch.equal(m.state_dict()[k], v), k


def test_parameters_are_views_of_one_flat_buffer():
    from nerfca_amd.model.Temporal import Temporal
    t = Temporal(model_def(T=8))
    b = t._binding
    assert b.flat.numel() == 77009
    names = [n for n, _ in t.named_parameters()]
    assert names[0] == "time_latents" and names[-1] == "output_linear.0.bias"
    off = 0
    for p in t.parameters():
        assert p.data_ptr() == b.flat.data_ptr() + 4 * off
        off += p.numel()
    with torch.no_grad():
        t.time_latents.add_(1.0)
    assert torch.equal(b.flat[:80].view(10, 8), t.time_latents.detach())
    t.to(torch.device("cpu"))  # _apply re-flattens
    assert t._binding._is_flat()


def test_schedules_and_windows(golden):
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.schedules import linear_param_decay
    g = golden("schedules")
    m = CPPN(model_def())
    for it, mask, a in zip(g.np("free_its"), g.np("free_masks"), g.np("free_alphas")):
        m.update_freq_mask_alpha(int(it), 150000)
        assert np.array_equal(m.freq_mask_alpha.numpy(), mask) and float(m.windowed_alpha) == float(a)
    its = g.np("decay_iters")
    assert np.array_equal(np.array([linear_param_decay(int(i), 1e-12, 1e-10, 100000, delay_steps=40000) for i in its], dtype=np.float64), g.np("decay_favor"))
    gp = golden("posenc")
    m = CPPN(model_def(pos_enc="nerfies_windowed"))
    for a in (0.0, 3.3, 12.0):
        m.windowed_alpha = a
        assert torch.equal(m.windowed_pos_enc(12, "pts"), gp[f"nerfies_window_a{a}"])
        assert torch.equal(m.pos_enc(gp["x"], 12, "pts"), gp[f"nerfies_a{a}"])
    m = CPPN(model_def())
    with pytest.raises(AttributeError):
        m._band_window()  # free_windowed before the first update_freq_mask_alpha (SURVEY 8a notes)


def test_temporal_quirks_raise_like_reference():
    from nerfca_amd.model.Temporal import Temporal
    d = model_def(T=8)
    d["use_time_latents"] = False
    with pytest.raises(UnboundLocalError):
        Temporal(d).forward_composite(torch.zeros(2, 3), torch.zeros(2))


# ----------------------------------------------------------------------------- data loader + geometry
VIEWS = [[-30, 30], [-30, -30], [60, -30], [60, 30], [-5, 40]]


def test_ray_geometry_and_table(golden, tmp_path):
    from nerfca_amd.train import data_helpers as DH, proj_helpers as PH
    g = golden("geometry")
    geo = dict(DSD=25.0, DSO=4.5, nDetector=[16, 16], dDetector=[2.0 / 16, 2.0 / 16], offDetector=[0.0, 0.0])
    assert np.array_equal(PH.source_matrix_tigre(np.array([0, 0, -4.5]), -30, 30), g.np("n16_pose_v0"))
    for i, (th, ph) in enumerate(VIEWS):
        ro, rd = PH.get_ray_values_tigre(th, ph, 0, geo, "cpu")
        assert np.array_equal(ro, g.np(f"n16_v{i}_o")) and np.array_equal(rd, g.np(f"n16_v{i}_d"))
    W = H = 5
    geo = dict(DSD=25.0, DSO=4.5, nDetector=[W, H], dDetector=[0.3, 0.5], offDetector=[0.05, -0.1])
    frames = []
    for k, (v, p) in enumerate(zip(g.np("table_views"), g.np("table_phase_in"))):
        fp, wp = tmp_path / f"i{k}.npy", tmp_path / f"v{k}.npy"
        np.save(fp, g.np("table_imgs")[k])
        np.save(wp, g.np("table_vars")[k])
        frames.append(dict(theta=float(v[0]), phi=float(v[1]), larm=0, file_path=str(fp), weighted_file_path=str(wp),
                           img_min_max=[0.2, 1.7], heart_phase=int(p)))
    rays, phases = DH.prepare_data_for_loader_tigre(frames, geo, W, H, 8, 0.5, "cpu")
    assert rays.dtype == np.float64 and np.array_equal(rays, g.np("table_rays"))
    assert phases.dtype == g.np("table_phases").dtype and np.array_equal(phases, g.np("table_phases"))
    assert torch.equal(DH.create_depth_values(3.4259, 5.5741, 192, "cpu"), golden("depth")["z"])


def test_randomize_depth_matches_reference_draw(golden):
    from nerfca_amd.train import model_helpers as MH
    g = golden("depth")
    torch.manual_seed(303)
    assert torch.equal(MH.randomize_depth(g["z"], "cpu"), g["z_jit"])       # same CPU generator stream
    assert torch.equal(MH.randomize_depth(g["z"], "cpu", t_rand=g["t_rand"]), g["z_jit"])


# ----------------------------------------------------------------------------- losses / render helpers
@pytest.mark.parametrize("dtn", ["f64", "f32"])
def test_losses_match_reference(golden, dtn):
    """The torch restatement of the loss functions (nerfca_amd/losses.py: the part functions the API exports; tests/injected_trainer.py:
    all_terms / weighted_sq_err built from them for the CPU tests of the data-parallel bookkeeping) against the reference's goldens.  The
    drop-in compute_losses / weighted_MSELoss are HIP-backed: tests/test_hip_parity.py::test_dropin_compute_losses_is_the_hip_kernel."""
    from types import SimpleNamespace
    import injected_trainer as IT
    from nerfca_amd._capi import NcaError
    from nerfca_amd.train import model_helpers as MH
    g = golden("losses")
    args = SimpleNamespace(favor_s_opt=None, skewness_val=1.0, entro_mask_thre=1e-4, entro_use_weighting=True,
                           entro_weighted_thresh=0.03, occl_reg_perc=0.2)
    a = g[f"{dtn}_sig_s"].clone().requires_grad_(True)
    b = g[f"{dtn}_sig_d"].clone().requires_grad_(True)
    with pytest.raises(NcaError, match="GPU"):          # no torch implementation behind the drop-in names
        MH.compute_losses(a, b, g[f"{dtn}_dists"], g[f"{dtn}_wpix"], args)
    with pytest.raises(NcaError, match="GPU"):
        MH.weighted_MSELoss()(g[f"{dtn}_mse_pred"], g[f"{dtn}_mse_gt"], g[f"{dtn}_wpix"])
    res = IT.all_terms(a, b, g[f"{dtn}_dists"], g[f"{dtn}_wpix"], args)
    names = ["blendw", "sig_s_max", "sig_d_max", "favor", "s_ent", "s_sum", "d_ent", "d_sum", "occl", "l1", "l2"]
    for n, v in zip(names, res):
        assert rel_err(v, g[f"{dtn}_{n}"]) < 1e-6, n
    (0.7 * res[3] + 1.3 * res[4] + 0.9 * res[6] + 0.5 * res[8] + 0.25 * res[9] + 2.0 * res[10]).backward()
    assert rel_err(a.grad, g[f"{dtn}_g_sig_s"]) < 1e-6 and rel_err(b.grad, g[f"{dtn}_g_sig_d"]) < 1e-6
    assert torch.equal(IT.weighted_sq_err(g[f"{dtn}_mse_pred"], g[f"{dtn}_mse_gt"], g[f"{dtn}_wpix"]), g[f"{dtn}_mse"])
    assert rel_err(MH.compute_occl_loss(b, g[f"{dtn}_dists"], 0.2, use_back=True), g[f"{dtn}_occl_back"]) < 1e-6


def test_render_helpers_refuse_cpu_tensors(golden):
    """render_volume_density[_composite] are HIP kernels (values: tests/test_hip_parity.py against the same goldens); there
    is no torch implementation behind them, so CPU tensors are an explicit error."""
    from nerfca_amd._capi import NcaError
    from nerfca_amd.train import model_helpers as MH
    g = golden("render")
    dirs = torch.zeros(12, 3, dtype=torch.float64)
    with pytest.raises(NcaError, match="GPU"):
        MH.render_volume_density_composite(g["raw_s"], g["raw_d"], g["I0"], dirs, g["z"], "softplus")
    with pytest.raises(NcaError, match="GPU"):
        MH.render_volume_density(g["raw_s"], g["I0"], dirs, g["z"], "softplus")
    assert torch.equal(MH._interval_lengths(g["z"], dirs), g["comp_f64_softplus_dists"])      # the one piece of host arithmetic
    assert torch.equal(MH._interval_lengths(g["z"], dirs.float()), g["comp_f32_softplus_dists"])


def test_sample_pdf_matches_oracle():
    from nerfca_amd.train import model_helpers as MH
    from oracle import nerfca_oracle as O
    gen = torch.Generator().manual_seed(3)
    bins = torch.sort(torch.rand(6, 15, generator=gen), -1)[0]
    w = torch.rand(6, 14, generator=gen)
    u = torch.rand(6, 9, generator=gen)
    assert torch.equal(MH.sample_pdf(bins, w, 9, "cpu", u=u), O.sample_pdf(bins, w, u))


def test_precision_switch_is_per_model_pair():
    import nerfca_amd
    from nerfca_amd import _capi
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    s, t = CPPN(model_def(F=32, early=1)), Temporal(model_def(F=32, early=1, T=4))
    assert s._binding.prec == _capi.PREC_F32
    nerfca_amd.set_precision("bf16", s, t)
    assert s._binding.prec == t._binding.prec == _capi.PREC_BF16
    lib = _capi.lib()
    assert lib.nca_packed_bytes(C.byref(s._binding.net), _capi.PREC_BF16) > 0
    # a CPPN with a skip layer runs in both precisions since ABI 11 (two images for the skip layer: encoded part + hidden part)
    late = CPPN(model_def(F=32, early=1, late=2))
    nb16, nb32 = lib.nca_packed_bytes(C.byref(late._binding.net), _capi.PREC_BF16), lib.nca_packed_bytes(C.byref(late._binding.net), _capi.PREC_F32)
    plain16 = lib.nca_packed_bytes(C.byref(CPPN(model_def(F=32, early=2, late=0))._binding.net), _capi.PREC_BF16)         # the same number of F-wide layers without the skip input
    assert nb32 > 0 and nb16 > plain16 > 0
    # configurations the bf16 kernels do not implement are explicit errors, not fallbacks: more than 16 latent dimensions
    wide_t = _capi.NcaNet(F=32, n_hidden=1, n_late=0, enc_mode=1, L=4, T=20, P=10, reserved=0)
    assert lib.nca_packed_bytes(C.byref(wide_t), _capi.PREC_BF16) == -2 and b"bf16" in lib.nca_last_error()
    assert lib.nca_packed_bytes(C.byref(wide_t), _capi.PREC_F32) > 0


def test_graft_entry_build_contract():
    """__graft_entry__.build() must succeed here (hipcc cross-compiles without a GPU) and agree with the header's ABI
    version; the Makefile must create the output directory itself (it is git-ignored, so absent in a fresh clone)."""
    import __graft_entry__ as G
    G.build()
    header = open(os.path.join(ROOT, "include", "nerfca_hip.h")).read()
    ver = int(re.search(r"#define NCA_ABI_VERSION (\d+)", header).group(1))
    from nerfca_amd import _capi
    assert _capi.lib().nca_abi_version() == ver
    assert "mkdir -p" in open(os.path.join(ROOT, "Makefile")).read()


def test_checkpoint_round_trip(tmp_path):
    """export.load_checkpoint rebuilds what CPPN.save / Temporal.save wrote (the reference has no loader): same class,
    same state dict, same window state, training_information passed through."""
    from nerfca_amd.export import load_checkpoint
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    for cls, kw in ((CPPN, {}), (Temporal, {"T": 8})):
        torch.manual_seed(3)
        m = cls(model_def(**kw))
        m.update_freq_mask_alpha(1234, 5000)
        path = tmp_path / f"{cls.__name__}.pth"
        m.save(str(path), {"n_iter": 1234})
        back, info = load_checkpoint(str(path))
        assert type(back) is cls and info == {"n_iter": 1234}
        assert list(back.state_dict().keys()) == list(m.state_dict().keys())
        for (k, a), b in zip(m.state_dict().items(), back.state_dict().values()):
            assert torch.equal(a, b), k
        assert torch.equal(back.freq_mask_alpha, m.freq_mask_alpha)      # what save() keeps for free_windowed
        assert back._binding._is_flat()          # the loaded parameters are views of one flat buffer again


def test_three_way_bf16_split_is_exact_and_accurate():
    """The arithmetic claim behind nca_wgrad_f32x3 (nerf-ca_amd/csrc/nca_kernels_f32.hip): x = x1 + x2 + x3 with three bf16
    pieces obtained by round-to-nearest-even and exact subtraction reproduces every normal f32 exactly, and the six piece
    products of weight <= 2^-16, summed in wider precision, are at least as close to the exact product sum as an f32 dot
    product is."""
    rng = np.random.default_rng(0)

    def rne_bf16(x):
        b = x.view(np.uint32).astype(np.uint64)
        b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
        return b.astype(np.uint32).view(np.float32)

    def split(x):
        p1 = rne_bf16(x)
        r1 = x - p1
        p2 = rne_bf16(r1)
        r2 = r1 - p2
        return p1, p2, rne_bf16(r2), r2

    v = (rng.standard_normal(1_000_000) * np.exp(rng.uniform(-30, 30, 1_000_000))).astype(np.float32)
    p1, p2, p3, r2 = split(v)
    assert np.array_equal(p3, r2)                                              # the third residual needs no rounding
    assert np.array_equal(p1.astype(np.float64) + p2 + p3, v.astype(np.float64))
    a = rng.standard_normal((2048, 48)).astype(np.float32)
    b = rng.standard_normal((2048, 48)).astype(np.float32)
    A, B = split(a)[:3], split(b)[:3]
    six = sum(A[i].astype(np.float64).T @ B[j].astype(np.float64) for i, j in ((1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)))
    exact = a.astype(np.float64).T @ b.astype(np.float64)
    err6 = np.abs(six - exact).max() / np.abs(exact).max()
    err32 = np.abs((a.T @ b).astype(np.float64) - exact).max() / np.abs(exact).max()
    assert err6 < 2e-7 and err6 <= err32


def test_on_disk_schema_loader_and_log_records(tmp_path):
    """The loop glue of train/run_composite.py:65-125 from the reference's on-disk schema (general.json, train-*.json, test-*.json,
    .npy images): ray table / phases as prepare_data_for_loader_tigre builds them, the variance-ray split of the importance sampler
    (:97-99), one held-out view; and the reference's log keys (:314-344, :393-404) as JSON lines."""
    import json
    from types import SimpleNamespace
    from nerfca_amd import synthetic
    from nerfca_amd.train import data_helpers as DH
    from nerfca_amd.train import run_log as RL
    W = H = 6
    geo = synthetic.xcat_geometry(W)
    geo = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in geo.items()}
    geo["nDetector"] = [W, H]
    rng = np.random.default_rng(0)
    frames = []
    for i, (theta, phi, ph) in enumerate([(0.1, 0.2, 0), (0.1, 0.2, 4), (1.3, -0.4, 7)]):
        img = rng.random((W * H,))
        wgt = 1.0 + rng.random((W * H,)) * (i > 0)             # the first image has no high-variance rays
        np.save(tmp_path / f"img{i}.npy", img)
        np.save(tmp_path / f"w{i}.npy", wgt)
        frames.append({"theta": theta, "phi": phi, "larm": 0, "file_path": str(tmp_path / f"img{i}.npy"), "weighted_file_path": str(tmp_path / f"w{i}.npy"),
                       "img_min_max": [0.0, 2.0], "heart_phase": ph, "image_id_str": f"im{i}"})
    (tmp_path / "general.json").write_text(json.dumps(geo))
    (tmp_path / "train.json").write_text(json.dumps({"frames": frames[:2]}))
    (tmp_path / "test.json").write_text(json.dumps({"frames": frames[2:] + frames[:1]}))
    args = SimpleNamespace(depth_samples_per_ray_coarse=8, weighted_loss_max=1.0, var_sample_thre=3.0)
    d = DH.load_training_data(str(tmp_path / "general.json"), str(tmp_path / "train.json"), str(tmp_path / "test.json"), args, "cpu")
    rays, phases = DH.prepare_data_for_loader_tigre(frames[:2], geo, W, H, 8, 1.0, "cpu")
    assert d.rays_train.dtype == torch.float64 and tuple(d.rays_train.shape) == (2 * W * H, 4, 3)
    assert np.array_equal(d.rays_train.numpy(), rays) and np.array_equal(d.phases_train.numpy(), phases)
    assert np.array_equal(d.var_ray_ids, np.argwhere(rays[:, -1, 0] > 1.03).flatten()) and d.var_ray_ids.min() >= W * H
    assert len(d.var_ray_ids) + len(d.non_var_ray_ids) == 2 * W * H and np.intersect1d(d.var_ray_ids, d.non_var_ray_ids).size == 0
    assert d.test_img_indices == ["im2"] and d.test_phase == 7 and tuple(d.test_origins.shape) == (W * H, 3) and d.test_origins.dtype == torch.float32
    want = DH.denormalize_image(np.load(frames[2]["file_path"]), W, H, [0.0, 2.0]).reshape(-1)
    assert np.allclose(d.test_image.numpy(), want.astype(np.float32))
    # log records under the reference's keys
    from nerfca_amd import _capi
    from nerfca_amd.train.trainer import TrainConfig
    tr = SimpleNamespace(cfg=TrainConfig(), s=SimpleNamespace(windowed_alpha=6.0), t=SimpleNamespace(windowed_alpha=5.5))
    tr.loss_weights = lambda n: (1e-12, 1e-10, 1e-8, 1e-8)
    terms = torch.arange(1, 14, dtype=torch.float64) * 1e-3
    rec = RL.train_record(tr, 100, terms, start_time=0.0)
    ref_keys = {"train_loss", "train_psnr", "train_pixel_loss_coarse", "train_pixel_loss_fine", "train_blendw", "train_sigma_s_max", "train_sigma_d_max",
                "train_favor_s_loss", "train_s_entropy_loss", "train_d_entropy_loss", "train_s_entropy_sum", "train_d_entropy_sum", "train_d_occl_loss",
                "train_s_l1", "train_s_l2", "favor_s_weight", "dynamic_entro_weight", "occl_weight", "l1_weight", "train_time", "train_static_windowed",
                "train_temp_windowed"}
    assert set(rec) == ref_keys
    assert rec["train_loss"] == 1e-3 and abs(rec["train_psnr"] - 30.0) < 1e-9 and rec["train_d_entropy_loss"] == terms[_capi.TERM_NAMES.index("d_entropy")]
    log = RL.JsonlLogger(str(tmp_path / "log.jsonl"))
    log.log(rec, step=100)
    log.log(RL.test_record({k: torch.tensor(0.5) for k in ("test_loss", "test_psnr", "test_pixel_loss_coarse", "test_favor_s_loss", "test_blendw",
                                                          "test_s_entropy_loss", "test_d_entropy_loss", "pred")}), step=100)
    lines = [json.loads(l) for l in (tmp_path / "log.jsonl").read_text().splitlines()]
    assert len(lines) == 2 and lines[0]["step"] == 100 and set(lines[1]) == {"test_loss", "test_psnr", "test_pixel_loss_coarse", "test_favor_s_loss",
                                                                              "test_blendw", "test_s_entropy_loss", "test_d_entropy_loss", "step"}


def test_shipped_library_is_not_a_timing_build():
    """Timing-only libraries (round 3's tools/elim_build.sh, built from the tag r03-kernels by tools/r03_experiments.sh: NCA_EXP != 0,
    kernels that leave work out, results wrong by construction), A/B variants and rounding-ablation builds say what they are in
    nca_build_info().  The library the package loads must be the product build."""
    from nerfca_amd import _capi
    info = _capi.build_info()
    assert "NCA_EXP=0" in info and f"abi={_capi.ABI_VERSION}" in info and "gfx950" in info, info
    if "NERFCA_LIB" not in os.environ:          # (an A/B variant under test says what it is)
        assert info.endswith("variant=0x800 ablation=0x0"), info          # no rounding ablation (tools/ablation_build.sh); 8 waves per workgroup, NCA_BF_PIPE2 / NCA_WGRAD_TR / NCA_ONCHIP_NR / NCA_BF_PIPE = 0
    assert _capi.get_option(_capi.OPT_STAGE_FP8) in (-1, 0, 1) and _capi.get_option(_capi.OPT_STAGE_FP8_MIN_TILES) >= 0
    assert _capi.get_option(_capi.OPT_WGRAD_REBUILD_WEIGHT_PCT) == int(os.environ.get("NCA_WGRAD_W", 115))
    with pytest.raises(_capi.NcaError):
        _capi.set_option(_capi.OPT_WGRAD_REBUILD_WEIGHT_PCT, 99)
    with pytest.raises(_capi.NcaError):
        _capi.get_option(99)


def test_flat_buffer_of_padded_and_biasless_nets():
    """Host logic of FieldBinding (no GPU): a net of 48 units is laid out at the kernels' width 64 -- every parameter the leading block
    of its padded matrix, as a view into ONE flat buffer in the library's natural order whose size is what the descriptor expects,
    padding zero; without biases the bias slots stay in the buffer as zero gaps; state_dict keeps the reference's shapes and
    load_state_dict writes through the views; split_grads cuts the same blocks out of a gradient in the buffer's layout."""
    from nerfca_amd import _capi
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd import synthetic
    sdef, tdef = synthetic.net_definitions("cpu", F=48, early=2)
    torch.manual_seed(0)
    for cls, d, bias in ((CPPN, sdef, True), (Temporal, tdef, True), (CPPN, sdef, False), (Temporal, tdef, False)):
        d = dict(d, use_bias=bias)
        m = cls(d)
        b = m._binding
        assert b.net.F == 64 and m.num_filters == 48
        assert b.flat.numel() == _capi.check(_capi.lib().nca_param_count(C.byref(b.net)))
        assert b._is_flat()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert shapes["early_pts_layers.2.weight"] == (48, 48) and shapes["output_linear.0.weight"] == (1, 48)
        assert ("early_pts_layers.0.bias" in shapes) == bias
        assert bool(b.gaps) == (not bias)
        # everything outside the parameter views is zero, and the views cover exactly the parameters
        mask = torch.ones_like(b.flat, dtype=torch.bool)
        for g in b.split_grads(mask):
            g.fill_(False)
        assert int((~mask).sum()) == sum(p.numel() for p in m.parameters()) and float(b.flat[mask].abs().max()) == 0.0
        # writes through load_state_dict land in the flat buffer
        sd = {k: torch.full_like(v, 0.5) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        assert b._is_flat() and float(b.flat[~mask].min()) == 0.5 and float(b.flat[mask].abs().max()) == 0.0
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
        assert float(b.flat[~mask].min()) == 1.5 and float(b.flat[mask].abs().max()) == 0.0
    # beyond 128 units (and other channel counts than 3 -> 1) a net is bound to the general kernels, at the next multiple of 16
    for F, cin, cout, width in ((256, 3, 1, 256), (200, 3, 1, 208), (48, 2, 3, 48), (40, 3, 2, 48)):
        m = CPPN(dict(sdef, num_filters=F, num_input_channels=cin, num_output_channels=cout))
        b = m._binding
        assert _capi.net_is_general(b.net) and b.net.F == width and b.net.reserved == _capi.net_channels(cin, cout)
        assert b.flat.numel() == _capi.check(_capi.lib().nca_param_count(C.byref(b.net))) and b._is_flat()
        assert tuple(m.state_dict()["output_linear.0.weight"].shape) == (cout, F) and tuple(m.state_dict()["output_linear.0.bias"].shape) == (cout,)
        assert _capi.check(_capi.lib().nca_packed_bytes(C.byref(b.net), _capi.PREC_F32)) >= 4 * width * width
    with pytest.raises(_capi.NcaError, match="1024"):
        CPPN(dict(sdef, num_filters=1040))
    with pytest.raises(_capi.NcaError, match="Temporal net takes 3 input channels"):
        Temporal(dict(tdef, num_input_channels=2))


def test_planner_host_arithmetic_through_the_c_abi():
    """What the planner decides before any launch is host arithmetic and needs no GPU: the size of the forward store
    (nca_render_store_bytes: 8-bit staged in bf16, f32 blocks in f32), the per-call options (NcaRays.plan_opts) against the
    process-wide ones, and the refusal of values outside an option's range -- through the C ABI, with pointers that are never read."""
    import ctypes as C
    from nerfca_amd import _capi
    L = _capi.lib()
    dummy = C.c_void_p(0x1000)            # non-NULL is all these entry points check of the ray pointers

    def rays(R, S, single=0, opts=None):
        r = _capi.NcaRays(R=R, S=S, ray_is_f64=1, origins=dummy, dirs=dummy, phase=dummy, phase_stride_r=0, phase_stride_s=0, z=dummy, z_stride_r=0,
                          dists=dummy, I0=dummy, act=_capi.ACT_SOFTPLUS, single_field=single, scale=1e-2, store_format=0, plan_opts=None, plan_out=None)
        if opts is not None:
            r.plan_opts = C.cast(C.pointer(opts), C.c_void_p)
        return r

    net_s = _capi.NcaNet(F=128, n_hidden=4, n_late=0, enc_mode=_capi.ENC_BANDS, L=12, T=0, P=0)
    net_d = _capi.NcaNet(F=128, n_hidden=4, n_late=0, enc_mode=_capi.ENC_BANDS, L=12, T=8, P=10)

    def store(R, S, prec, opts=None, single=0):
        r = rays(R, S, single, opts)
        return L.nca_render_store_bytes(C.byref(r), C.byref(net_s), None if single else C.byref(net_d), prec)

    # the bench batch: 65 536 rays x 192 samples = 196 608 wave tiles of 64 samples.  Per 32-sample tile and net: a 4 KiB e4m3 input
    # block + four 4 KiB hidden blocks; per wave tile and net: five 1 KiB mask fragments and 64 raw outputs
    tiles = 65536 * 3
    b16 = store(65536, 192, _capi.PREC_BF16)
    per_tile = 2 * (2 * (4096 + 4 * 4096)) + 2 * 5 * 1024 + 2 * 64 * 4
    assert per_tile * tiles <= b16 <= per_tile * (tiles + 8) + 4096, (b16, per_tile * tiles)          # (slack tile slots up to the 8 waves, alignment)
    f32 = store(65536, 192, _capi.PREC_F32)
    assert 3.0 < f32 / b16 < 4.5, (f32, b16)                        # 4-byte blocks of every layer input against 1-byte ones
    assert store(1, 1, _capi.PREC_BF16) > 0 and store(2048, 192, _capi.PREC_BF16) < store(4096, 192, _capi.PREC_BF16)
    assert store(64, 64, _capi.PREC_BF16, single=1) < store(64, 64, _capi.PREC_BF16)

    # per-call options win over the process-wide value and do not touch it
    before = _capi.get_option(_capi.OPT_STAGE_FP8)
    # stage_fp8 = 0 = "nothing in 8 bits": the BF16 store (round 5, NCA_STORE_BF16) -- per 32-sample tile and net a 7 KiB bf16 input
    # block (one shared here? no: sized for one per net) + four 8 KiB hidden blocks; masks and raw outputs as before
    bf = store(65536, 192, _capi.PREC_BF16, _capi.NcaPlanOpts(stage_fp8=0))
    per_tile_bf = 2 * (2 * (7168 + 4 * 8192)) + 2 * 5 * 1024 + 2 * 64 * 4
    assert per_tile_bf * tiles <= bf <= per_tile_bf * (tiles + 8) + 4096, (bf, per_tile_bf * tiles)
    assert store(65536, 192, _capi.PREC_BF16, _capi.NcaPlanOpts(stage_fp8=0, bf16_store=0)) == 0              # "no forward store": the backward will recompute
    assert store(65536, 192, _capi.PREC_BF16, _capi.NcaPlanOpts(bf16_store=0)) == b16                            # (the 8-bit store is not that option's business)
    assert store(65536, 192, _capi.PREC_F32, _capi.NcaPlanOpts(stage_fp8=0, bf16_store=0)) == f32               # (bf16 options: the f32 store is not their business)
    assert store(1024, 500, _capi.PREC_BF16, _capi.NcaPlanOpts(stage_fp8=-1, stage_fp8_min_tiles=10 ** 9, bf16_store=0)) == 0
    assert store(1024, 500, _capi.PREC_BF16, _capi.NcaPlanOpts(stage_fp8=-1, stage_fp8_min_tiles=10 ** 9)) > store(1024, 500, _capi.PREC_BF16)   # below the threshold: the bf16 store
    assert store(1024, 500, _capi.PREC_BF16, _capi.NcaPlanOpts(stage_fp8=-1, stage_fp8_min_tiles=8000)) > 0
    assert _capi.get_option(_capi.OPT_STAGE_FP8) == before and store(65536, 192, _capi.PREC_BF16) == b16

    # values outside an option's range are refused, per call and process-wide, with a message; so are unknown options and empty batches
    for bad in (_capi.NcaPlanOpts(stage_fp8=2), _capi.NcaPlanOpts(wgrad_rebuild_weight_pct=99), _capi.NcaPlanOpts(stage_fp8_min_tiles=-1), _capi.NcaPlanOpts(resident_min_tiles=-2),
                _capi.NcaPlanOpts(bf16_store=2), _capi.NcaPlanOpts(overlap_cus=-1)):
        assert store(64, 64, _capi.PREC_BF16, bad) == -1 and b"takes" in L.nca_last_error()          # NCA_E_INVALID
    with pytest.raises(_capi.NcaError):
        _capi.set_option(_capi.OPT_WGRAD_REBUILD_WEIGHT_PCT, 250)
    with pytest.raises(_capi.NcaError):
        _capi.set_option(0, 1)                      # the retired option's slot
    with pytest.raises(_capi.NcaError):
        _capi.NcaPlanOpts(onchip_min_tiles=0)       # ... and its name
    assert store(0, 64, _capi.PREC_BF16) < 0 and b"empty" in L.nca_last_error()
    # a net on the general kernels (more than 128 units): a forward store when every net of the batch is theirs (0 beside a fused-kernel net: that backward
    # recomputes); its workspaces carry a chunk of activations, bounded by the caller's cap; a net no kernels have is an explicit error, not a size
    wide = _capi.NcaNet(F=256, n_hidden=4, n_late=0, enc_mode=_capi.ENC_BANDS, L=12, T=0, P=0)
    r = rays(64, 64, 1)
    # its forward store: X0 (80 padded columns), five layer outputs of 256 units, four layers' ReLU bit masks (2 KiB per 128 x 128 tile) and the raw field, f32
    rows = 64 * 64
    assert L.nca_render_store_bytes(C.byref(r), C.byref(wide), None, _capi.PREC_F32) == (rows * (80 + 5 * 256) + 4 * (rows // 128) * 2 * 512 + rows) * 4
    mixed = rays(64, 64, 0)
    assert L.nca_render_store_bytes(C.byref(mixed), C.byref(net_s), C.byref(_capi.NcaNet(F=256, n_hidden=4, n_late=0, enc_mode=_capi.ENC_BANDS, L=12, T=8, P=10)), _capi.PREC_F32) == 0
    full = L.nca_render_fwd_workspace_nets(C.byref(r), C.byref(wide), None, _capi.PREC_F32, 0)
    assert full == (64 * 64) * (80 + 2 * 256) * 4 and L.nca_render_fwd_workspace(C.byref(r)) < full
    assert L.nca_render_fwd_workspace_nets(C.byref(r), C.byref(wide), None, _capi.PREC_F32, full // 3) == (64 * 64 // 4) * (80 + 2 * 256) * 4
    assert L.nca_render_fwd_workspace_nets(C.byref(r), C.byref(wide), None, _capi.PREC_BF16, 0) == -2 and b"bf16 mode runs nets of up to 128" in L.nca_last_error()
    assert L.nca_render_bwd_workspace(C.byref(r), C.byref(wide), None, _capi.PREC_F32, 1 << 40) > full > L.nca_render_bwd_workspace(C.byref(r), C.byref(wide), None, _capi.PREC_F32, 1 << 20) > 0
    assert L.nca_mlp_fwd_workspace(C.byref(wide), _capi.PREC_F32, 1000, 0) == 1024 * (80 + 2 * 256) * 4
    narrow = _capi.NcaNet(F=64, n_hidden=4, n_late=0, enc_mode=_capi.ENC_BANDS, L=12, T=0, P=0)
    assert L.nca_mlp_fwd_workspace(C.byref(narrow), _capi.PREC_F32, 1000, 0) == 0
    odd = _capi.NcaNet(F=200, n_hidden=4, n_late=0, enc_mode=_capi.ENC_BANDS, L=12, T=0, P=0)
    assert L.nca_render_store_bytes(C.byref(r), C.byref(odd), None, _capi.PREC_F32) == -2 and L.nca_param_count(C.byref(odd)) == -2 and b"multiple of 16" in L.nca_last_error()


def test_plan_scope_is_a_per_thread_stack():
    """fused.PlanScope: the innermost scope of THIS thread is the current one; another thread sees none of them (a second trainer on
    its own thread neither inherits nor disturbs the first one's options), and leaving a scope restores the one around it."""
    import threading
    from nerfca_amd import _capi, fused
    assert fused.PlanScope.current() is None
    seen = {}
    with fused.PlanScope(stage_fp8=0) as outer:
        assert fused.PlanScope.current() is outer and outer.opts.stage_fp8 == 0 and outer.opts.resident_min_tiles == _capi.OPT_UNSET
        with fused.PlanScope(resident_min_tiles=-1) as inner:
            assert fused.PlanScope.current() is inner and inner.opts.stage_fp8 == _capi.OPT_UNSET

            def other():
                seen["none"] = fused.PlanScope.current()
                with fused.PlanScope(stage_fp8=1) as mine:
                    seen["mine"] = fused.PlanScope.current() is mine
                seen["after"] = fused.PlanScope.current()
            t = threading.Thread(target=other)
            t.start()
            t.join()
            assert fused.PlanScope.current() is inner
        assert fused.PlanScope.current() is outer
    assert fused.PlanScope.current() is None
    assert seen == {"none": None, "mine": True, "after": None}
    assert set(outer.decided()) >= {"fwd_store_format", "bwd_kernel_mode", "stage_fp8", "wave_tiles"} and all(v == 0 for v in outer.decided().values())


def test_psnr_control_cache_is_current():
    """tests/golden/psnr_f32_controls.json (the cached f32 control runs of tests/test_psnr_gates.py) was taken on THESE f32 kernel sources and
    holds every entry the gates ask for: an edit of the f32 / loss kernels that forgets to re-take the controls (tools/psnr_run.py on the GPU
    box, then tools/psnr_cache.py) fails here, in the CPU suite, instead of silently turning the GPU gates into live f32 runs."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("psnr_cache", os.path.join(ROOT, "tools", "psnr_cache.py"))
    pc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pc)
    d = json.load(open(pc.OUT))
    assert d["f32_sources_sha"] == pc.f32_sources_sha(), "re-take the f32 controls: the f32 / loss kernel sources changed"
    for sd in range(5):
        for v in ("f32", "f32_kick2e-3", "f32_kick4e-3", "f32_bf16init"):
            e = d["entries"][pc.key(65536, 192, 1000, v, sd)]
            assert 60.0 < e["psnr_mse_db"] < 90.0 and 50.0 < e["test_psnr_reference_def_db"] < 90.0, (v, sd, e)
        assert pc.key(1024, 500, 5000, "f32", sd) in d["entries"]


def test_bench_headline_line_is_compact_and_parseable():
    """bench.py's LAST stdout line is what the driver parses (round 5's 24.6 KB line came back `parsed: null`): built from a canned FULL record
    of a real run, it must stay under 4 000 bytes (the driver's tail is 8 KB), be one line of valid JSON, carry the bench contract's keys with
    `roofline` and `cpu_baseline`, and name every fraction by its denominator."""
    import glob
    import json
    import bench
    canned = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_bench_full.json")))[-1]
    full = json.loads(open(canned).read().strip().splitlines()[-1])
    text = bench.headline_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in text and len(text.encode()) < bench.HEADLINE_LIMIT_BYTES <= 4000, len(text)
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "full_record"):
        assert k in line, k
    assert line["vs_baseline"] is None and line["unit"] == "rays/s" and line["data"] == "synthetic"
    for k in ("workload", "rays_per_step_per_gpu", "samples_per_ray", "parallelism", "hip_graph"):
        assert k in line["config"], k
    assert "model" not in line["config"]
    roof = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "step"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"] and abs(line["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    # one key per denominator: an HBM-bound kernel reports hbm_frac, an MFMA-bound one mfma_frac; nothing is called roofline_frac
    for kern in roof["kernels"].values():
        assert ("hbm_frac" in kern) == (kern["bound"] == "hbm") and "mfma_frac" in kern
    assert "roofline_frac" not in json.dumps({k: v for k, v in line.items() if k != "precisions"})
    # a record bloated with every optional leg still fits: the optional parts are shed, the contract's keys never
    fat = dict(full)
    fat["config"] = dict(full["config"], workload=full["config"]["workload"] + " " + "x" * 1500)
    fat["precisions"] = {f"p{i}": {"rays_per_s": 1.0, "ms_per_step": 1.0, "step_mfma_frac": 0.1, "pad": "y" * 200} for i in range(8)}
    t2 = bench.headline_line(fat, None)
    assert len(t2.encode()) < bench.HEADLINE_LIMIT_BYTES and "roofline" in json.loads(t2) and "cpu_baseline" in json.loads(t2)


def test_philox_restatement_known_answers_and_sampler_distribution():
    """tests/philox_ref.py (the checker of the device-side batch sampler, csrc/nca_rng.hpp) against Random123's published known-answer vectors
    for Philox4x32-10, and the sampler's distribution: the keyed Feistel map is a bijection of [0, n) for any n, a batch holds EXACTLY n_var
    variance-ray ids in an arrangement whose per-slot frequency is 1/2 at var_sample_perc = 50 (run_composite.py:250-260: two draws with
    replacement, concatenated and shuffled), a slice of slots equals the same slots of the whole draw."""
    import numpy as np
    import philox_ref as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = P.philox4x32_10(*[[c] for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    for n in (1, 2, 3, 37, 1000, 1024, 65536, 70001):
        assert sorted(P.perm(np.arange(n), n, P.perm_keys(12345, 77)).tolist()) == list(range(n)), n
    var, non = np.arange(100, 800), np.arange(800, 5000)
    freq = np.zeros(1024)
    for it in range(200):
        ids = P.ray_ids(3, it, 1024, 512, var, non, 5000)
        is_var = ids < 800
        assert int(is_var.sum()) == 512 and ids.min() >= 100 and ids.max() < 5000
        freq += is_var
    assert abs(freq.mean() / 200 - 0.5) < 1e-12 and freq.min() / 200 > 0.33 and freq.max() / 200 < 0.67       # (sd of a slot's frequency: 0.035)
    whole = P.ray_ids(9, 4, 1024, 128, var, non, 5000)
    assert np.array_equal(P.ray_ids(9, 4, 1024, 128, var, non, 5000, slot0=512, count=256), whole[512:768])
    assert not np.array_equal(whole, P.ray_ids(9, 5, 1024, 128, var, non, 5000))
    u = P.ray_ids(9, 4, 4096, 0, None, None, 5000)                  # var_sample_perc == 0: uniform over the table (run_composite.py:260)
    assert u.min() >= 0 and u.max() < 5000 and abs(u.mean() - 2500) < 150
    t = P.uniform(9, 4, 4096)
    assert t.dtype == np.float32 and t.min() >= 0.0 and t.max() < 1.0 and abs(t.mean() - 0.5) < 0.03


def test_rng_header_compiled_for_the_host_equals_the_numpy_restatement(tmp_path):
    """csrc/nca_rng.hpp is host-compilable: the SAME header the sampler kernels include, built with g++, against tests/philox_ref.py -- Philox
    words, the keyed bijection, the slot -> ray-id rule -- so that the CPU suite pins the restatement the GPU tests compare the kernels with."""
    import ctypes as C
    import subprocess
    import numpy as np
    import philox_ref as P
    src = tmp_path / "rng.cpp"
    src.write_text('''#include "nca_rng.hpp"
extern "C" void words(unsigned long long seed, long long it, int stream, unsigned long long idx, unsigned* o) { NcaU4 r = nca_rng_words(seed, it, stream, idx); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
extern "C" unsigned long long perm1(unsigned long long i, unsigned long long n, unsigned long long seed, long long it) { return nca_perm(i, n, nca_perm_half_bits(n), nca_perm_keys(seed, it)); }
extern "C" unsigned long long below(unsigned lo, unsigned hi, unsigned long long n) { return nca_rng_below(lo, hi, n); }
extern "C" float unit(unsigned w) { return nca_rng_unit(w); }
''')
    so = tmp_path / "rng.so"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "nerf-ca_amd", "csrc"), str(src), "-o", str(so)], check=True)
    L = C.CDLL(str(so))
    L.perm1.restype = L.below.restype = C.c_ulonglong
    L.perm1.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.c_longlong]
    L.below.argtypes = [C.c_uint, C.c_uint, C.c_ulonglong]
    L.words.argtypes = [C.c_ulonglong, C.c_longlong, C.c_int, C.c_ulonglong, C.POINTER(C.c_uint)]
    L.unit.restype, L.unit.argtypes = C.c_float, [C.c_uint]
    seed, it = (1 << 40) + 12345, (1 << 33) + 77
    idx = np.array([0, 1, 5, 1 << 20, (1 << 35) + 3], dtype=np.uint64)
    for stream in (P.STREAM_IDS, P.STREAM_PERM, P.STREAM_JITTER):
        ref = P.rng_words(seed, it, stream, idx)
        for k, i in enumerate(idx.tolist()):
            out = (C.c_uint * 4)()
            L.words(seed, it, stream, i, out)
            assert [int(r[k]) for r in ref] == list(out), (stream, i)
            assert np.float32(L.unit(out[0])) == P.unit(np.array([out[0]]))[0]
            for n in (7, 5000, (1 << 40) + 1):
                assert L.below(out[0], out[1], n) == int(P.below([out[0]], [out[1]], n)[0])
    for n in (1, 2, 37, 1000, 65536):
        ref = P.perm(np.arange(n), n, P.perm_keys(seed, it))
        got = [L.perm1(i, n, seed, it) for i in range(min(n, 2000))]
        assert ref[: len(got)].tolist() == got, n
