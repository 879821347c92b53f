"""Synthetic angiography data with the reference's on-disk semantics (no XCAT/MAGIX data exists
here): XCAT cone-beam geometry, the 4 preset training views x 10 cardiac phases, one held-out test
view; target images are rendered from a fixed random "teacher" network pair so that PSNR is
meaningful; per-view temporal-variance weights in [1,2] as preprocess/general_helpers.py:17-44.

Everything ends up in the reference's ray table format (train/data_helpers.py:141-165):
rays f64[N_img*W*H, 4, 3] rows (origin, direction, pixel x3, weight x3), phases i64[N].
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np
import torch

from .train.data_helpers import assemble_ray_table, create_depth_values
from .train.proj_helpers import get_ray_values_tigre

TRAIN_VIEWS = [[-30, 30], [-30, -30], [60, -30], [60, 30]]   # preprocess/general_helpers.py:131-136
TRAIN_VIEWS_8 = TRAIN_VIEWS + [[-60, 0], [0, 0], [30, 30], [90, 0]]
TEST_VIEW = [-5, 40]                                          # preprocess/general_helpers.py:94 (first only)
MAX_PIXEL_VALUE = math.log(8.670397)                          # preprocess/tigre_helpers.py:68
NEAR, FAR = 3.4259, 5.5741                                    # get_near_far for DSO=4.5


def xcat_geometry(n_det: int) -> dict:
    """XCAT cone beam in the loader's scaled units; detector n x n covering the same field of view."""
    return dict(DSD=25.0, DSO=4.5, nDetector=[n_det, n_det], dDetector=[2.0 / n_det, 2.0 / n_det], offDetector=[0.0, 0.0],
                near_thresh=NEAR, far_thresh=FAR, max_pixel_value=MAX_PIXEL_VALUE)


def magix_geometry(n_det: int, vol_voxels: int = 200) -> dict:
    """The reference's MAGIX / CCTA cone beam (preprocess/tigre_helpers.py:174-206: DSD 2000 mm, DSO 600 mm, a 200 mm detector,
    0.9 mm voxels) in the loader's units (x 1e-2, store_general_geo :66-80), detector n x n over the same field of view; near /
    far as get_near_far (:44-56) gives them for a centred cube of ``vol_voxels`` voxels (the reference takes the volume's size
    from its data file; 200 is the 200-pixel preset's counterpart)."""
    dso, half = 6.0, 0.5 * vol_voxels * 0.9e-2
    reach = math.hypot(half, half)
    return dict(DSD=20.0, DSO=dso, nDetector=[n_det, n_det], dDetector=[2.0 / n_det, 2.0 / n_det], offDetector=[0.0, 0.0],
                near_thresh=max(0.0, dso - reach), far_thresh=min(2 * dso, dso + reach), max_pixel_value=MAX_PIXEL_VALUE)


GEOMETRIES = {"xcat": xcat_geometry, "magix": magix_geometry}


def net_definitions(device, F=128, early=4, L=12, T=8, pos_enc="free_windowed", window_start=1):
    static = dict(num_early_layers=early, num_late_layers=0, num_filters=F, num_input_channels=3, num_output_channels=1,
                  use_bias=True, pos_enc=pos_enc, pos_enc_window_start=window_start, pos_enc_basis=L, fourier_sigma=0,
                  fourier_gaussian=None, act_func="relu", device=device)
    temporal = dict(static, num_input_times=1, use_time_latents=True, num_time_dim=T)
    return static, temporal


@dataclass
class SyntheticData:
    geo: dict
    rays_train: torch.Tensor       # f64 [N,4,3] on `device`
    phases_train: torch.Tensor     # i64 [N]
    n_images: int
    test_origins: torch.Tensor     # f32 [W*H,3]
    test_directions: torch.Tensor
    test_image: torch.Tensor       # f32 [W*H] log-intensity of the held-out view
    test_phase: int
    var_ray_ids: np.ndarray
    non_var_ray_ids: np.ndarray


def make_dataset(n_det: int, S: int, device, views=None, n_phases: int = 10, teacher_seed: int = 0, F: int = 128,
                 var_sample_thre: float = 3.0, weighted_loss_max: float = 1.0,
                 render: Optional[Callable] = None, teacher=None, chunk_rays: int = 65536, geometry: str = "xcat") -> SyntheticData:
    """Build the ray table.  ``render(static, temporal, origins, dirs, phase_ids, I0, z, dists) -> pix`` is
    the fused HIP forward by default; ``teacher`` = (static_model, temporal_model) overrides the
    seed-derived teacher pair."""
    from .model.CPPN import CPPN
    from .model.Temporal import Temporal
    from .train import model_helpers as MH
    views = TRAIN_VIEWS if views is None else views
    geo = GEOMETRIES[geometry](n_det)
    W = H = n_det
    if teacher is None:
        torch.manual_seed(teacher_seed)
        sdef, tdef = net_definitions(device, F=F, pos_enc="vanilla")
        ts, tt = CPPN(sdef).to(device), Temporal(tdef).to(device)
        with torch.no_grad():
            # a default-init net renders an almost constant image; give the phantom contrast in space
            # (output gains) and over the cardiac phase (latent gain) so that PSNR means something
            tt.time_latents.mul_(2.0)
            for m, gain, bias in ((ts, 40.0, -1.0), (tt, 60.0, -2.0)):
                m.output_linear[0].weight.mul_(gain)
                m.output_linear[0].bias.fill_(bias)
    else:
        ts, tt = teacher
    z = create_depth_values(geo["near_thresh"], geo["far_thresh"], S, device)
    if render is None:
        from .fused import render_rays

        def render(s, t, o, d, ph, I0, zz, dd):
            return render_rays(s, t, o, d, ph, I0, zz, dd)[0]

    def render_view(o_np, d_np, phase):
        o = torch.from_numpy(o_np.reshape(-1, 3)).to(device)
        d = torch.from_numpy(d_np.reshape(-1, 3)).to(device)
        out = []
        dists = MH._interval_lengths(z, d)
        with torch.no_grad():
            for i in range(0, o.shape[0], chunk_rays):
                oo, dd = o[i:i + chunk_rays], d[i:i + chunk_rays]
                ph = torch.full((oo.shape[0],), phase, dtype=torch.int32, device=device)
                I0 = torch.full((oo.shape[0],), MAX_PIXEL_VALUE, dtype=torch.float32, device=device)
                out.append(render(ts, tt, oo, dd, ph, I0, z, dists).float())
        return torch.cat(out).reshape(W, H).cpu().numpy()

    geom, pix, wgt, phases = [], [], [], []
    for (theta, phi) in views:
        ro, rd = get_ray_values_tigre(theta, phi, 0, geo, "cpu")
        imgs = [render_view(ro, rd, p) for p in range(n_phases)]                      # log-intensity [W,H]
        absorb = np.stack([MAX_PIXEL_VALUE - im for im in imgs], 0).reshape(n_phases, -1)
        var = np.var(np.exp(-absorb), axis=0).reshape(W, H) if n_phases > 1 else np.zeros((W, H))
        var = (var - var.min()) / (var.max() - var.min() + 1e-10) + 1.0                  # [1,2]
        for p in range(n_phases):
            geom.append(np.stack([ro, rd], 0))
            pix.append(imgs[p])
            wgt.append((var - 1) * weighted_loss_max + 1)
            phases.append(p)
    table, ph = assemble_ray_table(np.stack(geom, 0), np.stack(pix, 0), np.stack(wgt, 0), np.array(phases))
    var_ids = np.argwhere(table[:, -1, 0] > 1.0 + var_sample_thre / 100.0).flatten()
    non_var = np.setxor1d(var_ids, np.arange(table.shape[0]))
    to, td = get_ray_values_tigre(TEST_VIEW[0], TEST_VIEW[1], 0, geo, "cpu")
    test_phase = 3
    test_img = render_view(to, td, test_phase)
    return SyntheticData(geo=geo, rays_train=torch.from_numpy(table).to(device), phases_train=torch.from_numpy(ph).to(device),
                         n_images=len(phases), test_origins=torch.from_numpy(to.reshape(-1, 3)).to(device),
                         test_directions=torch.from_numpy(td.reshape(-1, 3)).to(device),
                         test_image=torch.from_numpy(test_img.reshape(-1)).to(device), test_phase=test_phase,
                         var_ray_ids=var_ids, non_var_ray_ids=non_var)
