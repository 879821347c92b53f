"""Data-loader side of the hot path -- drop-in for the reference's ``train/data_helpers.py``
functions that feed it (lines 129-171): the per-ray table and the depth vector.

The config-file parser and the wandb helpers of the reference are out of scope;
``load_config`` reads the same ``key = value`` files into a namespace.
"""
import ast
from types import SimpleNamespace

import numpy as np
import torch

from .proj_helpers import get_ray_values_tigre


def load_config(path: str, **overrides) -> SimpleNamespace:
    """Read a reference config file (train/composite.txt, train/3d.txt): ``key = value`` per line."""
    out = {}
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if "=" not in line:
                continue
            k, v = (t.strip() for t in line.split("=", 1))
            try:
                out[k] = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                out[k] = v
    out.update(overrides)
    return SimpleNamespace(**out)


def denormalize_image(image, img_width, img_height, img_min_max):
    """data_helpers.py:129-139: images are stored transposed; undo the min-max scaling when the
    stored image spans exactly [0, 1]."""
    image = image.reshape((img_width, img_height)).T
    if int(np.min(image)) == 0 and int(np.max(image)) == 1:
        return image * (img_min_max[1] - img_min_max[0]) + img_min_max[0]
    return image


def prepare_data_for_loader_tigre(data, geo_info, img_width, img_height, depth_samples_per_ray, weighted_loss_max, device, use_weighting=True):
    """data_helpers.py:141-165 -> (rays f64[N_img*W*H, 4, 3], phases i64[N]).

    Row layout per ray: origin, direction, pixel value x3, loss weight x3.  Ray id = img*W*H + w*H + h.
    """
    geom = np.stack([np.stack(get_ray_values_tigre(f["theta"], f["phi"], f["larm"], geo_info, device), 0) for f in data], 0)
    pix = np.stack([denormalize_image(np.load(f["file_path"]), img_width, img_height, f["img_min_max"]) for f in data], 0)
    if use_weighting:
        wgt = np.stack([np.load(f["weighted_file_path"]).reshape((img_width, img_height)).T for f in data], 0)
    else:
        wgt = np.ones((pix.shape[0], img_width, img_height))
    wgt = (wgt - 1) * weighted_loss_max + 1          # [1,2] -> [1, 1 + weighted_loss_max]
    return assemble_ray_table(geom, pix, wgt, np.array([f["heart_phase"] for f in data]))


def assemble_ray_table(geom, pix, wgt, heart_phases):
    """geom [N,2,W,H,3], pix/wgt [N,W,H], heart_phases [N] -> the reference's ray table and phase vector."""
    n_img, _, W, H, _ = geom.shape
    table = np.empty((n_img, W, H, 4, 3), dtype=np.float64)
    table[:, :, :, 0, :] = geom[:, 0]
    table[:, :, :, 1, :] = geom[:, 1]
    table[:, :, :, 2, :] = pix[..., None]
    table[:, :, :, 3, :] = wgt[..., None]
    phases = np.broadcast_to(np.asarray(heart_phases)[:, None, None], (n_img, W, H)).reshape(-1).copy()
    return table.reshape(-1, 4, 3), phases


class TrainingData(SimpleNamespace):
    """What ``CompositeTrainer`` / ``StaticTrainer`` take as ``data`` (the tensors run_composite.py:84-125 builds before its loop)."""


def load_training_data(general_file: str, train_file: str, test_file: str, run_args, device) -> TrainingData:
    """The loop glue of train/run_composite.py:65-125 from the reference's on-disk schema: ``general.json`` (TIGRE geometry, detector
    size, near / far thresholds, ``max_pixel_value``), ``train-*.json`` / ``test-*.json`` (``frames``: angles, ``file_path`` and
    ``weighted_file_path`` of the ``.npy`` images, ``img_min_max``, ``heart_phase``, ``image_id_str``) -> the ray table and phase
    vector, the high-variance ray ids of the importance sampler (:97-99) and the tensors of the ONE held-out view the loop
    evaluates (:68-70, :110-125).  ``run_args`` supplies ``depth_samples_per_ray_coarse``, ``weighted_loss_max``, ``var_sample_thre``."""
    import json
    with open(general_file) as fh:
        info = json.load(fh)
    with open(train_file) as fh:
        train = json.load(fh)["frames"]
    with open(test_file) as fh:
        test = json.load(fh)["frames"][:1]                 # "always only use one test image" (:72-74)
    W, H = info["nDetector"]
    rays, phases = prepare_data_for_loader_tigre(train, info, W, H, run_args.depth_samples_per_ray_coarse, run_args.weighted_loss_max, device)
    var_ids = np.argwhere(rays[:, -1, 0] > 1.0 + run_args.var_sample_thre / 100.0).flatten()
    non_var = np.setxor1d(var_ids, np.arange(rays.shape[0]))
    out = TrainingData(geo=info, rays_train=torch.from_numpy(rays).to(device), phases_train=torch.from_numpy(phases).to(device), n_images=len(train),
                       var_ray_ids=var_ids, non_var_ray_ids=non_var, train_img_indices=[f.get("image_id_str") for f in train],
                       test_img_indices=[f.get("image_id_str") for f in test])
    if test:
        f = test[0]
        to, td = get_ray_values_tigre(f["theta"], f["phi"], f["larm"], info, device)
        img = denormalize_image(np.load(f["file_path"]), W, H, f["img_min_max"])
        out.test_origins = torch.as_tensor(np.asarray(to), dtype=torch.float32).reshape(-1, 3).to(device)       # torch.Tensor(...) of the reference: f32
        out.test_directions = torch.as_tensor(np.asarray(td), dtype=torch.float32).reshape(-1, 3).to(device)
        out.test_image = torch.as_tensor(np.asarray(img), dtype=torch.float32).reshape(-1).to(device)
        out.test_phase = int(f["heart_phase"])
    return out


def create_depth_values(near_thresh, far_thresh, depth_samples_per_ray_coarse, device):
    """data_helpers.py:167-171."""
    t = torch.linspace(0.0, 1.0, depth_samples_per_ray_coarse)
    return (near_thresh * (1.0 - t) + far_thresh * t).to(device)
