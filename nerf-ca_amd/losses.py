"""Pixel loss and D2NeRF-style separation regularisers (train/model_helpers.py:189-262, 284-288).

The reference's PART functions (compute_ratio, compute_blendw_loss, compute_sigma_s_ray_loss, compute_occl_loss) as torch
operations on any device -- its API exports them by name (``from model_helpers import *``) and run_nerf.py's static loop calls
``compute_occl_loss`` on its own.  The two functions the composite script calls every step, ``compute_losses`` and
``weighted_MSELoss``, are HIP-backed in train/model_helpers.py (fused.loss_terms / fused.weighted_sq_err over nca_loss_fwd_bwd);
their torch restatements from the parts live with the CPU tests of the data-parallel bookkeeping (tests/injected_trainer.py:
``all_terms`` / ``weighted_sq_err``) -- nothing in this package assembles the 11-tuple in torch.
"""
from __future__ import annotations

import torch

_EPS = 1e-10
_CLIP = 1e-19


def blend_weight(sigma_s, sigma_d):
    """(blendw, max sigma_s, max sigma_d)  -- compute_ratio, model_helpers.py:189-198."""
    with torch.no_grad():
        top_s, top_d = sigma_s.max(), sigma_d.max()
    return sigma_d / (sigma_s + sigma_d + _EPS), top_s, top_d


def binary_entropy_of_blend(blendw, clip_threshold=_CLIP, skewness=1):
    """compute_blendw_loss, model_helpers.py:200-204."""
    b = torch.clip(blendw ** skewness, min=clip_threshold, max=1 - clip_threshold)
    nb = torch.clip(1 - b, min=clip_threshold)
    h = -(b * b.log() + nb * nb.log())
    return h.mean(dim=-1).mean()


def ray_entropy(sigma, dists, mask_threshold=0.1, clip_threshold=_CLIP, use_weighting=False, weighted_pixs=(), weighted_thresh=0.25):
    """compute_sigma_s_ray_loss, model_helpers.py:206-224 -> (masked mean entropy, mean ray sum)."""
    mass = sigma * dists
    total = mass.sum(dim=-1, keepdim=True)
    keep = (total >= mask_threshold).flatten().int()
    if len(weighted_pixs) > 0 and use_weighting:
        strong = torch.zeros_like(keep)
        strong[: weighted_pixs.shape[0]] = (weighted_pixs > 1 + weighted_thresh).int()
        keep = keep | strong
    prob = mass / total.clamp(min=clip_threshold)
    ent = keep * -(prob * (prob + _EPS).log()).sum(dim=-1)
    return ent.mean(), total.mean()


def occlusion(sigma, dists, reg_perc=0.1, use_back=False):
    """compute_occl_loss, model_helpers.py:226-248.  With use_back=False the (all-ones) back mask is
    OR-ed in, so every sample counts and this is the mean ray sum."""
    run = torch.cumsum(dists, dim=0)
    front = run < reg_perc * run[-1]
    back = run > (1 - reg_perc) * run[-1] if use_back else torch.ones_like(front)
    m = (front | back).to(sigma.dtype if sigma.dtype == dists.dtype else dists.dtype)
    return (sigma * dists * m).sum(dim=-1).mean()
