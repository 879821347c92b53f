"""Checkpoint loading and 4-D field export around the fused point kernel.

The reference only *saves* checkpoints (``CPPN.save`` / ``Temporal.save``, model/CPPN.py:164-180) and leaves
sampling the trained fields on a grid to downstream scripts; these helpers close that loop for users of the
drop-in modules.  Field evaluation goes through ``fused.eval_points`` (the HIP kernel), in chunks.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import fused as _fused
from .fused import eval_points


def load_checkpoint(filename, device=None):
    """Rebuild the model a ``save()`` call wrote: returns ``(model, training_information)``.

    The blob holds the constructor dictionary under ``"parameters"`` (with the device it was trained on), the
    state dict under ``"model"`` and, for windowed encodings, ``windowed_alpha`` / ``freq_mask_alpha``.  A blob with a
    ``time_latents`` entry (or ``num_time_dim`` in its parameters) is a ``Temporal``, anything else a ``CPPN``."""
    from .model.CPPN import CPPN
    from .model.Temporal import Temporal
    blob = torch.load(filename, map_location="cpu", weights_only=False)
    params = dict(blob["parameters"])
    if device is not None:
        params["device"] = device
    if params.get("fourier_gaussian") is not None:
        params["fourier_gaussian"] = params["fourier_gaussian"].to("cpu")
    is_temporal = "time_latents" in blob["model"] or "num_time_dim" in params
    model = (Temporal if is_temporal else CPPN)(params)
    model.load_state_dict(blob["model"])
    if "windowed_alpha" in blob:
        model.windowed_alpha = blob["windowed_alpha"]
    if "freq_mask_alpha" in blob:
        model.freq_mask_alpha = blob["freq_mask_alpha"]
    if device is not None:
        model = model.to(device)
    return model, blob.get("training_information", {})


@torch.no_grad()
def density_volume(static_model, temp_model, phase: Optional[int], resolution: Sequence[int] = (128, 128, 128),
                   bounds: Tuple[Tuple[float, float], ...] = ((-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0)), output_activation: str = "softplus",
                   scale_value: float = 1e-2, chunk_points: int = 1 << 22):
    """Sample the trained fields on a regular grid: returns ``(sigma_static, sigma_dynamic | None)`` as f32
    ``[nx, ny, nz]`` tensors on the models' device, ``sigma = act(raw) * scale_value`` as in
    render_volume_density_composite (model_helpers.py:72-84).  ``phase`` selects the heart phase of the dynamic field
    (one 3-D volume per phase = the 4-D reconstruction); ``temp_model=None`` exports the static field only."""
    dev = next(static_model.parameters()).device
    axes = [torch.linspace(lo, hi, n, device=dev) for (lo, hi), n in zip(bounds, resolution)]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3)
    out_s = torch.empty(grid.shape[0], dtype=torch.float32, device=dev)
    out_d = torch.empty_like(out_s) if temp_model is not None else None
    line = 256                                                   # grid points are handed to the compositing kernel as rows of 256
    i0 = torch.zeros(1, dtype=torch.float32, device=dev)
    dz = torch.zeros(line, dtype=torch.float64, device=dev)      # only the activation is wanted here, not the line integral
    for i in range(0, grid.shape[0], chunk_points):
        pts = grid[i:i + chunk_points]
        n = pts.shape[0]
        pad = (-n) % line
        raw_s = torch.nn.functional.pad(eval_points(static_model, pts)[:, 0], (0, pad)).reshape(-1, line)
        if temp_model is not None:
            ph = torch.full((n,), int(phase), dtype=torch.int32, device=dev)
            raw_d = torch.nn.functional.pad(eval_points(temp_model, pts, ph)[:, 0], (0, pad)).reshape(-1, line)
            _, ss, sd = _fused.composite_raw(raw_s, raw_d, i0, dz, output_activation, False, scale_value, False)
            out_d[i:i + n] = sd.reshape(-1)[:n]
        else:                                                    # one field: the kernel's single mode returns sigma un-scaled,
            _, ss, _ = _fused.composite_raw(raw_s, raw_s, i0, dz, output_activation, False, scale_value, False)   # so use the scaled pair mode
        out_s[i:i + n] = ss.reshape(-1)[:n]
    shape = tuple(int(n) for n in resolution)
    return out_s.reshape(shape), (out_d.reshape(shape) if out_d is not None else None)
