"""Dynamic attenuation field -- drop-in for the reference's ``model/Temporal.py``.

Input = cat[posenc(x), time_latents[phase]] (model/Temporal.py:113-151).  As in the reference only
``use_time_latents=True`` with ``num_late_layers == 0`` is a working configuration: the reference
raises UnboundLocalError otherwise (Temporal.py:128-135, 149), and so does this class.
"""
import torch
import torch.nn as nn

from .. import _capi
from ._field import FieldBase


class Temporal(FieldBase):
    def __init__(self, model_definition: dict) -> None:
        super().__init__()
        d = model_definition
        self._setup_common(d)
        self.num_input_times = d["num_input_times"]
        self.use_time_latents = d["use_time_latents"]
        self.num_time_dim = 0
        if self.use_time_latents:
            self.num_time_dim = d["num_time_dim"]
            self.fixed_frame_ids = torch.arange(0, 10)
            self.time_latents = nn.Parameter(torch.rand((self.fixed_frame_ids.shape[0], self.num_time_dim)))
        self.input_features_pts = self._enc_features
        self.input_features_time = self.num_input_times
        if self.use_pos_enc != "none":
            self.windowed_alpha = 0
        self.input_features = self.input_features_pts + (self.num_time_dim if self.use_time_latents else self.input_features_time)
        self._build_layers(self.input_features)
        if self.use_time_latents:
            self._bind(time_dim=self.num_time_dim, phases=int(self.fixed_frame_ids.shape[0]))

    def forward_composite(self, x: torch.Tensor, ts: torch.Tensor) -> torch.Tensor:
        """points f32[n,3], phase ids [n] -> f32[n,1]  (model/Temporal.py:138-151)."""
        if not self.use_time_latents:
            raise UnboundLocalError("local variable 'learned_time_pts' referenced before assignment "
                                    "(use_time_latents=False has no working path in the reference, Temporal.py:149)")
        if self.num_late_layers > 0:
            raise UnboundLocalError("local variable 'outputs' referenced before assignment "
                                    "(num_late_layers > 0 has no output in the reference, Temporal.py:128-135)")
        return self._points(x, ts)

    def query_time(self, xs: torch.Tensor, ts: torch.Tensor) -> torch.Tensor:
        """points f32[n,3], latent VECTORS f32[n, num_time_dim] -> f32[n,1] (model/Temporal.py:113-136: the inner call of
        forward_composite; on its own it evaluates the field at latents that are not rows of ``time_latents``, e.g. a cardiac
        phase interpolated between two frames).  The kernels gather latents from a table by id, so the distinct vectors
        become temporary tables of ``fixed_frame_ids`` rows each (one launch per table).  Gradients reach the network
        weights AND the passed vectors, per point as autograd gives them (nca_mlp_bwd's ``g_latents``: W0[:, latent columns]^T D_0 of
        every point) -- vectors that require a gradient (a learned or interpolated latent) get it."""
        if self.num_late_layers > 0:
            raise UnboundLocalError("local variable 'outputs' referenced before assignment "
                                    "(num_late_layers > 0 has no output in the reference, Temporal.py:128-135)")
        from ..fused import eval_points_with_latents
        return eval_points_with_latents(self, xs, ts)

    def pos_enc(self, values, pos_enc_basis):
        return self._encode(values, pos_enc_basis)

    def windowed_pos_enc(self, pos_enc_basis):
        return self._window_vector(pos_enc_basis)
