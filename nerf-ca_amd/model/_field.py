"""Shared implementation of the two coordinate networks.

The classes own ``nn.Parameter``s under the reference's state-dict keys
(``early_pts_layers.{0,2,..}.{weight,bias}``, ``skip_connection.0.*``, ``late_pts_layers.*``,
``output_linear.0.*``, ``time_latents``) so that ``torch.optim``, ``state_dict()`` and ``save()``
behave as in the reference, but they never run those ``nn.Linear`` modules: evaluation goes
through the fused HIP kernels (``fused.eval_points`` / ``fused.render_rays``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _capi
from ..fused import FieldBinding, eval_points
from ..schedules import freq_mask, nerfies_window

_WINDOWED = ("nerfies_windowed", "free_windowed")


def _dense(n_in: int, n_out: int, bias: bool, relu: bool) -> nn.Sequential:
    mods = [nn.Linear(n_in, n_out, bias=bias)]
    if relu:
        mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class FieldBase(nn.Module):
    """Positional encoding bookkeeping + parameter containers + binding to the HIP library."""

    def _setup_common(self, d: dict) -> None:
        self.version = "v0.00"
        self.model_definition = d
        self.device = d["device"]
        self.num_early_layers = d["num_early_layers"]
        self.num_late_layers = d["num_late_layers"]
        self.num_filters = d["num_filters"]
        self.num_input_channels = d["num_input_channels"]
        self.num_output_channels = d["num_output_channels"]
        self.use_bias = d["use_bias"]
        self.use_pos_enc = d["pos_enc"]
        self.first_act_func = nn.ReLU()
        self.act_func = nn.ReLU()
        self.store_activations = False
        self.activation_dictionary = {}
        if not (1 <= self.num_input_channels <= 8 and 1 <= self.num_output_channels <= 8):
            raise _capi.NcaError("the kernels take 1 .. 8 input channels and produce 1 .. 8 output channels")
        self.pos_enc_basis = 0
        enc_features = self.num_input_channels
        if self.use_pos_enc != "none":
            self.pos_enc_basis = d["pos_enc_basis"]
            self.pos_enc_window_start = d["pos_enc_window_start"]
            enc_features = self.num_input_channels * (1 + 2 * self.pos_enc_basis)
            if self.use_pos_enc == "fourier":
                enc_features = self.num_input_channels * 2 * self.pos_enc_basis
                self.fourier_sigma = d["fourier_sigma"]
                self.fourier_coefficients = (d["fourier_gaussian"] * self.fourier_sigma).to(self.device)
        self._enc_features = enc_features
        self._win_cache = None

    def _build_layers(self, n_in: int) -> None:
        """Same construction order as the reference so a given torch seed yields the same init."""
        F, b = self.num_filters, self.use_bias
        early = list(_dense(n_in, F, b, True))
        for _ in range(self.num_early_layers):
            early += list(_dense(F, F, b, True))
        self.early_pts_layers = nn.ModuleList(early)
        if self.num_late_layers > 0:
            self.skip_connection = _dense(F + n_in, F, b, True)
            late = []
            for _ in range(self.num_late_layers - 1):
                late += list(_dense(F, F, b, True))
            self.late_pts_layers = nn.ModuleList(late)
        self.output_linear = _dense(F, self.num_output_channels, b, False)

    def _bind(self, time_dim: int, phases: int) -> None:
        mode = _capi.ENC_NONE
        if self.use_pos_enc != "none" and self.pos_enc_basis > 0:
            mode = _capi.ENC_FOURIER if self.use_pos_enc == "fourier" else _capi.ENC_BANDS
        if self.num_filters < 1 or self.num_filters > 1024:
            raise _capi.NcaError(f"num_filters = {self.num_filters}: the kernels run nets of up to 1024 units per layer")
        chan = _capi.net_channels(self.num_input_channels, self.num_output_channels)
        if chan and time_dim > 0:
            raise _capi.NcaError("a Temporal net takes 3 input channels and produces 1 output channel")
        # The fused kernels exist for 32, 64 and 128 units (3 -> 1 channels); the general kernels (f32; wider nets, other channel counts:
        # CPPN.py:40-65 takes any) for every multiple of 16 up to 1024.  A net of another width runs as the next one up with zero-weight
        # units (FieldBinding).
        if self.num_filters > 128 or chan:
            width = (self.num_filters + 15) // 16 * 16
        else:
            width = 32 if self.num_filters <= 32 else (64 if self.num_filters <= 64 else 128)
        net = _capi.NcaNet(F=width, n_hidden=self.num_early_layers, n_late=self.num_late_layers, enc_mode=mode,
                           L=self.pos_enc_basis if mode != _capi.ENC_NONE else 0, T=time_dim, P=phases, reserved=chan)
        object.__setattr__(self, "_binding", FieldBinding(self, net))

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        b = getattr(self, "_binding", None)
        if b is not None:
            b.reflatten()
        self._win_cache = None
        return out

    # -- encoding state -----------------------------------------------------------------------
    def _param_device(self):
        return next(self.parameters()).device

    def _band_window(self) -> torch.Tensor:
        """f32[L] multiplier of each frequency band (ones when the encoding is not windowed)."""
        L = self.pos_enc_basis
        if self.use_pos_enc == "free_windowed":
            if not hasattr(self, "freq_mask_alpha"):
                raise AttributeError(f"'{type(self).__name__}' object has no attribute 'freq_mask_alpha' "
                                     "(call update_freq_mask_alpha before the first forward)")
            return self.freq_mask_alpha
        if self.use_pos_enc == "nerfies_windowed":
            return nerfies_window(L, self.windowed_alpha)
        return torch.ones(L, dtype=torch.float32)

    def _enc_buffers(self):
        """(window f32[L] | None, fourier f32[3L] | None) on the parameters' device, for the kernels."""
        dev = self._param_device()
        b = self._binding.net
        if b.enc_mode == _capi.ENC_BANDS:
            pinned = self._binding.static_window      # a persistent device vector owned by a graph-captured step
            if pinned is not None:
                return pinned, None
            w = self._band_window()
            key = (w.data_ptr(), w._version, dev)
            if self._win_cache is None or self._win_cache[0] != key:
                self._win_cache = (key, w.detach().to(device=dev, dtype=torch.float32).contiguous(), w)
            return self._win_cache[1], None
        if b.enc_mode == _capi.ENC_FOURIER:
            return None, self.fourier_coefficients.detach().to(device=dev, dtype=torch.float32).contiguous()
        return None, None

    def update_freq_mask_alpha(self, current_iter, max_iter):
        self.freq_mask_alpha, self.windowed_alpha = freq_mask(self.pos_enc_basis, current_iter, max_iter, self.pos_enc_window_start)

    def update_windowed_alpha(self, current_iter, max_iter):
        self.windowed_alpha = (self.pos_enc_basis * current_iter) / max_iter

    def _window_vector(self, pos_enc_basis):
        return nerfies_window(pos_enc_basis, self.windowed_alpha).to(self.device)

    def _encode(self, values: torch.Tensor, pos_enc_basis: int) -> torch.Tensor:
        """Materialised encoding (diagnostic API; the fused kernels never build this tensor)."""
        if pos_enc_basis <= 0:
            return values
        if self.use_pos_enc == "fourier":
            tiled = torch.cat(pos_enc_basis * [values], dim=-1)
            arg = 2 * np.pi * tiled * self.fourier_coefficients.to(values.device)
            return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)
        lead = values.shape[:-1]
        freqs = (2.0 ** torch.arange(0, pos_enc_basis)).to(values.device)
        scaled = values[..., None, :] * freqs[:, None]
        feat = torch.sin(torch.stack([scaled, scaled + 0.5 * torch.pi], dim=-2))
        if self.use_pos_enc in _WINDOWED:
            feat = self._band_window().to(values.device)[..., None, None] * feat
        return torch.cat([values, feat.reshape((*lead, -1))], dim=-1)

    def activations(self, store_activations: bool) -> None:
        self.store_activations = store_activations
        if not store_activations:
            self.activation_dictionary = {}

    def save(self, filename, training_information: dict) -> None:
        blob = {"version": self.version, "parameters": self.model_definition,
                "training_information": training_information, "model": self.state_dict()}
        if "nerfies_windowed" in self.use_pos_enc:
            blob["windowed_alpha"] = self.windowed_alpha
        if "free_windowed" in self.use_pos_enc:
            blob["freq_mask_alpha"] = self.freq_mask_alpha
        torch.save(blob, f=filename)

    def _points(self, x: torch.Tensor, phase=None) -> torch.Tensor:
        lead = x.shape[:-1]
        out = eval_points(self, x.reshape(-1, x.shape[-1]), phase)
        return out.reshape(*lead, self.num_output_channels)
