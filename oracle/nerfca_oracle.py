"""CPU oracle for the NeRF-CA hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, with plain torch CPU tensor ops, the arithmetic of the reference's
ray-sampling -> (static + dynamic) MLP -> log-space X-ray compositing path and the pieces
around it (losses, schedules, ray geometry).  It exists so that the HIP kernels can be
checked against something that is *not* the HIP kernels.

Who may import it: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``nerf-ca_amd/`` imports, calls or executes it; the
product path raises if the HIP library is missing instead of falling back here.

Parity status: PINNED.  ``tests/test_oracle_vs_golden.py`` checks every function below
against fixtures in ``tests/golden/*.npz`` that were produced by importing the real
reference (``/root/reference``) in the build container with
``tests/golden/make_golden.py`` (committed).  The reference itself has no tests and no
golden vectors of its own, so those fixtures are the pin.

Third-party arithmetic under the path: torch CPU kernels of the installed wheel
(torch 2.10.0+rocm7.0): Linear/ReLU/Softplus/sin/sum/cumsum/searchsorted/sort/Adam.

All ``file:line`` citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as TF

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# net description
# --------------------------------------------------------------------------------------


@dataclass
class NetSpec:
    """Shape of one coordinate MLP (model/CPPN.py:6-69, model/Temporal.py:6-93)."""

    num_filters: int = 128
    num_early_layers: int = 4
    num_late_layers: int = 0
    num_input_channels: int = 3
    num_output_channels: int = 1
    pos_enc: str = "free_windowed"  # none | fourier | nerfies_windowed | free_windowed | <other>=plain
    pos_enc_basis: int = 12
    pos_enc_window_start: int = 1
    fourier_coefficients: Optional[Tensor] = None  # gaussian * sigma, shape [C * L]
    num_time_dim: int = 0  # 0 -> static net (CPPN); >0 -> dynamic net (Temporal)
    num_phases: int = 10  # Temporal.py:25 fixed_frame_ids = arange(0, 10)
    emulate_bf16: bool = False  # NOT reference behaviour: round where the bf16 HIP path rounds (see _q)
    # NOT reference behaviour either: with emulate_bf16, also round what the bf16 HIP path's BACKWARD rounds when it runs from a
    # forward store with fp8 staging (nca_layout.hpp): value = samples per ray (tiles of 64 consecutive samples of a ray share
    # a power-of-two scale), 0 = off.  emulate_onchip_last: the last hidden layer's weight gradient is formed on chip from
    # bf16 operands (an option of bf16 staging, emulate_stage_formats = None), else from staged output gradients like the others.
    emulate_fp8_stage: int = 0
    emulate_onchip_last: bool = False
    emulate_stage_formats: Optional[Tuple[str, str]] = ("e5m2", "e4m3")  # (output gradients, layer inputs); ("bf16", "bf16"): the bf16 store
    #                                                                      (NCA_STORE_BF16: mode-5 arithmetic, nothing in 8 bits); None: round 3's
    #                                                                      bf16 staging (last layer recomputed: _StagedOut)

    @property
    def enc_features(self) -> int:
        c, L = self.num_input_channels, self.pos_enc_basis
        if self.pos_enc == "none":
            return c
        if self.pos_enc == "fourier":
            return c * 2 * L  # CPPN.py:37
        return c + c * 2 * L  # CPPN.py:34

    @property
    def in_features(self) -> int:
        return self.enc_features + self.num_time_dim  # Temporal.py:51-53


def param_names(spec: NetSpec) -> List[str]:
    """state_dict keys in ``parameters()`` order (probed)."""
    names: List[str] = []
    if spec.num_time_dim > 0:
        names.append("time_latents")
    for i in range(spec.num_early_layers + 1):
        names += [f"early_pts_layers.{2 * i}.weight", f"early_pts_layers.{2 * i}.bias"]
    if spec.num_late_layers > 0:
        names += ["skip_connection.0.weight", "skip_connection.0.bias"]
        for i in range(spec.num_late_layers - 1):
            names += [f"late_pts_layers.{2 * i}.weight", f"late_pts_layers.{2 * i}.bias"]
    names += ["output_linear.0.weight", "output_linear.0.bias"]
    return names


def param_shapes(spec: NetSpec) -> Dict[str, Tuple[int, ...]]:
    F, K = spec.num_filters, spec.in_features
    shapes: Dict[str, Tuple[int, ...]] = {}
    if spec.num_time_dim > 0:
        shapes["time_latents"] = (spec.num_phases, spec.num_time_dim)
    shapes["early_pts_layers.0.weight"] = (F, K)
    shapes["early_pts_layers.0.bias"] = (F,)
    for i in range(1, spec.num_early_layers + 1):
        shapes[f"early_pts_layers.{2 * i}.weight"] = (F, F)
        shapes[f"early_pts_layers.{2 * i}.bias"] = (F,)
    if spec.num_late_layers > 0:
        shapes["skip_connection.0.weight"] = (F, F + K)
        shapes["skip_connection.0.bias"] = (F,)
        for i in range(spec.num_late_layers - 1):
            shapes[f"late_pts_layers.{2 * i}.weight"] = (F, F)
            shapes[f"late_pts_layers.{2 * i}.bias"] = (F,)
    shapes["output_linear.0.weight"] = (spec.num_output_channels, F)
    shapes["output_linear.0.bias"] = (spec.num_output_channels,)
    return shapes


def init_params(spec: NetSpec, gen: torch.Generator, dtype=torch.float32) -> Dict[str, Tensor]:
    """nn.Linear default init U(+-1/sqrt(fan_in)) for W and b; latents U[0,1) (Temporal.py:26)."""
    out: Dict[str, Tensor] = {}
    shapes = param_shapes(spec)
    for name in param_names(spec):
        shp = shapes[name]
        if name == "time_latents":
            out[name] = torch.rand(shp, generator=gen, dtype=dtype)
            continue
        if name.endswith(".weight"):
            fan_in = shp[1]
        else:
            fan_in = shapes[name[: -len("bias")] + "weight"][1]
        bound = 1.0 / math.sqrt(fan_in)
        out[name] = (torch.rand(shp, generator=gen, dtype=dtype) * 2 - 1) * bound
    return out


# --------------------------------------------------------------------------------------
# a5: positional encoding + window schedules
# --------------------------------------------------------------------------------------


def freq_mask_alpha(L: int, current_iter: int, max_iter: int, window_start: int) -> Tuple[Tensor, float]:
    """FreeNeRF frequency mask (CPPN.py:144-159 / Temporal.py:189-204).

    Returns (mask f32[L], windowed_alpha).  Inactive bands are 1e-8 (not 0) because of the
    clip; the upper clip 1-1e-8 rounds to 1.0f in float32.
    """
    if current_iter < max_iter:
        mask = np.zeros(L)
        ptr = (L * current_iter) / max_iter + window_start
        ip = int(ptr)
        mask[: ip + 1] = 1.0
        mask[ip : ip + 1] = ptr - ip
        return torch.clip(torch.from_numpy(mask), 1e-8, 1 - 1e-8).float(), ptr
    return torch.ones(L).float(), L + 1


def nerfies_window(L: int, alpha: float) -> Tensor:
    """Nerfies cosine easing window (CPPN.py:137-142)."""
    bands = torch.arange(0, L)
    x = torch.clip(alpha - bands, 0.0, 1.0)
    return 0.5 * (1 + torch.cos(torch.pi * x + torch.pi))


def windowed_alpha(L: int, current_iter: int, max_iter: int) -> float:
    """CPPN.py:161-162."""
    return (L * current_iter) / max_iter


def encode(x: Tensor, spec: NetSpec, window: Optional[Tensor]) -> Tensor:
    """CPPN.pos_enc / Temporal.pos_enc (CPPN.py:112-135, Temporal.py:153-177).

    Feature order: [x, y, z, then per band k: sin(x,y,z * 2^k), sin(x,y,z * 2^k + pi/2)].
    ``cos`` is evaluated as ``sin(fl32(xb) + fl32(pi/2))`` exactly as the reference does.
    """
    L = spec.pos_enc_basis
    if spec.pos_enc == "none" or L <= 0:
        return x
    if spec.pos_enc == "fourier":
        basis = torch.cat(L * [x], dim=-1)
        value = 2 * np.pi * basis * spec.fourier_coefficients.to(x)
        return torch.cat([torch.sin(value), torch.cos(value)], dim=-1)
    batch = x.shape[:-1]
    scales = 2.0 ** torch.arange(0, L)
    xb = x[..., None, :] * scales[:, None].to(x.dtype)
    feat = torch.sin(torch.stack([xb, xb + 0.5 * torch.pi], dim=-2))
    if spec.pos_enc in ("nerfies_windowed", "free_windowed"):
        feat = window.to(x.dtype)[..., None, None] * feat
    feat = feat.reshape((*batch, -1))
    return torch.cat([x, feat], dim=-1)


# --------------------------------------------------------------------------------------
# a6 / a7: the two MLPs
# --------------------------------------------------------------------------------------


def _q(x: Tensor) -> Tensor:
    """Straight-through bf16 rounding (value rounded, gradient passed through)."""
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def _q8(x: Tensor, fmt: str) -> Tensor:
    """Round to an 8-bit float (e4m3: OCP e4m3fn, e5m2) with saturation, as v_cvt_scalef32_pk_{fp8,bf8}_f32 does under
    MODE.FP16_OVFL (round to nearest even; measured on gfx950, tools/fp8_cvt_probe.hip)."""
    dt, top = (torch.float8_e4m3fn, 448.0) if fmt == "e4m3" else (torch.float8_e5m2, 57344.0)
    return x.clamp(-top, top).to(dt).to(x.dtype)


H8_LOG2, D8_LOG2 = 2, 4          # nca_layout.hpp: NCA_H8_LOG2, NCA_D8_LOG2


def _tile_scales(g: Tensor, S: int) -> Tensor:
    """Per-sample power of two 2^(D8_LOG2 - e), where 2^e <= max |g| < 2^(e+1) over the sample's tile of 64 consecutive samples
    of its ray (g = d loss / d raw output, [R*S, 1] ray-major); tiles whose gradients are all zero get 1."""
    R = g.shape[0] // S
    nt = (S + 63) // 64
    gp = TF.pad(g.detach().reshape(R, S), (0, nt * 64 - S)).reshape(R, nt, 64)
    am = gp.abs().amax(dim=-1, keepdim=True)
    _, ex = torch.frexp(am)                      # am = m 2^ex with m in [0.5, 1): floor(log2 am) = ex - 1
    sc = torch.ldexp(torch.ones_like(am), D8_LOG2 - (ex - 1))
    sc = torch.where(am > 0, sc, torch.ones_like(sc))
    return sc.expand(R, nt, 64).reshape(R, nt * 64)[:, :S].reshape(-1, 1)


class _StagedLinear(torch.autograd.Function):
    """One F-wide layer of the bf16 HIP path, forward AND backward arithmetic, for the fp8-staging emulation:
    forward  y = bf16(x) bf16(W)^T + b (f32 accumulation);
    backward dx = bf16(dy) bf16(W) (the deltas are packed to bf16 as the next contraction's operand);
             dW, db from e5m2(bf16(dy) s_tile) / s_tile and, where the layer input crossed HBM as e4m3, from
             e4m3(bf16(x) 2^H8_LOG2) / 2^H8_LOG2 (both converted from the packed bf16 pairs) -- else from the bf16 operands.
             First layer (`first`): its input gradient is only needed for the time latents, which the kernels form in the
             weight-gradient pass (one-hot phase columns of the input block times the staged D_0, then the f32 weights)."""

    @staticmethod
    def forward(ctx, x, W, b, state, d8: Optional[str], h8: Optional[str], first: bool):
        xq, Wq = x.to(torch.bfloat16).to(x.dtype), W.to(torch.bfloat16).to(W.dtype)
        ctx.save_for_backward(x, xq, Wq, W)
        ctx.state, ctx.d8, ctx.h8, ctx.first = state, d8, h8, first
        return xq @ Wq.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, xq, Wq, W = ctx.saved_tensors
        dq = dy.to(torch.bfloat16).to(dy.dtype)
        if ctx.d8:
            sc = ctx.state["scale"]
            dd = _q8(dq * sc, ctx.d8) / sc
        else:
            dd = dq
        dx = dd @ W if ctx.first else dq @ Wq
        hh = _q8(xq * 2.0 ** H8_LOG2, ctx.h8) / 2.0 ** H8_LOG2 if ctx.h8 else xq
        return dx, dd.t() @ hh, dd.sum(0), None, None, None, None


class _StagedOut(torch.autograd.Function):
    """The F -> 1 output layer of the bf16 HIP path under the BF16-staging emulation: forward in f32 on the f32 activations (as
    the kernels do); backward: the input gradient g Wo in f32 (the previous layer rounds it to bf16), the weight gradient from
    the bf16-rounded layer input (the dgrad kernel forms it from the packed activations of the recomputed last layer)."""

    @staticmethod
    def forward(ctx, h, Wo, bo):
        ctx.save_for_backward(h, Wo)
        return h @ Wo.t() + bo

    @staticmethod
    def backward(ctx, g):
        h, Wo = ctx.saved_tensors
        return g @ Wo, g.t() @ h.to(torch.bfloat16).to(h.dtype), g.sum(0)


class _StagedTail(torch.autograd.Function):
    """Last F-wide layer + ReLU + the F -> 1 output layer of the bf16 HIP path under FP8 staging (mode 5, nca_layout.hpp): nothing
    is recomputed, and the block the weight-gradient kernel gets for the last layer is relu'(z) g without the factor Wo:
        S = e5m2(bf16(g) relu' s_tile)^T / s_tile  x  e4m3(bf16(x) 2^H8_LOG2) / 2^H8_LOG2,      s = column sums of the first factor,
        dW = Wo[f] S,  db = Wo[f] s,  dWo[f] = <bf16(W)[f], S[f]> + b[f] s[f],  dbo = sum g,
        dx = (bf16(Wo g) relu') bf16(W)                                  (the chain operand keeps Wo and is rounded to bf16)."""

    @staticmethod
    def forward(ctx, x, W, b, Wo, bo, state, d8: Optional[str], h8: Optional[str]):
        xq, Wq = x.to(torch.bfloat16).to(x.dtype), W.to(torch.bfloat16).to(W.dtype)
        z = xq @ Wq.t() + b
        ctx.save_for_backward(xq, Wq, b, Wo, z)
        ctx.state, ctx.d8, ctx.h8 = state, d8, h8
        return torch.relu(z) @ Wo.t() + bo

    @staticmethod
    def backward(ctx, g):
        xq, Wq, b, Wo, z = ctx.saved_tensors
        on = (torch.relu(z).to(torch.bfloat16) > 0).to(g.dtype)          # the stored mask: the packed (bf16) activation is positive
        chain = (g @ Wo).to(torch.bfloat16).to(g.dtype) * on              # D_{NL-1} as the sweep's operand
        gq = g.to(torch.bfloat16).to(g.dtype) * on                        # what goes to the weight-gradient kernel
        if ctx.d8:
            sc = ctx.state["scale"]
            gq = _q8(gq * sc, ctx.d8) / sc
        hh = _q8(xq * 2.0 ** H8_LOG2, ctx.h8) / 2.0 ** H8_LOG2 if ctx.h8 else xq
        S, s = gq.t() @ hh, gq.sum(0)
        wo = Wo.reshape(-1)
        dWo = ((Wq * S).sum(1) + b * s).reshape(Wo.shape)
        return chain @ Wq, wo[:, None] * S, wo * s, dWo, g.sum(0), None, None, None


def mlp(params: Dict[str, Tensor], spec: NetSpec, feats: Tensor) -> Tensor:
    """Shared body of CPPN.forward (CPPN.py:98-110) and Temporal.query_time (Temporal.py:125-134).

    With ``spec.emulate_bf16`` (used only to test the bf16 HIP kernels) the MFMA operands are rounded
    to bf16 exactly where the kernel rounds them: layer inputs (encoded features, ReLU outputs) and
    weights of the F-wide layers; biases, accumulation and the F->1 output layer stay f32."""
    def wide_layers():
        """(weight key, bias key, takes the skip input cat[feats, h]) of every F-wide layer in order (CPPN.py:98-106)."""
        out = [(f"early_pts_layers.{2 * i}.weight", f"early_pts_layers.{2 * i}.bias", False) for i in range(spec.num_early_layers + 1)]
        if spec.num_late_layers > 0:
            if spec.num_time_dim > 0:
                raise UnboundLocalError("Temporal with num_late_layers > 0 has no output in the reference (Temporal.py:128-135)")
            out.append(("skip_connection.0.weight", "skip_connection.0.bias", True))
            out += [(f"late_pts_layers.{2 * i}.weight", f"late_pts_layers.{2 * i}.bias", False) for i in range(spec.num_late_layers - 1)]
        return out

    if spec.emulate_bf16 and spec.emulate_fp8_stage > 0:
        layers = wide_layers()
        NL = len(layers)
        state: dict = {}
        fd, fh = spec.emulate_stage_formats or (None, None)
        h = feats
        for i, (wk, bk, skip) in enumerate(layers):
            last = i == NL - 1
            d8 = None if (last and spec.emulate_onchip_last) else (None if fd == "bf16" else fd)     # on chip: bf16 registers, nothing is staged
            # every staged layer's input is e4m3 when its weight gradient is formed: the hidden blocks cross HBM as e4m3, the bf16
            # input block is rounded to e4m3 inside the weight-gradient kernel.  ("bf16", "bf16"): the BF16 store of round 5 -- the same
            # mode-5 arithmetic (nothing recomputed, the last layer's block without Wo, dWo from the sums) with nothing rounded to 8 bits
            h8 = (None if fh == "bf16" else fh) if d8 is not None else None
            x = torch.cat([feats, h], dim=-1) if skip else h          # (a skip layer's weight gradient is two jobs over the same two stored blocks)
            if last and fd is not None and NL >= 2:
                # fp8 staging (a store needs a hidden layer): last layer and output layer as the mode-5 kernels treat them
                raw = _StagedTail.apply(x, params[wk], params[bk], params["output_linear.0.weight"], params["output_linear.0.bias"], state, d8, h8)
                break
            h = torch.relu(_StagedLinear.apply(x, params[wk], params[bk], state, d8, h8, i == 0))
        else:
            raw = _StagedOut.apply(h, params["output_linear.0.weight"], params["output_linear.0.bias"])
        if raw.requires_grad:      # d loss / d raw arrives before the layers' backward runs: fix the tile scales there
            raw.register_hook(lambda g: state.__setitem__("scale", _tile_scales(g, spec.emulate_fp8_stage)))
        return raw
    if spec.emulate_bf16:
        h = feats
        for wk, bk, skip in wide_layers():
            x = torch.cat([feats, h], dim=-1) if skip else h
            h = torch.relu(TF.linear(_q(x), _q(params[wk]), params[bk]))
        return TF.linear(h, params["output_linear.0.weight"], params["output_linear.0.bias"])
    h = feats
    for i in range(spec.num_early_layers + 1):
        h = torch.relu(TF.linear(h, params[f"early_pts_layers.{2 * i}.weight"], params[f"early_pts_layers.{2 * i}.bias"]))
    if spec.num_late_layers > 0:
        if spec.num_time_dim > 0:
            # Temporal.py:128-135: the `else` binds to this `if`, so `outputs` is never
            # assigned when num_late_layers > 0 -> UnboundLocalError in the reference.
            raise UnboundLocalError("Temporal with num_late_layers > 0 has no output in the reference (Temporal.py:128-135)")
        h = torch.relu(TF.linear(torch.cat([feats, h], dim=-1), params["skip_connection.0.weight"], params["skip_connection.0.bias"]))
        for i in range(spec.num_late_layers - 1):
            h = torch.relu(TF.linear(h, params[f"late_pts_layers.{2 * i}.weight"], params[f"late_pts_layers.{2 * i}.bias"]))
    return TF.linear(h, params["output_linear.0.weight"], params["output_linear.0.bias"])


def static_forward(params: Dict[str, Tensor], spec: NetSpec, x: Tensor, window: Optional[Tensor]) -> Tensor:
    """CPPN.forward (CPPN.py:88-110): f32[n,3] -> f32[n,1]."""
    return mlp(params, spec, encode(x, spec, window))


def dynamic_forward(params: Dict[str, Tensor], spec: NetSpec, x: Tensor, ts: Tensor, window: Optional[Tensor]) -> Tensor:
    """Temporal.forward_composite (Temporal.py:138-151): latent gather + query_time (113-136)."""
    lat = params["time_latents"][ts.flatten().long()]
    return mlp(params, spec, torch.cat([encode(x, spec, window), lat], dim=-1))


# --------------------------------------------------------------------------------------
# a2 / a3: depth jitter and query points
# --------------------------------------------------------------------------------------


def depth_values(near: float, far: float, n: int) -> Tensor:
    """data_helpers.py:167-171."""
    t = torch.linspace(0.0, 1.0, n)
    return near * (1.0 - t) + far * t


def stratified_depths(z: Tensor, t_rand: Tensor) -> Tensor:
    """randomize_depth with the uniform draw injected (model_helpers.py:3-12)."""
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


def query_points(origins: Tensor, directions: Tensor, z: Tensor) -> Tensor:
    """model_helpers.py:118-120: computed in the rays' dtype (f64 in the real script), then .float().

    ``z`` is [S] (shared) or [R, S] (fine pass).  Output f32[R*S, 3], ray-major (n = r*S + s).
    """
    pts = origins[..., None, :] + directions[..., None, :] * z[..., :, None]
    return pts.reshape((-1, 3)).float()


# --------------------------------------------------------------------------------------
# a8 / a9: compositing
# --------------------------------------------------------------------------------------


def activation(name: str):
    """get_activation_func (model_helpers.py:63-70): anything but 'softplus'/'clamp' is Sigmoid."""
    if name == "softplus":
        return TF.softplus
    if name == "clamp":
        return lambda v: TF.hardtanh(TF.softplus(v), min_val=0.0, max_val=1.0)
    return torch.sigmoid


def ray_dists(z: Tensor, dtype: torch.dtype) -> Tensor:
    """model_helpers.py:73-74: last interval is 1e-10 in the ray directions' dtype."""
    tail = torch.tensor([1e-10], dtype=dtype).expand(z[..., :1].shape)
    return torch.cat((z[..., 1:] - z[..., :-1], tail), dim=-1)


def composite(raw_s: Tensor, raw_d: Tensor, I0: Tensor, dirs: Tensor, z: Tensor, act: str = "softplus", scale: float = 1e-2):
    """render_volume_density_composite (model_helpers.py:72-84)."""
    dists = ray_dists(z, dirs.dtype)
    f = activation(act)
    sig_s = f(raw_s[..., -1]) * scale
    sig_d = f(raw_d[..., -1]) * scale
    pix = I0 - torch.sum((sig_s + sig_d) * dists, dim=-1)
    return pix, sig_s, sig_d, dists


def composite_single(raw: Tensor, I0: Tensor, dirs: Tensor, z: Tensor, act: str = "softplus", scale: float = 1e-2):
    """render_volume_density (model_helpers.py:86-97): returns the UN-scaled sigma."""
    dists = ray_dists(z, dirs.dtype)
    sig = activation(act)(raw[..., -1])
    pix = I0 - torch.sum(sig * dists * scale, dim=-1)
    return pix, sig, dists


# --------------------------------------------------------------------------------------
# a12: hierarchical sampling
# --------------------------------------------------------------------------------------


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    """sample_pdf with the uniform draw ``u`` injected (model_helpers.py:162-187)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_lo, bin_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_lo + (u - cdf_lo) / denom * (bin_hi - bin_lo)


# --------------------------------------------------------------------------------------
# a10 / a11: the per-step prediction functions
# --------------------------------------------------------------------------------------


def predict_static(params, spec, window, origins, directions, I0, z_jit, act="softplus"):
    """obtain_train_predictions_static minus the RNG draw (model_helpers.py:99-113)."""
    pts = query_points(origins, directions, z_jit)
    raw = static_forward(params, spec, pts, window).reshape(list(origins.shape[:-1]) + [z_jit.shape[-1], spec.num_output_channels])
    return composite_single(raw, I0, directions, z_jit, act)


def predict_iter(ps, spec_s, win_s, pd, spec_d, win_d, origins, directions, phases, I0, z_jit, act="softplus",
                 fine=None):
    """obtain_train_predictions_iter minus the RNG draws (model_helpers.py:115-160).

    ``phases`` is [R, S] (run_composite.py:265).  ``fine`` is None or a dict with keys
    ps, spec_s, win_s, pd, spec_d, win_d, n_fine, u (the injected sample_pdf draw [R, n_fine]) and optionally
    detach_depths (see below).
    Returns the reference's 8-tuple.
    """
    R, S = origins.shape[0], z_jit.shape[0]
    dt = next(iter(ps.values())).dtype       # f32 as the reference; f64 parameters give the f64 check run of the tests
    pts = query_points(origins, directions, z_jit).to(dt)
    ph = phases.flatten().int()
    raw_s = static_forward(ps, spec_s, pts, win_s).reshape(R, S, -1)
    raw_d = dynamic_forward(pd, spec_d, pts, ph, win_d).reshape(R, S, -1)
    pix, sig_s, sig_d, dists = composite(raw_s, raw_d, I0, directions, z_jit, act)
    out_f = (None, None, None, None)
    if fine is not None and fine["n_fine"] > 0:
        nf = fine["n_fine"]
        tot = S + nf
        eps = torch.ones_like(sig_s[:, :1]) * 1e-10
        tsum = sig_s + sig_d
        w = torch.cat([eps, torch.abs(tsum[:, 1:] - tsum[:, :-1])], dim=-1)
        w = w / torch.max(w)  # batch-wide max (model_helpers.py:139)
        zb = z_jit[None, :].repeat(R, 1)
        mid = 0.5 * (zb[..., 1:] + zb[..., :-1])
        zf = sample_pdf(mid, w[..., 1:-1], fine["u"])
        z_all, _ = torch.sort(torch.cat([zf, zb.detach()], -1), -1)
        if fine.get("z_all") is not None:     # test hook: depths sampled elsewhere (constants), e.g. by the kernel under test
            z_all = fine["z_all"]
        if fine.get("detach_depths", False):
            # NOT the reference's behaviour (its autograd differentiates through sample_pdf, the sort and the query points
            # into the coarse nets): only there to quantify the drop-in's documented deviation in the tests.
            z_all = z_all.detach()
        pts_f = query_points(origins, directions, z_all).to(dt)
        z0 = z_all[0, :]  # dists from ray 0 only (model_helpers.py:150)
        ph_f = phases[:, 0, None].repeat(1, tot).flatten()
        raw_sf = static_forward(fine["ps"], fine["spec_s"], pts_f, fine["win_s"]).reshape(R, tot, -1)
        raw_df = dynamic_forward(fine["pd"], fine["spec_d"], pts_f, ph_f, fine["win_d"]).reshape(R, tot, -1)
        out_f = composite(raw_sf, raw_df, I0, directions, z0, act)
    return (pix, sig_s, sig_d, dists) + tuple(out_f)


# --------------------------------------------------------------------------------------
# a13 / a14 / a15: losses and schedules
# --------------------------------------------------------------------------------------


def weighted_mse(pred: Tensor, gt: Tensor, w: Tensor) -> Tensor:
    """weighted_MSELoss.forward (model_helpers.py:284-288); caller takes .mean()."""
    return ((pred - gt) ** 2) * w


def blend_ratio(sig_s: Tensor, sig_d: Tensor):
    """compute_ratio (model_helpers.py:189-198)."""
    with torch.no_grad():
        ms, md = torch.max(sig_s), torch.max(sig_d)
    return sig_d / (sig_s + sig_d + 1e-10), ms, md


def blendw_entropy(blendw: Tensor, clip: float = 1e-19, skew: float = 1.0) -> Tensor:
    """compute_blendw_loss (model_helpers.py:200-204)."""
    b = torch.clip(blendw ** skew, min=clip, max=1 - clip)
    rb = torch.clip(1 - b, min=clip)
    return torch.mean(-(b * torch.log(b) + rb * torch.log(rb)), dim=-1).mean()


def ray_entropy(sig: Tensor, dists: Tensor, mask_thre: float = 0.1, clip: float = 1e-19, use_weighting=False,
                weighted_pixs=(), weighted_thresh: float = 0.25):
    """compute_sigma_s_ray_loss (model_helpers.py:206-224)."""
    sd = sig * dists
    tot = torch.sum(sd, dim=-1, keepdim=True)
    mask = torch.where(tot < mask_thre, 0.0, 1.0).flatten().int()
    if len(weighted_pixs) > 0 and use_weighting:
        wm = torch.zeros(mask.shape).int()
        wm[: weighted_pixs.shape[0]] = torch.where(weighted_pixs > 1 + weighted_thresh, 1.0, 0.0).int()
        mask = torch.bitwise_or(wm, mask)
    p = sd / torch.clip(tot, min=clip)
    ent = mask * -torch.sum(p * torch.log(p + 1e-10), dim=-1)
    return ent.mean(), tot.mean()


def occlusion(sig: Tensor, dists: Tensor, reg_perc: float = 0.1, use_back: bool = False) -> Tensor:
    """compute_occl_loss (model_helpers.py:226-248).  With use_back=False the back mask is all
    ones and is OR-ed in, so the result is the mean ray sum."""
    cum = torch.cumsum(dists, dim=0).unsqueeze(0).repeat((sig.shape[0], 1))
    front = reg_perc * cum[-1, -1]
    back = (1 - reg_perc) * cum[-1, -1]
    m_front = torch.where(cum < front, 1.0, 0.0).int()
    m_back = torch.ones(m_front.shape)
    if use_back:
        m_back = torch.where(cum > back, 1.0, 0.0)
    mask = torch.bitwise_or(m_front, m_back.int())
    return torch.sum(sig * dists * mask, dim=-1).mean()


@dataclass
class LossArgs:
    """The run_args fields compute_losses reads (model_helpers.py:250-262; composite.txt)."""

    favor_s_opt: Optional[str] = None
    skewness_val: float = 1.0
    entro_mask_thre: float = 1e-4
    entro_use_weighting: bool = True
    entro_weighted_thresh: float = 0.03
    occl_reg_perc: float = 0.2


def compute_losses(sig_s: Tensor, sig_d: Tensor, dists: Tensor, weighted_pixs: Tensor, a: LossArgs):
    """compute_losses (model_helpers.py:250-262) -> the reference's 11-tuple."""
    bw, ms, md = blend_ratio(sig_s, sig_d)
    favor = blendw_entropy(bw, skew=a.skewness_val)
    s_ent, s_sum = ray_entropy(sig_s, dists, mask_thre=a.entro_mask_thre)
    d_ent, d_sum = ray_entropy(sig_d, dists, mask_thre=a.entro_mask_thre, use_weighting=a.entro_use_weighting,
                               weighted_pixs=weighted_pixs, weighted_thresh=a.entro_weighted_thresh)
    occl = occlusion(sig_d, dists, a.occl_reg_perc)
    l1 = torch.sum(sig_s * dists, dim=-1).sum()
    l2 = torch.sum((sig_s * dists) ** 2, dim=-1).sum()
    return bw.mean(), ms, md, favor, s_ent, s_sum, d_ent, d_sum, occl, l1, l2


def linear_param_decay(curr_iter, start_weight, end_weight, steps, delay_steps=0):
    """model_helpers.py:264-269."""
    if curr_iter < delay_steps:
        return 0
    alpha = min((curr_iter - delay_steps) / steps, 1.0)
    return (1.0 - alpha) * start_weight + alpha * end_weight


@dataclass
class ScheduleArgs:
    """composite.txt:50-66 loss-weight schedule."""

    favor_s_weight_start: float = 1e-12
    favor_s_weight_end: float = 1e-10
    favor_s_weight_delay_steps: int = 40000
    dynamic_entro_weight_start: float = 1e-10
    dynamic_entro_weight_end: float = 1e-8
    occl_weight_start: float = 1e-8
    occl_weight_end: float = 1e-4
    l1_weight_start: float = 1e-8
    l1_weight_end: float = 1e-15
    hyperparam_decay_steps: int = 100000


def loss_weights(n_iter: int, s: ScheduleArgs):
    """run_composite.py:276-279."""
    fav = linear_param_decay(n_iter, s.favor_s_weight_start, s.favor_s_weight_end, s.hyperparam_decay_steps, s.favor_s_weight_delay_steps)
    ent = linear_param_decay(n_iter, s.dynamic_entro_weight_start, s.dynamic_entro_weight_end, s.hyperparam_decay_steps)
    occ = linear_param_decay(n_iter, s.occl_weight_start, s.occl_weight_end, s.hyperparam_decay_steps, s.favor_s_weight_delay_steps)
    l1 = linear_param_decay(n_iter, s.l1_weight_start, s.l1_weight_end, s.hyperparam_decay_steps)
    return fav, ent, occ, l1


def composite_total_loss(pix, sig_s, sig_d, dists, gt, wpix, n_iter, largs: LossArgs, sargs: ScheduleArgs):
    """run_composite.py:287-292: pixel + weighted regularisers (static entropy is NOT in the loss)."""
    pixel = weighted_mse(pix, gt, wpix).mean()
    terms = compute_losses(sig_s, sig_d, dists, wpix, largs)
    fav_w, ent_w, occ_w, l1_w = loss_weights(n_iter, sargs)
    favor, d_ent, occl, l1, l2 = terms[3], terms[6], terms[8], terms[9], terms[10]
    loss = pixel + fav_w * favor + ent_w * d_ent + occ_w * occl + l1_w * l2 + l1_w * l1
    return loss, pixel, terms


def static_total_loss(pix, sig, dists, gt, wpix, occl_weight: float, occl_reg_perc: float):
    """run_nerf.py:227-230."""
    pixel = weighted_mse(pix, gt, wpix).mean()
    occl = torch.sum(occlusion(sig, dists, occl_reg_perc))
    return pixel + occl_weight * occl, pixel, occl


# --------------------------------------------------------------------------------------
# a17: ray geometry and the ray table
# --------------------------------------------------------------------------------------


def _rx(a):
    return np.array([[1, 0, 0, 0], [0, np.cos(a), -np.sin(a), 0], [0, np.sin(a), np.cos(a), 0], [0, 0, 0, 1]])


def _rz(a):
    return np.array([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])


def pose_tigre(theta: float, phi: float, dso: float) -> np.ndarray:
    """source_matrix_tigre (train/proj_helpers.py:50-63): Rz(-theta) Rz(pi/2) Rx(phi) Rx(-pi/2) T([0,0,-DSO])."""
    rot = _rz(-np.deg2rad(theta)) @ (_rz(np.pi / 2) @ _rx(np.deg2rad(phi))) @ _rx(-np.pi / 2)
    T = np.identity(4)
    T[:3, 3] = [0, 0, -dso]
    return rot @ T


def ray_values_tigre(theta: float, phi: float, geo: dict) -> Tuple[np.ndarray, np.ndarray]:
    """get_ray_values_tigre (train/proj_helpers.py:65-90) -> (origins, directions) f32[W, H, 3].

    Directions are NOT normalised.  Element [w, h] belongs to detector column w, row h.
    """
    pose = torch.from_numpy(pose_tigre(theta, phi, geo["DSO"])).float()
    W, H = geo["nDetector"]
    ii, jj = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="xy")
    uu = (ii.t() + 0.5 - W / 2) * geo["dDetector"][0] + geo["offDetector"][0]
    vv = (jj.t() + 0.5 - H / 2) * geo["dDetector"][1] + geo["offDetector"][1]
    dirs = torch.stack([uu / geo["DSD"], vv / geo["DSD"], torch.ones_like(uu)], -1)
    rd = torch.sum(torch.matmul(pose[:3, :3], dirs[..., None]), -1)
    ro = pose[:3, -1].expand(rd.shape)
    return ro.numpy(), rd.numpy()


def denormalize_image(img: np.ndarray, W: int, H: int, mm: Sequence[float]) -> np.ndarray:
    """data_helpers.py:129-139 (reshape(W,H).T, then undo min-max if the image is in [0,1])."""
    img = img.reshape((W, H)).T
    if int(np.min(img)) == 0 and int(np.max(img)) == 1:
        return img * (mm[1] - mm[0]) + mm[0]
    return img


def build_ray_table(frames: List[dict], images: List[np.ndarray], var_images: List[np.ndarray], geo: dict,
                    weighted_loss_max: float) -> Tuple[np.ndarray, np.ndarray]:
    """prepare_data_for_loader_tigre with the file loads injected (data_helpers.py:141-165).

    Returns rays f64[N_img*W*H, 4, 3] (rows: origin, direction, pixel x3, weight x3) and
    phases i64[N].  Ray id = img*W*H + w*H + h.
    """
    W, H = geo["nDetector"]
    rays = np.stack([np.stack(ray_values_tigre(f["theta"], f["phi"], geo), 0) for f in frames], 0)
    imgs = np.stack([denormalize_image(im, W, H, f["img_min_max"]) for im, f in zip(images, frames)], 0)
    imgs = np.repeat(imgs[:, None, :, :, None], 3, axis=-1)
    wimg = np.stack([v.reshape((W, H)).T for v in var_images], 0)
    wimg = (wimg - 1) * weighted_loss_max + 1
    wimg = np.repeat(wimg[:, None, :, :, None], rays.shape[-1], axis=-1)
    ph = np.array([f["heart_phase"] for f in frames])
    ph = np.tile(ph[:, None, None], (W, H))
    table = np.concatenate([rays, imgs, wimg], 1)
    table = np.transpose(table, [0, 2, 3, 1, 4])
    return np.reshape(table, [-1, table.shape[-2], table.shape[-1]]), np.reshape(ph, [-1])


# --------------------------------------------------------------------------------------
# reference-equivalent CPU training step (used as the CPU baseline by bench.py)
# --------------------------------------------------------------------------------------


class OracleTrainer:
    """One reference-equivalent composite training step on CPU: forward, all losses, autograd
    backward, Adam + LinearLR (run_composite.py:209-215, 283-308).  Parameters are leaf
    tensors; arithmetic is the same torch CPU ops the reference executes."""

    def __init__(self, ps, spec_s, pd, spec_d, lr=1e-3, lr_end_factor=0.01, lr_decay_steps=150000,
                 largs: Optional[LossArgs] = None, sargs: Optional[ScheduleArgs] = None, act="softplus",
                 window_decay_steps=150000):
        self.spec_s, self.spec_d = spec_s, spec_d
        self.ps = {k: v.clone().requires_grad_(True) for k, v in ps.items()}
        self.pd = {k: v.clone().requires_grad_(True) for k, v in pd.items()}
        plist = list(self.pd.values()) + list(self.ps.values())  # run_composite.py:192
        self.opt = torch.optim.Adam([{"params": plist, "lr": lr}], lr=lr)
        self.sched = torch.optim.lr_scheduler.LinearLR(self.opt, start_factor=1, end_factor=lr_end_factor, total_iters=lr_decay_steps)
        self.largs = largs or LossArgs()
        self.sargs = sargs or ScheduleArgs()
        self.act = act
        self.window_decay_steps = window_decay_steps

    def windows(self, n_iter):
        def one(spec):
            if spec.pos_enc == "free_windowed":
                return freq_mask_alpha(spec.pos_enc_basis, n_iter, self.window_decay_steps, spec.pos_enc_window_start)[0]
            if spec.pos_enc == "nerfies_windowed":
                return nerfies_window(spec.pos_enc_basis, windowed_alpha(spec.pos_enc_basis, n_iter, self.window_decay_steps))
            return None
        return one(self.spec_s), one(self.spec_d)

    def step(self, n_iter, origins, directions, phases_rs, I0, z_jit, gt, wpix):
        win_s, win_d = self.windows(n_iter)
        pix, sig_s, sig_d, dists = predict_iter(self.ps, self.spec_s, win_s, self.pd, self.spec_d, win_d, origins, directions,
                                                phases_rs, I0, z_jit, self.act)[:4]
        loss, pixel, terms = composite_total_loss(pix, sig_s, sig_d, dists, gt, wpix, n_iter, self.largs, self.sargs)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        self.sched.step()
        return loss.detach(), pixel.detach(), terms


class OracleStaticTrainer:
    """One reference-equivalent iteration of the static-only loop (train/run_nerf.py:186-231): stratified depths,
    static render, weighted MSE + occl_weight_start * sum(compute_occl_loss), Adam + LinearLR."""

    def __init__(self, ps, spec, lr=1e-3, lr_end_factor=0.01, lr_decay_steps=150000, occl_weight_start=1e-8, occl_reg_perc=0.2,
                 act="softplus", window_decay_steps=150000):
        self.spec = spec
        self.ps = {k: v.clone().requires_grad_(True) for k, v in ps.items()}
        plist = list(self.ps.values())
        self.opt = torch.optim.Adam([{"params": plist, "lr": lr}], lr=lr)
        self.sched = torch.optim.lr_scheduler.LinearLR(self.opt, start_factor=1, end_factor=lr_end_factor, total_iters=lr_decay_steps)
        self.occl_weight_start, self.occl_reg_perc, self.act = occl_weight_start, occl_reg_perc, act
        self.window_decay_steps = window_decay_steps

    def window(self, n_iter):
        spec = self.spec
        if spec.pos_enc == "free_windowed":
            return freq_mask_alpha(spec.pos_enc_basis, n_iter, self.window_decay_steps, spec.pos_enc_window_start)[0]
        if spec.pos_enc == "nerfies_windowed":
            return nerfies_window(spec.pos_enc_basis, windowed_alpha(spec.pos_enc_basis, n_iter, self.window_decay_steps))
        return None

    def loss(self, n_iter, origins, directions, I0, z_jit, gt, wpix):
        pix, sig, dists = predict_static(self.ps, self.spec, self.window(n_iter), origins, directions, I0, z_jit, self.act)
        pixel = weighted_mse(pix, gt, wpix).mean()
        occl = torch.sum(occlusion(sig, dists, self.occl_reg_perc))
        return pixel + self.occl_weight_start * occl, pixel, occl, pix, sig

    def step(self, n_iter, origins, directions, I0, z_jit, gt, wpix):
        loss, pixel, occl, _, _ = self.loss(n_iter, origins, directions, I0, z_jit, gt, wpix)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        self.sched.step()
        return loss.detach(), pixel.detach(), occl.detach()

