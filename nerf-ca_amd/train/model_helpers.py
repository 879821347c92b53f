"""Drop-in for the reference's ``train/model_helpers.py``: same function names, arguments and return
values, with the ray path routed through the fused HIP kernels.

  obtain_train_predictions_iter / _static  -> ONE fused launch per call (sampling of query points,
      positional encoding, both MLPs, activation and the ray sum never leave the chip); the
      reference's Python chunk loop (get_minibatches*, model_helpers.py:14-61) has no counterpart
      because nothing is materialised per chunk -- ``batch_size`` is accepted and ignored.
  get_predictions_static / _composite      -> fused point kernels.
  render_volume_density[_composite]        -> elementwise + row sum on raw fields the caller already
      holds (the evaluation path of run_composite.py:361, 407-413).

An extra keyword ``t_rand`` on the two ``obtain_*`` functions injects the stratified-sampling draw
(the reference draws it internally with ``torch.rand``); tests use it to replay golden vectors.
"""
import torch

from .. import fused as _fused
from ..losses import (binary_entropy_of_blend as compute_blendw_loss, blend_weight as _blend_weight,  # noqa: F401
                      occlusion as compute_occl_loss, ray_entropy as compute_sigma_s_ray_loss)
from ..schedules import exp_param_decay, linear_param_decay  # noqa: F401


def compute_losses(static_sigma, temp_sigma, dists, weighted_pixs, run_args):
    """model_helpers.py:250-262 -> the reference's 11-tuple, from the fused HIP loss kernel under autograd (nca_loss_fwd_bwd: values
    forward; backward in term-gradient mode, the eleven upstream scalars weighting the terms' gradients).  GPU tensors only."""
    return _fused.loss_terms(static_sigma, temp_sigma, dists, weighted_pixs, run_args)


class weighted_MSELoss(torch.nn.Module):
    """model_helpers.py:284-288: (preds - gts)^2 * weights, elementwise (the caller takes .mean()); a HIP kernel under autograd."""

    def forward(self, preds, gts, weights):
        return _fused.weighted_sq_err(preds, gts, weights)


def compute_ratio(sigma_s, sigma_d, favor_s_opt=None, sigma_s_max=None, sigma_d_max=None, weight_max=0.05):
    return _blend_weight(sigma_s, sigma_d)


def randomize_depth(z_vals, device, t_rand=None):
    """Stratified jitter inside each depth bin (model_helpers.py:3-12): ONE vector per step, shared by
    all rays.  The uniform draw comes from the CPU generator, as in the reference."""
    mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    hi = torch.cat([mid, z_vals[..., -1:]], -1)
    lo = torch.cat([z_vals[..., :1], mid], -1)
    u = torch.rand(z_vals.shape) if t_rand is None else t_rand
    return (lo + (hi - lo) * u.to(device)).to(device)


def get_minibatches(inputs, chunksize=1024 * 8):
    return [[inputs[i:i + chunksize]] for i in range(0, inputs.shape[0], chunksize)]


def get_minibatches_time(inputs, time_inputs, chunksize=1024 * 8):
    return [[inputs[i:i + chunksize], time_inputs[i:i + chunksize]] for i in range(0, inputs.shape[0], chunksize)]


def get_predictions_static(static_model, flattened_query_points, chunksize):
    """f32[n,3] -> f32[n,1] (model_helpers.py:28-39); one fused launch, no chunk loop."""
    return static_model(flattened_query_points)


def get_predictions_composite(static_model, temp_model, flattened_query_points, flattened_time_points, chunksize, use_nerf_acc=False):
    """model_helpers.py:41-61 -> (static f32[n,1], dynamic f32[n,1])."""
    s = None if use_nerf_acc else static_model(flattened_query_points)
    d = temp_model.forward_composite(flattened_query_points, flattened_time_points)
    return s, d


def get_activation_func(output_activation):
    """model_helpers.py:63-70: only the exact strings 'softplus' and 'clamp' are special."""
    if output_activation == "softplus":
        return torch.nn.Softplus()
    if output_activation == "clamp":
        return lambda x: torch.nn.functional.hardtanh(torch.nn.Softplus()(x), min_val=0.0, max_val=1.0)
    return torch.nn.Sigmoid()


def _interval_lengths(depth_values, like):
    """cat(z[1:] - z[:-1], 1e-10) with the tail in the ray directions' dtype (model_helpers.py:73-74)."""
    tail = torch.full((1,), 1e-10, dtype=like.dtype, device=like.device).expand(depth_values[..., :1].shape)     # (a fill, not a host copy: capturable)
    return torch.cat((depth_values[..., 1:] - depth_values[..., :-1], tail), dim=-1)


def _require_composable(field, depth_values, what):
    """The stand-alone compositing kernels take [R,S,C] raw fields on the GPU and ONE depth vector shared by all rays (how
    the reference calls them, run_composite.py:361, 407-413).  Anything else is an explicit error: there is no torch
    implementation behind these functions."""
    _fused._require_cuda(field, what)
    if field.dim() != 3 or depth_values.dim() != 1 or field.shape[-2] != depth_values.shape[0]:
        raise _fused._capi.NcaError(f"{what}: expected a raw field [R,S,C] and depth values [S], got {tuple(field.shape)} and "
                                    f"{tuple(depth_values.shape)} (per-ray depth values are only supported by the fused "
                                    f"obtain_train_predictions_* path)")


def render_volume_density_composite(static_radiance_field, temp_radiance_field, initial_intensities, ray_directions, depth_values,
                                    output_activation="softplus", scale_value=1e-2):
    """model_helpers.py:72-84 as one HIP kernel (nca_composite_fwd / _bwd under autograd)."""
    _require_composable(static_radiance_field, depth_values, "render_volume_density_composite")
    _fused._require_cuda(temp_radiance_field, "render_volume_density_composite")
    dists = _interval_lengths(depth_values, ray_directions)
    f64 = ray_directions.dtype == torch.float64
    pix, ss, sd = _fused.composite_raw(static_radiance_field[..., -1], temp_radiance_field[..., -1], initial_intensities, dists,
                                       output_activation, False, scale_value, f64)
    return pix, ss, sd, dists


def render_volume_density(radiance_field, initial_intensities, ray_directions, depth_values, output_activation="softplus", scale_value=1e-2):
    """model_helpers.py:86-97 (returns the UN-scaled sigma) as one HIP kernel."""
    _require_composable(radiance_field, depth_values, "render_volume_density")
    dists = _interval_lengths(depth_values, ray_directions)
    f64 = ray_directions.dtype == torch.float64
    pix, sa = _fused.composite_raw(radiance_field[..., -1], None, initial_intensities, dists, output_activation, True, scale_value, f64)
    return pix, sa, dists


def obtain_train_predictions_static(static_model, batch_origins, batch_directions, batch_initial_intensities, depth_values,
                                    output_activation, batch_size, device, t_rand=None):
    """model_helpers.py:99-113 -> (pix[R], un-scaled sigma[R,S], dists[S]) in one fused launch."""
    z = randomize_depth(depth_values, device, t_rand)
    dists = _interval_lengths(z, batch_directions)
    pix, sigma = _fused.render_rays(static_model, None, batch_origins, batch_directions, None, batch_initial_intensities, z, dists,
                                    act=output_activation, single=True)
    return pix, sigma, dists


class _BatchMax(torch.autograd.Function):
    """torch.max(weights) of model_helpers.py:139 for a ray-sharded batch: the maximum over ALL ranks' rays (``reduce_max``
    all-reduces the local maximum), and in the backward the summed upstream gradient of all ranks goes to the element that
    attains it, on the rank that holds it -- what autograd does with the single-process maximum."""

    @staticmethod
    def forward(ctx, w, reduce_max):
        local = w.max()
        glob = local.detach().clone().reshape(1)
        if reduce_max is not None:
            reduce_max(glob)
        ctx.reduce_max = reduce_max
        ctx.save_for_backward(w, glob)
        return glob.reshape(())

    @staticmethod
    def backward(ctx, g):
        w, glob = ctx.saved_tensors
        hit = (w == glob)
        # [upstream gradient, number of elements that attain the maximum]: both are sums over the rays of ALL ranks (torch.max
        # hands its gradient evenly to ties, wherever they sit)
        both = torch.stack([g.detach().reshape(()).to(torch.float64), hit.sum().to(torch.float64)])
        if ctx.reduce_max is not None:
            red = getattr(ctx.reduce_max, "sum", None)
            if red is None:
                raise RuntimeError("the backward of a ray-sharded batch maximum needs a reducer with a `sum` method (all-reduce SUM in place)")
            red(both)
        each = (both[0] / both[1].clamp(min=1.0)).to(w.dtype)
        return torch.where(hit, each, torch.zeros((), dtype=w.dtype, device=w.device)), None


def obtain_train_predictions_iter(static_model_coarse, temp_model_coarse, static_model_fine, temp_model_fine, batch_origins,
                                  batch_directions, batch_phases, batch_initial_intensities, depth_values, output_activation,
                                  batch_size, depth_samples_per_ray_fine, device, t_rand=None, u_fine=None, reduce_max=None,
                                  depth_gradients=None):
    """model_helpers.py:115-160 -> the reference's 8-tuple; the coarse pass is one fused launch.

    Fine pass (``depth_samples_per_ray_fine > 0``, off in the reference's configs): all eight outputs equal the reference's
    (goldens with injected draws).  The reference never detaches the sampled depths, so its autograd also differentiates the
    fine losses through ``sample_pdf`` / ``sort`` / the query points / the positional encoding back into the COARSE nets (and
    through the ray-0 ``dists``).  ``depth_gradients`` (default: on whenever autograd is recording) does the same: the HIP
    sampler gets its backward into the coarse densities (``nca_fine_depths_bwd``) and the fused render returns
    d loss / d depth (``nca_render_bwd_depth``) and d loss / d dists.  With
    ``depth_gradients=False`` the depths come from the HIP sampling kernel and are constants of the
    step, as in NeRF's own hierarchical sampling: forward values identical, fine-net gradients identical, the coarse nets
    then learn from the coarse terms only (the through-depth term is ~1e4 times their regular gradient,
    tests/test_hip_parity.py::test_trainer_with_fine_pass_vs_oracle).  ``u_fine`` injects the uniform draw of ``sample_pdf``;
    ``reduce_max`` (ray-sharded batches) makes the batch-wide weight maximum global, see ``fused.fine_depths``; with depth
    gradients it should also carry a ``sum`` attribute (all-reduce SUM, in place) for the backward of that maximum, as
    ``trainer._MaxReducer`` does."""
    z = randomize_depth(depth_values, device, t_rand)
    dists_c = _interval_lengths(z, batch_directions)
    pix_c, sig_s_c, sig_d_c = _fused.render_rays(static_model_coarse, temp_model_coarse, batch_origins, batch_directions, batch_phases,
                                                 batch_initial_intensities, z, dists_c, act=output_activation)
    pix_f = sig_s_f = sig_d_f = dists_f = None
    if depth_samples_per_ray_fine > 0:
        R, n_coarse = pix_c.shape[0], z.shape[0]
        if u_fine is None:                                         # the draw comes from the CPU generator, as in the reference
            u_fine = torch.rand(R, depth_samples_per_ray_fine)
        if depth_gradients is None:            # the reference's behaviour whenever autograd is recording
            depth_gradients = bool(torch.is_grad_enabled() and sig_s_c.requires_grad)
        if depth_gradients:
            # the sampled depths stay in the autograd graph as in the reference (model_helpers.py:135-146): HIP sampler forward,
            # HIP backward into the coarse densities (nca_fine_depths_bwd); the fused render below returns d loss / d depth
            z_all = _fused.fine_depths_autograd(sig_s_c, sig_d_c, z, u_fine, reduce_max=reduce_max)
        else:
            # weights (batch-wide max, :139), sample_pdf and sort(cat[fine, coarse]) in one HIP pass per ray
            z_all = _fused.fine_depths(sig_s_c, sig_d_c, z, u_fine, reduce_max=reduce_max)
        z0 = z_all[0, :]                                           # dists of ray 0 for every ray (model_helpers.py:150)
        dists_f = _interval_lengths(z0, batch_directions)
        phase_per_ray = batch_phases[:, 0] if batch_phases.dim() > 1 else batch_phases
        pix_f, sig_s_f, sig_d_f = _fused.render_rays(static_model_fine, temp_model_fine, batch_origins, batch_directions, phase_per_ray,
                                                     batch_initial_intensities, z_all, dists_f, act=output_activation)
    return pix_c, sig_s_c, sig_d_c, dists_c, pix_f, sig_s_f, sig_d_f, dists_f


def sample_pdf(bins, weights, N_samples, device, u=None):
    """Inverse-transform sampling of fine depths (model_helpers.py:162-187); ``u`` injects the draw."""
    w = weights + 1e-5
    cdf = torch.cumsum(w / w.sum(dim=-1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if u is None:
        u = torch.rand(list(cdf.shape[:-1]) + [N_samples])
    u = u.to(w)
    hit = torch.searchsorted(cdf, u, right=True)
    lo = (hit - 1).clamp(min=0)
    hi = hit.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = cdf.gather(1, lo), cdf.gather(1, hi)
    b_lo, b_hi = bins.gather(1, lo), bins.gather(1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return b_lo + (u - c_lo) / span * (b_hi - b_lo)
