"""The composite training step of run_composite.py:227-312 around the fused ray path, with ray-sharded
data parallelism: one process per GPU, every rank draws the SAME global batch (same seed) and
renders its contiguous slice; one all-reduce(SUM) of the flat gradient per step, then the identical
Adam update everywhere.

Loss normalisation under sharding: terms that are means over rays use local sums
divided by the GLOBAL ray count; ``static_l1``/``static_l2`` are sums over the batch and stay sums.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from ..schedules import linear_param_decay
from . import model_helpers as MH


@dataclass
class TrainConfig:
    """The hot-path keys of train/composite.txt (defaults = that file)."""
    depth_samples_per_ray_coarse: int = 500
    depth_samples_per_ray_fine: int = 0            # composite.txt:26; > 0 needs the fine model pair (run_composite.py:194-207)
    fine_depth_gradients: Optional[bool] = None    # as the reference, the fine losses also differentiate through the sampled depths (None = True)
    img_sample_size: int = 1024
    batch_size: int = 32768
    lr: float = 1e-3
    lr_end_factor: float = 0.01
    lr_decay_steps: int = 150000
    var_sample_perc: float = 50
    var_sample_thre: float = 3
    entro_mask_thre: float = 1e-4
    entro_use_weighting: bool = True
    entro_weighted_thresh: float = 0.03
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
    weighted_loss_max: float = 1
    occl_reg_perc: float = 0.2
    skewness_val: float = 1.0
    favor_s_opt: Optional[str] = None
    output_activation: str = "softplus"
    static_pos_enc: str = "free_windowed"
    temp_pos_enc: str = "free_windowed"
    static_pos_enc_window_decay_steps: int = 150000
    temp_pos_enc_window_decay_steps: int = 150000


class _MaxReducer:
    """reduce_max of the fine pass under ray sharding: called on a tensor it takes the maximum over the ranks in place;
    ``sum`` adds over the ranks (the backward of that maximum, model_helpers._BatchMax)."""

    def __call__(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)

    @staticmethod
    def sum(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


class _FromRank0(torch.autograd.Function):
    """Ray 0 of the GLOBAL batch sits on rank 0: every rank renders its fine pass with that ray's depths (model_helpers.py:150).
    Forward: broadcast; backward: the ranks' gradients are summed and handed to rank 0's ray."""

    @staticmethod
    def forward(ctx, z0, rank):
        ctx.rank = rank
        out = z0.detach().clone()
        dist.broadcast(out, src=0)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.detach().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return (g if ctx.rank == 0 else torch.zeros_like(g)), None


def _scoped(method):
    """Run a trainer method inside the trainer's own planner scope (fused.PlanScope): its options apply to the library calls the
    method makes, and the planner's decisions land in ``trainer.plan_scope`` -- not in (or from) another trainer's."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *a, **kw):
        with self.plan_scope:
            return method(self, *a, **kw)
    return wrapper


class CompositeTrainer:
    def __init__(self, cfg: TrainConfig, static_model, temp_model, data, device, rank: int = 0, world: int = 1,
                 seed: int = 0, fused_adam: Optional[bool] = None, fused_loss: Optional[bool] = None,
                 static_model_fine=None, temp_model_fine=None, plan_opts: Optional[dict] = None):
        """``plan_opts``: this trainer's planner options (keys of ``_capi.NcaPlanOpts``: stage_fp8, stage_fp8_min_tiles,
        resident_min_tiles, wgrad_rebuild_weight_pct, overlap_cus) -- they apply to this trainer's library calls only; ``self.plan()`` is
        what the planner decided for them.

        Every device operation of a step goes through the HIP library: the renderer (``_render``), the fine-pass sampler
        (``_fine_depths`` / ``_fine_depths_autograd``), the batch preparation (``_prepare``) and the evaluation's loss terms
        (``_eval_terms``) are methods so that the CPU tests of the data-parallel bookkeeping can put the oracle behind them
        (tests/injected_trainer.py); nothing in this module has a torch implementation of the path."""
        from ..fused import PlanScope
        self.plan_scope = PlanScope(**(plan_opts or {}))
        self._bad_ids = None               # device i32[1]: ray ids outside the table that nca_prepare_batch clamped (checked at the syncing calls)
        self.cfg, self.s, self.t, self.data, self.device = cfg, static_model, temp_model, data, device
        self.s_fine, self.t_fine = static_model_fine, temp_model_fine
        self.n_fine = int(cfg.depth_samples_per_ray_fine)
        if self.n_fine > 0 and (static_model_fine is None or temp_model_fine is None):
            raise ValueError("depth_samples_per_ray_fine > 0 needs static_model_fine and temp_model_fine (run_composite.py:194-205)")
        self.rank, self.world, self.seed = rank, world, seed
        on_cuda = device.type == "cuda" if isinstance(device, torch.device) else str(device).startswith("cuda")
        if on_cuda:                        # (allocated here, not at first use: the first use may be inside a graph capture)
            self._bad_ids = torch.zeros(1, dtype=torch.int32, device=device)
        if fused_loss is None:             # default on the GPU: the autograd-free step with the HIP loss kernel (step_fused);
            fused_loss = on_cuda           # fused_loss=False: the reference's loss FUNCTIONS (model_helpers.compute_losses, HIP behind them) under autograd
        self.fused_loss = bool(fused_loss)
        import os
        # A ray id outside the table raises in the reference (NumPy's IndexError, run_composite.py:262).  The library clamps and counts; the
        # host-launched steps read the counter EVERY step (one device read beside ~20 launches) and raise; NERFCA_STRICT=0 defers the check to
        # the syncing calls (early_stop / evaluate), which is also what the graph-replayed step does (nothing may synchronise inside it).
        self.strict_ids = os.environ.get("NERFCA_STRICT", "1") != "0"
        self.sampler = None                # fused.BatchSampler: the device-side Philox streams of (seed, iteration) -- created at first use on the GPU
        self.stop_flag = None              # device bool: the reference's early-stop predicate of the last step (see early_stop)
        self.always_allreduce = False      # all-reduce even with one rank (exercises the collective path)
        self._dev_gen = None
        self.params = list(temp_model.parameters()) + list(static_model.parameters())   # run_composite.py:192
        if self.n_fine > 0:
            self.params += list(temp_model_fine.parameters()) + list(static_model_fine.parameters())             # :207
        kw = {}
        if fused_adam is None:
            fused_adam = device.type == "cuda" if isinstance(device, torch.device) else str(device).startswith("cuda")
        if fused_adam and all(p.is_contiguous() for p in self.params):      # (nets padded to a kernel width own strided views: torch's fused Adam wants one layout)
            kw["fused"] = True
        self.opt = torch.optim.Adam([{"params": self.params, "lr": cfg.lr}], lr=cfg.lr, **kw)
        self.sched = torch.optim.lr_scheduler.LinearLR(self.opt, start_factor=1, end_factor=cfg.lr_end_factor, total_iters=cfg.lr_decay_steps)
        self.depth = MH_depth(data.geo, cfg.depth_samples_per_ray_coarse, device)
        self.I0 = torch.full((cfg.img_sample_size,), data.geo["max_pixel_value"], dtype=torch.float32, device=device)
        self.n_var = int((cfg.var_sample_perc / 100.0) * cfg.img_sample_size) if cfg.var_sample_perc > 0 else 0

    # -- the device operations of a step (HIP library; see the class docstring) ------------------
    def _render(self, static_model, temp_model, o, d, phases, I0, z, dists, act):
        """(pix, sigma_s, sigma_d) of one ray batch: the fused render (obtain_train_predictions_iter's coarse or fine half)."""
        return MH._fused.render_rays(static_model, temp_model, o, d, phases, I0, z, dists, act=act)

    def _fine_depths(self, sig_s, sig_d, z, u, reduce_max):
        """z_all[R, S + n_fine] of the hierarchical pass, the depths constants of the step (fine_depth_gradients=False)."""
        return MH._fused.fine_depths(sig_s, sig_d, z, u, reduce_max=reduce_max)

    def _fine_depths_autograd(self, sig_s, sig_d, z, u, reduce_max):
        """The same with the sampled depths in the autograd graph, as the reference has them (model_helpers.py:135-146): HIP sampler,
        HIP backward into the coarse densities."""
        return MH._fused.fine_depths_autograd(sig_s, sig_d, z, u, reduce_max=reduce_max)

    def _pixel_loss(self, pix, gt, w):
        """loss_fn(pred, gt, weights).mean() of run_composite.py:287 (weighted_MSELoss: a HIP kernel under autograd)."""
        return MH.weighted_MSELoss()(pix, gt, w).mean()

    def _loss_terms(self, sig_s, sig_d, dists, w):
        """compute_losses' 11-tuple (model_helpers.py:250-262) under autograd: the HIP loss kernel behind the reference's function."""
        return MH.compute_losses(sig_s, sig_d, dists, w, self.cfg)

    # -- per-step host work (identical on every rank) ------------------------------------------
    def update_windows(self, n_iter: int) -> None:
        c = self.cfg
        nets = [(self.s, c.static_pos_enc, c.static_pos_enc_window_decay_steps), (self.t, c.temp_pos_enc, c.temp_pos_enc_window_decay_steps)]
        if self.n_fine > 0:                # run_composite.py:241-247
            nets += [(self.s_fine, c.static_pos_enc, c.static_pos_enc_window_decay_steps), (self.t_fine, c.temp_pos_enc, c.temp_pos_enc_window_decay_steps)]
        for m, enc, steps in nets:
            if enc == "nerfies_windowed":
                m.update_windowed_alpha(n_iter, steps)
            elif enc == "free_windowed":
                m.update_freq_mask_alpha(n_iter, steps)

    def draw_ray_ids(self, n_iter: int) -> np.ndarray:
        """Importance sampling of run_composite.py:250-260 with a per-step seeded generator."""
        rng = np.random.default_rng([self.seed, n_iter])
        c, d = self.cfg, self.data
        if c.var_sample_perc > 0 and len(d.var_ray_ids) > 0:
            ids = np.concatenate((rng.choice(d.non_var_ray_ids, size=c.img_sample_size - self.n_var),
                                  rng.choice(d.var_ray_ids, size=self.n_var)))
            rng.shuffle(ids)
            return ids
        return rng.integers(low=0, high=d.rays_train.shape[0], size=c.img_sample_size)

    def _sampler(self):
        """The device-side sampler of this trainer (fused.BatchSampler): ids and jitter are functions of (seed, iteration, slot)."""
        if self.sampler is None:
            from ..fused import BatchSampler
            c, d = self.cfg, self.data
            use_var = c.var_sample_perc > 0 and len(d.var_ray_ids) > 0
            self.sampler = BatchSampler(self.seed, c.img_sample_size, self.n_var if use_var else 0, d.var_ray_ids if use_var else None,
                                        d.non_var_ray_ids if use_var else None, int(d.rays_train.shape[0]), self.device)
        return self.sampler

    def draw_ray_ids_device(self, n_iter: int) -> torch.Tensor:
        """The importance sampling of run_composite.py:250-260 drawn ON the GPU by the library (nca_draw_ray_ids): the ids of the GLOBAL
        batch of iteration ``n_iter`` -- img_sample_size - n_var i.i.d. uniform draws from the non-variance rays and n_var from the variance
        rays in a uniformly random arrangement (csrc/nca_rng.hpp) -- identical on every rank, no host RNG / shuffle / H2D copy.  The
        graph-replayed step draws the same ids inside its first kernel (nca_begin_step)."""
        return self._sampler().ray_ids(n_iter)

    def _draw_ids(self, n_iter: int):
        return self.draw_ray_ids_device(n_iter)

    def draw_jitter(self, n_iter: int) -> torch.Tensor:
        """randomize_depth's uniform draw t_rand f32[S] of iteration ``n_iter`` (model_helpers.py:8): on the GPU the library's jitter stream
        (a device tensor; the graph step draws the same values in nca_begin_step), on the CPU (the injected tests) a seeded torch draw."""
        on_cuda = torch.device(self.device).type == "cuda"
        if on_cuda:
            return self._sampler().uniform(n_iter, int(self.depth.shape[0]))
        g = torch.Generator().manual_seed(self.seed * 1000003 + n_iter)
        return torch.rand(self.depth.shape, generator=g)

    def _draws_injected(self) -> bool:
        """A test (or a caller replaying the reference's own draws) replaced ``draw_ray_ids_device`` / ``draw_jitter`` / ``_draw_ids``: the
        graph step then takes the ids and the jitter from those methods (copied per step) instead of drawing them on the device."""
        cls = CompositeTrainer
        for name in ("draw_ray_ids_device", "draw_jitter", "_draw_ids", "loss_weights", "update_windows"):
            if name in self.__dict__ or getattr(type(self), name) is not getattr(cls, name):
                return True
        return False

    def draw_fine_u(self, n_iter: int) -> torch.Tensor:
        """sample_pdf's uniform draw for the GLOBAL batch (model_helpers.py:170), from a per-step seeded CPU generator so
        that every rank draws the same table and takes its rows."""
        g = torch.Generator().manual_seed(self.seed * 1000003 + n_iter + 500009)
        return torch.rand((self.cfg.img_sample_size, self.n_fine), generator=g)

    def loss_weights(self, n_iter: int):
        c = self.cfg
        return (linear_param_decay(n_iter, c.favor_s_weight_start, c.favor_s_weight_end, c.hyperparam_decay_steps, c.favor_s_weight_delay_steps),
                linear_param_decay(n_iter, c.dynamic_entro_weight_start, c.dynamic_entro_weight_end, c.hyperparam_decay_steps),
                linear_param_decay(n_iter, c.occl_weight_start, c.occl_weight_end, c.hyperparam_decay_steps, c.favor_s_weight_delay_steps),
                linear_param_decay(n_iter, c.l1_weight_start, c.l1_weight_end, c.hyperparam_decay_steps))

    # -- one optimisation step -----------------------------------------------------------------
    def local_loss(self, n_iter: int, ids: np.ndarray, t_rand: torch.Tensor):
        """Loss contribution of this rank's slice of the global batch (sums to the global loss)."""
        c = self.cfg
        R = len(ids)
        lo, hi = (R * self.rank) // self.world, (R * (self.rank + 1)) // self.world
        my = ids[lo:hi] if torch.is_tensor(ids) else torch.as_tensor(ids[lo:hi], device=self.device)
        o, d, gt, w, phases, z, dists = self._prepare(my, t_rand)        # (run_composite.py:262-273, model_helpers.py:3-12, 73-74)
        pix, sig_s, sig_d = self._render(self.s, self.t, o, d, phases, self.I0[: hi - lo], z, dists, c.output_activation)
        share = (hi - lo) / R                                            # local mean -> share of the global mean
        pixel = self._pixel_loss(pix, gt, w) * share
        terms = self._loss_terms(sig_s, sig_d, dists, w)
        fav_w, ent_w, occ_w, l1_w = self.loss_weights(n_iter)
        favor, d_ent, occl, l1, l2 = terms[3], terms[6], terms[8], terms[9], terms[10]
        loss = pixel + fav_w * favor * share + ent_w * d_ent * share + occ_w * occl * share + l1_w * l2 + l1_w * l1
        if self.n_fine > 0:
            # hierarchical pass (model_helpers.py:131-158, run_composite.py:294-301).  The weights are normalised by the
            # maximum over the GLOBAL batch (:139) -> MAX all-reduce inside the sampler; the fine rendering takes its
            # interval lengths from ray 0 of the GLOBAL batch (:150) -> broadcast from rank 0.  The sampled depths stay in
            # the autograd graph as in the reference, or are constants of the step (fine_depth_gradients=False); see
            # model_helpers.obtain_train_predictions_iter.
            sharded = self.world > 1
            red = _MaxReducer() if sharded else None
            u = self.draw_fine_u(n_iter)[lo:hi].to(self.device)
            if c.fine_depth_gradients is None or c.fine_depth_gradients:
                # as the reference: the sampled depths stay in the autograd graph (model_helpers.py:135-146) and the fused
                # render returns d loss / d depth, so the fine losses also reach the COARSE nets; the batch-wide maximum and
                # ray 0's depths cross the ranks in both directions
                z_all = self._fine_depths_autograd(sig_s, sig_d, z, u, red)
                z0 = _FromRank0.apply(z_all[0, :], self.rank) if sharded else z_all[0, :]
            else:
                z_all = self._fine_depths(sig_s.detach(), sig_d.detach(), z, u, red)
                z0 = z_all[0, :].clone()
                if sharded:
                    dist.broadcast(z0, src=0)
            dists_f = MH._interval_lengths(z0, d)
            pix_f, sig_sf, sig_df = self._render(self.s_fine, self.t_fine, o, d, phases, self.I0[: hi - lo], z_all, dists_f, c.output_activation)
            pixel_f = self._pixel_loss(pix_f, gt, torch.ones_like(w)) * share       # weighted_pixs_ones (:297)
            tf = self._loss_terms(sig_sf, sig_df, dists_f, w)
            loss = loss + pixel_f + fav_w * tf[3] * share + ent_w * tf[6] * share + occ_w * tf[8] * share + l1_w * tf[10] + l1_w * tf[9]
            self.last_fine_terms_autograd = tf      # (the early stop reads the fine pass's entropy / favor terms, run_composite.py:298-310)
        return loss, pixel, terms

    def plan(self) -> dict:
        """What the planner decided in THIS trainer's last forward / backward (NcaPlan, include/nerfca_hip.h)."""
        return self.plan_scope.decided()

    def check_ray_ids(self) -> None:
        """One device read: raise if a ray id of any step so far fell outside the ray table (the library clamped it)."""
        if self._bad_ids is not None:
            n = int(self._bad_ids.item())
            if n:
                from .. import _capi
                raise _capi.NcaError(f"{n} ray ids outside the ray table of {self.data.rays_train.shape[0]} rows reached nca_prepare_batch (clamped, results invalid)")

    @_scoped
    def step(self, n_iter: int):
        if self.fused_loss:
            return self.step_fused(n_iter)
        self.update_windows(n_iter)
        loss, pixel, terms = self.local_loss(n_iter, self._draw_ids(n_iter), self.draw_jitter(n_iter))
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        # the reference overwrites dynamic_entropy_loss / favor_s_loss with the fine pass's values before its check (run_composite.py:298-310)
        tt = self.last_fine_terms_autograd if self.n_fine > 0 else terms
        share = 1.0 / self.world if self.world > 1 else 1.0             # (local means: equal slices)
        pair = self._stop_pair(n_iter, d_entropy=tt[6] * share, favor=tt[3] * share)
        if self.world > 1 or self.always_allreduce:
            pair = self.allreduce_grads(pair)
        self.opt.step()
        self.sched.step()
        self._note_early_stop(n_iter, pair)
        return loss.detach(), pixel.detach(), terms

    def _prepare(self, my: torch.Tensor, t_rand: torch.Tensor):
        """This rank's rays of the step and the jittered depths: ``(o, d, gt, w, phases, z, dists)`` as run_composite.py:262-273 and
        model_helpers.py:3-12, 73-74 build them -- ONE library launch (fused.prepare_batch: bit-identical to the reference's torch
        operations, tests/test_hip_parity.py) on the ray table as the loader leaves it (data_helpers.py:141-165: f64 [N,4,3] and one
        int64 phase per ray, resident on the GPU)."""
        from ..fused import prepare_batch
        rt, pt = self.data.rays_train, self.data.phases_train
        if not (rt.is_cuda and rt.dtype == torch.float64 and rt.dim() == 3 and rt.is_contiguous() and pt.is_cuda and pt.dim() == 1):
            from .. import _capi
            raise _capi.NcaError(f"CompositeTrainer needs the ray table on the GPU as prepare_data_for_loader_tigre builds it (f64 [N,4,3], contiguous; one phase id "
                                 f"per ray): got {rt.dtype} {tuple(rt.shape)} on {rt.device}, phases {pt.dtype} {tuple(pt.shape)}.  There is no CPU or torch path.")
        pt = self._phases_i64()
        out = prepare_batch(my.to(torch.int64).contiguous(), rt, pt, self.depth, t_rand, bad_ids=self._bad_ids)
        if self.strict_ids and not getattr(self, "_in_graph_setup", False) and not torch.cuda.is_current_stream_capturing():
            self.check_ray_ids()            # (host-launched steps: an id outside the table raises at once, as NumPy's IndexError does in the reference)
        return out

    def fused_gradients(self, n_iter: int):
        """What ``loss.backward()`` yields in the reference (run_composite.py:283-306) for this rank's slice of step ``n_iter``'s
        batch, without an autograd graph and without touching the optimiser: fused forward -> fused loss kernel (values + d loss /
        d(pix, sigma)) -> fused backward.  Returns ``(terms f64[13], flat gradient of the static net, of the dynamic net)``."""
        from ..fused import _RayBatch, fused_losses, render_backward_raw, render_forward_raw
        c = self.cfg
        self.update_windows(n_iter)
        ids = self.draw_ray_ids_device(n_iter)
        R = ids.shape[0]
        lo, hi = (R * self.rank) // self.world, (R * (self.rank + 1)) // self.world
        my = ids[lo:hi]
        o, d, gt, w, phases, z, dists = self._prepare(my, self.draw_jitter(n_iter))
        return self._micro_batched(o, d, gt, w, phases, z, dists, R, self.loss_weights(n_iter), None)

    def _micro_batched(self, o, d, gt, w, phases, z, dists, R, weights, weights_dev, flat_out=None):
        """Fused forward -> loss kernel -> fused backward of this rank's rays, over as many ray micro-batches as keep the forward
        store under the limit; ``(terms, grads_s, grads_d)`` summed.  Also the body of the captured graph step (the number of
        micro-batches is fixed at capture, like everything else about the step's structure).  ``flat_out``: the step's ONE flat f32 buffer
        ``[dynamic net's gradient | static net's | the 13 terms as f32]`` -- the gradients and the terms are then written straight into it
        (what the all-reduce and the library's Adam read: no concatenation, no conversion kernels)."""
        from ..fused import _RayBatch, fused_losses, render_backward_raw, render_forward_raw
        c = self.cfg
        bs, bd = self.s._binding, self.t._binding
        # Rays are independent given the weights and every loss term is a sum over rays (times the GLOBAL 1/R), so the
        # step may run over ray micro-batches and add up: that keeps the forward store (the tensors autograd would keep)
        # under fused.STORE_FORWARD_LIMIT_BYTES at any batch size instead of falling back to the recompute backward.
        from .. import fused as FU
        n_loc = o.shape[0]
        whole = _RayBatch(o, d, phases, self.I0[:n_loc], z, dists, c.output_activation, False, 1e-2)
        need = FU.forward_store_bytes(whole, bs, bd)
        limit = FU.store_limit_bytes(self.device)
        micro = n_loc if need <= limit or need == 0 or limit <= 0 else max(1, int(n_loc * (limit / need)))
        while micro < n_loc and micro > 1:          # (a store's size is not linear in the rays: slack tile slots, alignment -- make sure the micro-batch's really fits)
            probe = _RayBatch(o[:micro], d[:micro], phases[:micro], self.I0[:micro], z, dists, c.output_activation, False, 1e-2)
            if FU.forward_store_bytes(probe, bs, bd) <= limit:
                break
            micro = max(1, int(micro * 0.9))
        self.micro_batches = (n_loc + micro - 1) // micro
        terms = grads_s = grads_d = None
        nd, ns = bd.flat.numel(), bs.flat.numel()
        if flat_out is not None and micro == n_loc:
            # the whole rank's batch in one pass: the forward leaves pix to the loss kernel (its per-tile ray sums), the loss kernel writes the
            # terms behind the gradients, the backward writes the gradients in place
            sums, sig_s, sig_d, keep = render_forward_raw(whole, bs, bd, for_backward=True, want_pix=False)
            terms, g_pix, g_s, g_d = fused_losses(sums, gt, w, sig_s, sig_d, dists, c, weights, inv_R=1.0 / R, weights_dev=weights_dev,
                                                  terms_f32=flat_out[nd + ns:nd + ns + 13])
            grads_s, grads_d = render_backward_raw(whole, bs, bd, keep, g_pix, g_s, g_d, out_s=flat_out[nd:nd + ns], out_d=flat_out[:nd])
            return terms, grads_s, grads_d
        for m0 in range(0, n_loc, micro):
            m1 = min(n_loc, m0 + micro)
            batch = whole if micro == n_loc else _RayBatch(o[m0:m1], d[m0:m1], phases[m0:m1], self.I0[: m1 - m0], z, dists, c.output_activation, False, 1e-2)
            pix, sig_s, sig_d, keep = render_forward_raw(batch, bs, bd, for_backward=True)
            t_m, g_pix, g_s, g_d = fused_losses(pix, gt[m0:m1], w[m0:m1], sig_s, sig_d, dists, c, weights, inv_R=1.0 / R, weights_dev=weights_dev)
            gs_m, gd_m = render_backward_raw(batch, bs, bd, keep, g_pix, g_s, g_d)
            del keep
            if terms is None:
                terms, grads_s, grads_d = t_m, gs_m, gd_m
            else:       # sums, except the two logged maxima
                mx = torch.maximum(terms[3:5], t_m[3:5])
                terms = terms + t_m
                terms[3:5] = mx
                grads_s += gs_m
                grads_d += gd_m
        if flat_out is not None:
            flat_out[:nd].copy_(grads_d)
            flat_out[nd:nd + ns].copy_(grads_s)
            flat_out[nd + ns:nd + ns + 13].copy_(terms)
        return terms, grads_s, grads_d

    @_scoped
    def step_fused(self, n_iter: int):
        """Same step without an autograd graph: fused forward -> fused loss kernel (values + d loss/d(pix,
        sigma)) -> fused backward -> (all-reduce) -> Adam.  Returns (loss, pixel, terms f64[13]) on device;
        the entries of ``terms`` are this rank's share of the global value (they sum over ranks).  With a fine model
        pair (``depth_samples_per_ray_fine > 0``) see ``_step_fused_fine``."""
        if self.n_fine > 0:
            return self._step_fused_fine(n_iter)
        bs, bd = self.s._binding, self.t._binding
        terms, grads_s, grads_d = self.fused_gradients(n_iter)
        pair = self._stop_pair(n_iter, terms)
        if self.world > 1 or self.always_allreduce:
            # ONE collective per step: the early-stop predicate's two scalars ride behind the gradients
            nd, ns = grads_d.numel(), grads_s.numel()
            flat = torch.cat([grads_d, grads_s] + ([pair] if pair is not None else []))
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            grads_d, grads_s = flat[:nd], flat[nd:nd + ns]
            if pair is not None:
                pair = flat[nd + ns:]
        for p, g in zip(self.t.parameters(), bd.split_grads(grads_d)):
            p.grad = g
        for p, g in zip(self.s.parameters(), bs.split_grads(grads_s)):
            p.grad = g
        self.opt.step()
        self.sched.step()
        self._note_early_stop(n_iter, pair)
        return terms[0], terms[1], terms

    def _step_fused_fine(self, n_iter: int):
        """The hierarchical step of run_composite.py:283-308 without an autograd graph: coarse forward -> coarse loss kernel ->
        HIP sampler (batch-wide maximum all-reduced under sharding) -> fine forward on the merged depths with ray 0's interval
        lengths -> loss kernel with unit pixel weights and weighted regularisers (:296-299) -> fine backward (+ d loss / d depth)
        -> sampler backward into the coarse densities (the reference does not detach the sampled depths, model_helpers.py:135-146;
        ``fine_depth_gradients=False`` skips this) -> coarse backward -> one all-reduce -> Adam over the four nets.
        Returns (loss, coarse pixel loss, coarse terms + fine terms); ``self.last_fine_terms`` keeps the fine pass's own."""
        c, dev = self.cfg, self.device
        self.update_windows(n_iter)
        ids = self.draw_ray_ids_device(n_iter)
        R = ids.shape[0]
        lo, hi = (R * self.rank) // self.world, (R * (self.rank + 1)) // self.world
        z = MH.randomize_depth(self.depth, dev, self.draw_jitter(n_iter))
        u = self.draw_fine_u(n_iter)[lo:hi].to(dev)
        terms, terms_f, order = self._fine_device_work(ids[lo:hi], R, z, u, self.loss_weights(n_iter))
        sharded = self.world > 1
        pair = self._stop_pair(n_iter, terms_f)
        if sharded or self.always_allreduce:
            flat = torch.cat([g for _, _, g in order] + ([pair] if pair is not None else []))
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            off = 0
            for i, (m, b, g) in enumerate(order):
                order[i] = (m, b, flat[off:off + g.numel()])
                off += g.numel()
            if pair is not None:
                pair = flat[off:]
        for m, b, g in order:
            for p, gr in zip(m.parameters(), b.split_grads(g)):
                p.grad = gr
        self.opt.step()
        self.sched.step()
        self.last_fine_terms = terms_f
        total = terms.clone()
        total[0] = terms[0] + terms_f[0]                                   # loss += the fine pass's assembled loss (:301)
        self._note_early_stop(n_iter, pair)
        return total[0], terms[1], total

    def _fine_device_work(self, my, R, z, u, weights, weights_dev=None):
        """The device side of the hierarchical step for this rank's ray ids ``my`` (of a global batch of ``R``): everything between
        the ray gather and the gradient all-reduce.  ``weights`` are the four loss weights as host floats; with ``weights_dev``
        (f64[4] on the device) nothing of the step depends on host scalars and the whole call can be captured in a HIP graph.
        Returns (coarse terms, fine terms, [(model, binding, flat gradient)] in ``self.params`` order)."""
        from .. import fused as FU
        from ..fused import _RayBatch, fused_losses, render_backward_raw, render_forward_raw
        c, dev = self.cfg, self.device
        n_loc = my.shape[0]
        rays = self.data.rays_train.index_select(0, my)
        phases = self.data.phases_train.index_select(0, my)
        o, d, gt, w = rays[:, 0, :], rays[:, 1, :], rays[:, 2, 0], rays[:, 3, 0]
        dists = MH._interval_lengths(z, d)
        bs, bd, bsf, bdf = self.s._binding, self.t._binding, self.s_fine._binding, self.t_fine._binding
        sharded = self.world > 1
        red = _MaxReducer() if sharded else None
        I0 = self.I0[:n_loc]
        # coarse pass (the whole local batch: the sampler normalises by the batch-wide maximum)
        batch = _RayBatch(o, d, phases, I0, z, dists, c.output_activation, False, 1e-2)
        pix, sig_s, sig_d, keep = render_forward_raw(batch, bs, bd, for_backward=True)
        terms, g_pix, g_s, g_d = fused_losses(pix, gt, w, sig_s, sig_d, dists, c, weights, inv_R=1.0 / R, weights_dev=weights_dev)
        # fine depths
        z_all, saved = FU.fine_depths_forward(sig_s, sig_d, z, u, red)
        z0 = z_all[0, :].clone()
        if sharded:
            dist.broadcast(z0, src=0)                                     # ray 0 of the GLOBAL batch (model_helpers.py:150)
        dists_f = MH._interval_lengths(z0, d)
        batch_f = _RayBatch(o, d, phases, I0, z_all, dists_f, c.output_activation, False, 1e-2)
        pix_f, sig_sf, sig_df, keep_f = render_forward_raw(batch_f, bsf, bdf, for_backward=True)
        depth_grads = c.fine_depth_gradients is None or c.fine_depth_gradients
        # (with depth gradients the loss kernel also returns d loss / d (ray 0's interval lengths): through the ray sums
        # pix = I0 - sum sigma dists and through the regularisers, all of which contain sigma * dists)
        lf = fused_losses(pix_f, gt, w, sig_sf, sig_df, dists_f, c, weights, inv_R=1.0 / R, unit_mse=True, weights_dev=weights_dev,
                          want_dists_grad=depth_grads)
        terms_f, g_pix_f, g_sf, g_df = lf[:4]
        res = render_backward_raw(batch_f, bsf, bdf, keep_f, g_pix_f, g_sf, g_df, want_depth_grad=depth_grads)
        grads_sf, grads_df = res[0], res[1]
        del keep_f
        if depth_grads:
            g_zall = res[2]
            g_dists = lf[4]
            g_z0 = torch.zeros_like(z0, dtype=torch.float64)               # dists = cat(z0[1:] - z0[:-1], [1e-10])
            g_z0[1:] += g_dists[:-1]
            g_z0[:-1] -= g_dists[:-1]
            if sharded:
                dist.all_reduce(g_z0, op=dist.ReduceOp.SUM)
            if self.rank == 0:
                g_zall[0, :] += g_z0.to(g_zall.dtype)
            g_tot = FU.fine_depths_backward(saved, g_zall, _MaxReducer.sum if sharded else None)
            g_s = g_s + g_tot
            g_d = g_d + g_tot
        grads_s, grads_d = render_backward_raw(batch, bs, bd, keep, g_pix, g_s, g_d)
        del keep
        order = [(self.t, bd, grads_d), (self.s, bs, grads_s), (self.t_fine, bdf, grads_df), (self.s_fine, bsf, grads_sf)]   # self.params order
        return terms, terms_f, order

    # -- early stop (run_composite.py:310-312) ---------------------------------------------------
    def _stop_pair(self, n_iter: int, terms=None, d_entropy=None, favor=None):
        """This rank's share of [dynamic entropy, favor] (f32[2] on the device) for the early-stop predicate, or None while the
        frequency windows are still opening (the predicate is not evaluated then).  Under ray sharding the pair is appended to
        the flat gradient buffer and summed by the step's ONE all-reduce."""
        if n_iter < self.cfg.static_pos_enc_window_decay_steps:
            return None
        if terms is not None:
            d_entropy, favor = terms[8], terms[5]           # (nca_loss_fwd_bwd order: this rank's share of the global means)
        return torch.stack([d_entropy.detach().reshape(()), favor.detach().reshape(())]).to(torch.float32)

    def _note_early_stop(self, n_iter: int, pair) -> None:
        """The reference breaks its loop when ``dynamic_entropy_loss < 1e-15 or favor_s_loss < 1e-15`` once the frequency windows
        are fully open (the fine pass's terms when there is one, run_composite.py:298-312).  ``pair`` = the two values summed
        over the ranks (``_stop_pair``); evaluated on the device into ``self.stop_flag`` without a host sync."""
        self.stop_flag = None if pair is None else (pair < 1e-15).any()

    def global_terms(self, terms: torch.Tensor) -> torch.Tensor:
        """What a logger wants under ray sharding: ``terms`` as ``step_fused`` / ``step_graph`` return them hold this rank's SHARE of
        every global value (loss, pixel loss and the regularisers are sums over the ranks) and this rank's own maxima
        (``sigma_s_max``, ``sigma_d_max``).  Returns the global vector -- one all-reduce(SUM) and one all-reduce(MAX) of 13 doubles;
        call it at the logging cadence, not every step (``run_composite.py:314-336`` logs every ``log_every`` steps).  With one rank
        it is the identity."""
        if self.world <= 1:
            return terms
        from .. import _capi
        i0, i1 = _capi.TERM_NAMES.index("sigma_s_max"), _capi.TERM_NAMES.index("sigma_d_max")
        tot = terms.detach().clone()
        mx = tot[[i0, i1]].clone()
        tot[[i0, i1]] = 0
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        tot[[i0, i1]] = mx
        return tot

    def early_stop(self) -> bool:
        """Host-side read of the flag (one device sync): call it at the logging cadence, not every step.  (Also where a ray id
        outside the table, clamped by the library, surfaces: ``check_ray_ids``.)"""
        self.check_ray_ids()
        return bool(self.stop_flag) if self.stop_flag is not None else False

    # -- the same step as a replayed HIP graph ---------------------------------------------------
    def _device_schedules(self):
        """The step's schedules as the library computes them on the device (nca_begin_step), or None where one of them has no device form:
        the FreeNeRF windows of every net (free_windowed; encodings without a schedule keep their constant vector) and the four
        linear_param_decay loss weights.  Returns (NcaSchedules, window vectors per net, weights f64[4])."""
        from .. import _capi
        c, dev = self.cfg, self.device
        nets = [(self.s, c.static_pos_enc, c.static_pos_enc_window_decay_steps), (self.t, c.temp_pos_enc, c.temp_pos_enc_window_decay_steps)]
        sch = _capi.NcaSchedules()
        wins, slot = [], {}
        for m, enc, steps in nets:
            L = int(m.pos_enc_basis)
            if enc == "nerfies_windowed":
                return None                # (cosine easing in torch's f32 arithmetic on the host: no bit-identical device form)
            if m._binding.net.enc_mode != _capi.ENC_BANDS or L <= 0:
                wins.append(None)
                continue
            if enc == "free_windowed":
                key = (L, int(steps), int(m.pos_enc_window_start))
                if key not in slot:        # identical schedules share ONE vector: the forward then stores the encoded input once
                    if len(slot) >= 4 or L > 64:
                        return None
                    vec = torch.zeros(L, dtype=torch.float32, device=dev)
                    k = len(slot)
                    sch.window[k].kind, sch.window[k].L, sch.window[k].window_start = _capi.WINDOW_FREE, L, int(m.pos_enc_window_start)
                    sch.window[k].decay_steps, sch.window[k].out = int(steps), vec.data_ptr()
                    slot[key] = vec
                wins.append(slot[key])
            else:                          # plain bands: all ones, no schedule
                wins.append(m._band_window().detach().to(device=dev, dtype=torch.float32).contiguous().clone())
        sch.n_windows = len(slot)
        weights = torch.zeros(4, dtype=torch.float64, device=dev)
        for k, (w0, w1, delay) in enumerate(((c.favor_s_weight_start, c.favor_s_weight_end, c.favor_s_weight_delay_steps),
                                             (c.dynamic_entro_weight_start, c.dynamic_entro_weight_end, 0),
                                             (c.occl_weight_start, c.occl_weight_end, c.favor_s_weight_delay_steps),
                                             (c.l1_weight_start, c.l1_weight_end, 0))):
            sch.weight[k].start, sch.weight[k].end, sch.weight[k].steps, sch.weight[k].delay = float(w0), float(w1), int(c.hyperparam_decay_steps), int(delay)
        sch.weights_out = weights.data_ptr()
        return sch, wins, weights

    def _graph_setup(self) -> None:
        """Capture gather -> jitter -> fused forward -> loss kernel -> fused backward (-> all-reduce) -> Adam+LinearLR once.

        DEVICE mode (the default): everything that changes from step to step is MADE on the device by the step's first kernel
        (nca_begin_step) from the iteration counter the library's Adam increments -- ray ids, depth jitter, band windows, loss weights --
        so a replay needs no host work at all.  RECORD mode (draws injected by a test or a caller, the hierarchical pass, schedules without
        a device form): the ray ids and one small host-pinned record (depth jitter, band windows, loss weights) are copied per step."""
        from ..fused import FusedAdam, begin_step
        c, dev = self.cfg, self.device
        S = self.depth.shape[0]
        Ls, Ld = self.s.pos_enc_basis, self.t.pos_enc_basis
        fine = self.n_fine > 0
        if fine and self.world > 1:
            raise RuntimeError("the graph-replayed hierarchical step runs on one rank (step_graph routes the sharded case to the host-launched step)")
        dsched = None if (fine or self._draws_injected()) else self._device_schedules()
        self._device_mode = dsched is not None
        Lsf, Ldf = (self.s_fine.pos_enc_basis, self.t_fine.pos_enc_basis) if fine else (0, 0)
        nf = S + Ls + Ld + Lsf + Ldf
        off64 = (4 * nf + 7) // 8 * 8
        self._rec_layout = (S, Ls, Ld, off64)
        self._rec_fine = (Lsf, Ldf)
        rec32 = rec64 = None
        if not self._device_mode:
            self._rec_host = [torch.empty(off64 + 32, dtype=torch.uint8).pin_memory() for _ in range(4)]
            self._rec_done = [None] * 4
            self._rec_dev = torch.zeros(off64 + 32, dtype=torch.uint8, device=dev)
            rec32 = self._rec_dev[: 4 * nf].view(torch.float32)
            rec64 = self._rec_dev[off64:].view(torch.float64)
        R = c.img_sample_size
        lo, hi = (R * self.rank) // self.world, (R * (self.rank + 1)) // self.world
        self._slice = (lo, hi)
        self._ids_buf = torch.zeros(hi - lo, dtype=torch.int64, device=dev)
        self._iter_dev = torch.zeros(1, dtype=torch.int64, device=dev)       # the iteration the next replay runs (device mode)
        self._graph_iter = None
        # (every tensor the captured kernels read must outlive the graph: a local that dies with this function hands its memory
        # back to the allocator and the replays read whatever the next owner wrote there -- tests/test_configs.py)
        nets = [self.t, self.s] + ([self.t_fine, self.s_fine] if fine else [])           # self.params order
        self.adam = FusedAdam(nets, lr=c.lr, end_factor=c.lr_end_factor, total_iters=c.lr_decay_steps, iter_counter=self._iter_dev)
        bs, bd = self.s._binding, self.t._binding
        split = self.world > 1 or self.always_allreduce
        out = {}
        nd, ns = bd.flat.numel(), bs.flat.numel()
        # the step's ONE flat buffer: [gradients in self.params order | the 13 loss terms as f32] -- what the all-reduce sums and Adam reads
        self._flat = torch.zeros(sum(b.flat.numel() for b in self.adam.bindings) + 13, dtype=torch.float32, device=dev)
        out["flat"] = self._flat
        if fine:
            self._u_buf = torch.zeros((hi - lo, self.n_fine), dtype=torch.float32, device=dev)

        def front_fine():
            z = MH.randomize_depth(self.depth, dev, rec32[:S])
            terms, terms_f, order = self._fine_device_work(self._ids_buf, R, z, self._u_buf, (0.0, 0.0, 0.0, 0.0), weights_dev=rec64)
            total = terms.clone()
            total[0] = terms[0] + terms_f[0]
            out["terms"], out["terms_f"] = total, terms_f
            self._flat.copy_(torch.cat([g for _, _, g in order] + [terms_f.to(torch.float32)]))       # (+ the fine pass's terms: the early-stop pair is among them)

        def front():
            if self._device_mode:
                prep = begin_step(self._sampler(), 0, lo, hi - lo, self.data.rays_train, self._phases_i64(), self.depth, iter_dev=self._iter_dev,
                                  schedules=self._sched, bad_ids=self._bad_ids)
                wdev = self._weights_dev
            else:
                prep = self._prepare(self._ids_buf, rec32[:S])
                wdev = rec64
            o, d, gt, w, phases, z, dists = prep
            # (one micro-batch at the bench size; several where the whole batch's forward store would not fit -- BASELINE configs[3]'s
            # 512^2 x 256 in f32 keeps 376 GiB -- instead of the recompute backward the graph step silently fell back to until round 3)
            terms, _, _ = self._micro_batched(o, d, gt, w, phases, z, dists, R, (0.0, 0.0, 0.0, 0.0), wdev, flat_out=self._flat)
            out["terms"] = terms

        def back():
            off, gs = 0, []
            for b in self.adam.bindings:
                gs.append(self._flat[off:off + b.flat.numel()])
                off += b.flat.numel()
            self.adam.step(gs)

        if fine:
            front = front_fine

        if self._device_mode:
            self._sched, wins, self._weights_dev = dsched
            self._win_dev = wins
            bs.static_window, bd.static_window = wins[0], wins[1]
        else:
            bs.static_window = rec32[S:S + Ls] if Ls > 0 else None
            bd.static_window = rec32[S + Ls:S + Ls + Ld] if Ld > 0 else None
            if (Ls == Ld and Ls > 0 and c.static_pos_enc == c.temp_pos_enc and c.static_pos_enc_window_decay_steps == c.temp_pos_enc_window_decay_steps
                    and getattr(self.s, "pos_enc_window_start", None) == getattr(self.t, "pos_enc_window_start", None)):
                bd.static_window = bs.static_window     # identical schedules: one vector, the encoded input is stored once
        if fine:
            bsf, bdf = self.s_fine._binding, self.t_fine._binding
            o0 = S + Ls + Ld
            bsf.static_window = rec32[o0:o0 + Lsf] if Lsf > 0 else None
            bdf.static_window = rec32[o0 + Lsf:o0 + Lsf + Ldf] if Ldf > 0 else None
        self._in_graph_setup = True
        try:
            if not self._device_mode:
                self._write_record(0)
            saved = [b.flat.clone() for b in self.adam.bindings]
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # eager warm-up on the capture stream's allocator pool
                front()
                back()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for b, keep_flat in zip(self.adam.bindings, saved):      # undo the warm-up's optimiser step
                b.flat.copy_(keep_flat)
            for t in self.adam.exp_avg + self.adam.exp_avg_sq:
                t.zero_()
            self.adam._step.zero_()
            self._iter_dev.zero_()
            self._graphs = []
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                front()
                if split and self._capture_collective():
                    # the step's ONE collective inside the graph (torch records RCCL collectives issued on a capturing stream): a sharded step
                    # is then one replay -- front, all-reduce, Adam -- instead of two graphs with a host-issued collective between them
                    dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
                    split = False
                    self._collective_captured = True
                if not split:
                    back()
            self._graphs.append(g1)
            if split:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool()):
                    back()
                self._graphs.append(g2)
        finally:
            self._in_graph_setup = False
            bs.static_window = bd.static_window = None
            if fine:
                self.s_fine._binding.static_window = self.t_fine._binding.static_window = None
        self._graph_out = out

    def _capture_collective(self) -> bool:
        """Whether the gradient all-reduce is recorded INTO the step graph: with the RCCL backend (NERFCA_GRAPH_COLLECTIVE=0: two graph
        segments with a host-issued collective between them, the round-5 structure; gloo collectives run on the host and cannot be captured)."""
        import os
        if os.environ.get("NERFCA_GRAPH_COLLECTIVE", "1") == "0" or not dist.is_initialized():
            return False
        return dist.get_backend() == "nccl"

    def _phases_i64(self) -> torch.Tensor:
        pt = self.data.phases_train
        if pt.dtype != torch.int64:
            if getattr(self, "_phases64", None) is None:
                self._phases64 = pt.to(torch.int64)
            pt = self._phases64
        return pt

    def _write_record(self, n_iter: int) -> None:
        """RECORD mode: fill the next pinned record with this step's host-side scalars and enqueue its copy to the device."""
        S, Ls, Ld, off64 = self._rec_layout
        k = n_iter % len(self._rec_host)
        if self._rec_done[k] is not None:
            self._rec_done[k].synchronize()          # the copy that last used this pinned buffer has run
        host = self._rec_host[k]
        h32 = host[: 4 * (S + Ls + Ld + sum(getattr(self, "_rec_fine", (0, 0))))].view(torch.float32)
        h32[:S] = self.draw_jitter(n_iter)
        if Ls > 0:
            h32[S:S + Ls] = self.s._band_window()
        if Ld > 0:
            h32[S + Ls:S + Ls + Ld] = self.t._band_window()
        Lsf, Ldf = getattr(self, "_rec_fine", (0, 0))
        if Lsf > 0:
            h32[S + Ls + Ld:S + Ls + Ld + Lsf] = self.s_fine._band_window()
        if Ldf > 0:
            h32[S + Ls + Ld + Lsf:S + Ls + Ld + Lsf + Ldf] = self.t_fine._band_window()
        host[off64:].view(torch.float64).copy_(torch.tensor([float(x) for x in self.loss_weights(n_iter)], dtype=torch.float64))
        self._rec_dev.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._rec_done[k] = ev

    @_scoped
    def step_graph(self, n_iter: int):
        """``step_fused`` with the device work replayed from a captured HIP graph and the library's Adam + LinearLR
        (its own moment buffers: do not interleave with ``step``/``step_fused`` in one run).  In DEVICE mode (see ``_graph_setup``) the host
        launches the graph and nothing else: ids, jitter, windows and loss weights of iteration ``n_iter`` are made by the graph's first kernel
        from a device counter that the graph's Adam increments (a call that is not the successor of the previous one sets the counter first).
        Returns (loss, pixel, terms) as ``step_fused`` does; the tensors are overwritten by the next call.  The modules' own schedule state
        (``freq_mask_alpha``) is NOT advanced per step in DEVICE mode: ``update_windows(n)`` before ``evaluate`` / ``save``, as callers do."""
        if self.n_fine > 0 and self.world > 1:
            # The hierarchical step under ray sharding has four collectives between its kernels (the sampler's batch-wide maximum,
            # ray 0's depths, and both again on the way back): nothing long enough to replay is left between them.  Same step,
            # launched from the host, torch Adam.
            return self._step_fused_fine(n_iter)
        if getattr(self, "_graphs", None) is None:
            self.update_windows(n_iter)
            self._graph_setup()
        if self._device_mode:
            if self._graph_iter != n_iter:
                self._iter_dev.fill_(n_iter)
        else:
            self.update_windows(n_iter)
            lo, hi = self._slice
            self._ids_buf.copy_(self._ids_for(n_iter)[lo:hi])
            if self.n_fine > 0:
                self._u_buf.copy_(self.draw_fine_u(n_iter)[lo:hi], non_blocking=False)       # sample_pdf's uniform draws of this step
            self._write_record(n_iter)
            self._prefetch_ids(n_iter + 1)          # (before the replay: the small kernels of the draw run beside the graph's first kernels)
        self._graph_iter = n_iter + 1
        self._graphs[0].replay()
        if len(self._graphs) > 1:
            if self.world > 1 or self.always_allreduce:
                dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
            self._graphs[1].replay()
        terms = self._graph_out["terms"]
        if self.n_fine > 0:
            self.last_fine_terms = self._graph_out["terms_f"]
        # the tail of the flat buffer: the 13 terms as f32, summed over the ranks by the gradient all-reduce; [dynamic entropy, favor] = 8, 5
        if n_iter >= self.cfg.static_pos_enc_window_decay_steps:
            tail = self._flat[-13:]
            self._note_early_stop(n_iter, torch.stack([tail[8], tail[5]]))
        else:
            self._note_early_stop(n_iter, None)
        return terms[0], terms[1], terms

    def _prefetch_ids(self, n_iter: int) -> None:
        """Draw the ray ids of iteration ``n_iter`` on a side stream.  The importance sampling of run_composite.py:250-260 is a dozen small
        launches (two index draws, a concatenation, a random permutation = a radix sort, gathers) that depend on nothing of the step
        before: drawn between two graph replays on the replay's stream they are ~75 us of nothing else happening -- 10 % of a step at the
        reference's default batch of 1 024 rays, 0.5 % at 65 536 (profiles/r05_small_batch_trace.txt).  ``_ids_for`` hands the
        prefetched vector out if the caller does ask for that iteration next (same generator, same seed: the same ids either way)."""
        if getattr(self, "_ids_stream", None) is None:
            self._ids_stream = torch.cuda.Stream(device=self.device)
            self._ids_stream.wait_stream(torch.cuda.current_stream())       # (once: the id tables were created on the caller's stream)
        with torch.cuda.stream(self._ids_stream):
            ids = self.draw_ray_ids_device(n_iter)
            ev = torch.cuda.Event()
            ev.record()
        self._ids_prefetch = (n_iter, ids, ev)

    def _ids_for(self, n_iter: int) -> torch.Tensor:
        pf, self._ids_prefetch = getattr(self, "_ids_prefetch", None), None
        if pf is not None and pf[0] == n_iter:
            _, ids, ev = pf
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            ids.record_stream(cur)          # (allocated on the side stream, consumed here)
            return ids
        return self.draw_ray_ids_device(n_iter)

    def allreduce_grads(self, extra=None):
        """ONE all-reduce(SUM) over a flat f32 buffer of every gradient (152 914 floats by default); ``extra`` (a small f32
        tensor, e.g. the early-stop pair) rides behind the gradients and comes back summed."""
        grads = [p.grad for p in self.params]
        both = grads + ([extra.to(grads[0].dtype)] if extra is not None else [])
        flat = torch._utils._flatten_dense_tensors(both)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        outs = torch._utils._unflatten_dense_tensors(flat, both)
        for g, r in zip(grads, outs):
            g.copy_(r)
        return outs[-1] if extra is not None else None

    def _eval_terms(self, pix, gt, ones, sig_s, sig_d, dists, n_iter):
        """(test_loss, pixel, favor, blendw, static entropy, dynamic entropy) of the held-out view: the HIP loss kernel, values only --
        all terms and the assembled test loss in one pass (unit pixel weights; run_composite.py:363-391)."""
        from ..fused import fused_losses
        tk, _, _, _ = fused_losses(pix, gt, ones, sig_s, sig_d, dists, self.cfg, self.loss_weights(n_iter), inv_R=1.0 / pix.shape[0], want_grads=False)
        return tk[0], tk[1], tk[5], tk[2], tk[6], tk[8]

    # -- held-out view (run_composite.py:346-413) ------------------------------------------------
    @torch.no_grad()
    @_scoped
    def evaluate(self, n_iter: int, chunk_rays: int = 65536):
        """The display_every block of the reference: render the held-out view (one fixed depth jitter drawn at set-up,
        run_composite.py:134), the weighted pixel loss with unit weights, all loss terms and ``test_loss`` with this
        iteration's weights, ``test_psnr = -10 log10(test_loss)`` (the reference's definition, :391), and the static /
        dynamic images each field renders on its own (:407-413; un-normalised ``I0 - sum sigma dists``)."""
        c, d, dev = self.cfg, self.data, self.device
        self.check_ray_ids()
        if getattr(self, "_test_jitter", None) is None:
            self._test_jitter = torch.rand(self.depth.shape, generator=torch.Generator().manual_seed(self.seed * 7919 + 1))
        z = MH.randomize_depth(self.depth, dev, self._test_jitter)
        dists = MH._interval_lengths(z, d.test_directions)
        pix, sig_s, sig_d = [], [], []
        for i in range(0, d.test_origins.shape[0], chunk_rays):
            o, dd = d.test_origins[i:i + chunk_rays], d.test_directions[i:i + chunk_rays]
            ph = torch.full((o.shape[0],), d.test_phase, dtype=torch.int32, device=dev)
            I0 = torch.full((o.shape[0],), d.geo["max_pixel_value"], dtype=torch.float32, device=dev)
            p, a, b = self._render(self.s, self.t, o, dd, ph, I0, z, dists, c.output_activation)
            pix.append(p); sig_s.append(a); sig_d.append(b)
        pix, sig_s, sig_d = torch.cat(pix), torch.cat(sig_s), torch.cat(sig_d)
        gt = d.test_image.to(pix.dtype)
        ones = torch.ones_like(gt)
        test_loss, pixel, favor, blendw, s_ent, d_ent = self._eval_terms(pix, gt, ones, sig_s, sig_d, dists, n_iter)
        I0 = d.geo["max_pixel_value"]
        mse = ((pix.float() - d.test_image) ** 2).mean()
        return {"test_loss": test_loss, "test_psnr": -10.0 * torch.log10(test_loss), "test_pixel_loss_coarse": pixel,
                "test_favor_s_loss": favor, "test_blendw": blendw, "test_s_entropy_loss": s_ent, "test_d_entropy_loss": d_ent,
                "test_mse": mse, "test_psnr_mse": -10.0 * torch.log10(mse), "pred": pix.float(),
                "pred_static": (I0 - (sig_s.double() * dists).sum(-1)).float(), "pred_dynamic": (I0 - (sig_d.double() * dists).sum(-1)).float()}


def normalize_image(img: torch.Tensor) -> torch.Tensor:
    """(x - min) / (max - min), as the reference does before logging images (run_composite.py:405-413)."""
    return (img - img.min()) / (img.max() - img.min())


class StaticTrainer:
    """The static-only loop of train/run_nerf.py:186-231 (BASELINE configs[0]) around the fused single-field render:
    loss = weighted MSE + occl_weight_start * sum(compute_occl_loss(sigma, dists, occl_reg_perc)); Adam + LinearLR.
    Ray sharding and the gradient all-reduce work as in ``CompositeTrainer``."""

    def __init__(self, cfg: TrainConfig, static_model, data, device, rank: int = 0, world: int = 1, seed: int = 0,
                 fused_adam: Optional[bool] = None):
        self.cfg, self.s, self.data, self.device = cfg, static_model, data, device
        self.rank, self.world, self.seed = rank, world, seed
        self.params = [p for _, p in static_model.named_parameters()]          # run_nerf.py:159-161
        kw = {}
        if fused_adam is None:
            fused_adam = torch.device(device).type == "cuda"
        if fused_adam and all(p.is_contiguous() for p in self.params):      # (nets padded to a kernel width own strided views: torch's fused Adam wants one layout)
            kw["fused"] = True
        self.opt = torch.optim.Adam([{"params": self.params, "lr": cfg.lr}], lr=cfg.lr, **kw)
        self.sched = torch.optim.lr_scheduler.LinearLR(self.opt, start_factor=1, end_factor=cfg.lr_end_factor, total_iters=cfg.lr_decay_steps)
        self.depth = MH_depth(data.geo, cfg.depth_samples_per_ray_coarse, device)
        self.I0 = torch.full((cfg.img_sample_size,), data.geo["max_pixel_value"], dtype=torch.float32, device=device)
        self.n_var = int((cfg.var_sample_perc / 100.0) * cfg.img_sample_size) if cfg.var_sample_perc > 0 else 0
        self._dev_gen = None

    def update_window(self, n_iter: int) -> None:
        """run_nerf.py:191-197."""
        c = self.cfg
        if c.static_pos_enc == "nerfies_windowed" and n_iter > 0:
            self.s.update_windowed_alpha(n_iter, c.static_pos_enc_window_decay_steps)
        elif c.static_pos_enc == "free_windowed":
            self.s.update_freq_mask_alpha(n_iter, c.static_pos_enc_window_decay_steps)

    draw_ray_ids = CompositeTrainer.draw_ray_ids
    draw_ray_ids_device = CompositeTrainer.draw_ray_ids_device
    draw_jitter = CompositeTrainer.draw_jitter
    _sampler = CompositeTrainer._sampler
    sampler = None

    def loss_on(self, n_iter: int, origins, directions, I0, gt, w, t_rand, share: float = 1.0):
        """Loss of one ray set (run_nerf.py:218-226); ``share`` = local / global ray count under sharding."""
        c = self.cfg
        pix, sigma, dists = MH.obtain_train_predictions_static(self.s, origins, directions, I0, self.depth, c.output_activation,
                                                               c.batch_size, self.device, t_rand=t_rand)
        pixel = MH.weighted_MSELoss()(pix, gt, w).mean() * share
        occl = torch.sum(MH.compute_occl_loss(sigma, dists, c.occl_reg_perc)) * share
        return pixel + c.occl_weight_start * occl, pixel, occl, pix

    def step(self, n_iter: int):
        c = self.cfg
        self.update_window(n_iter)
        on_gpu = torch.device(self.device).type == "cuda"
        ids = self.draw_ray_ids_device(n_iter) if on_gpu else self.draw_ray_ids(n_iter)
        R = len(ids)
        lo, hi = (R * self.rank) // self.world, (R * (self.rank + 1)) // self.world
        my = ids[lo:hi] if torch.is_tensor(ids) else torch.as_tensor(ids[lo:hi], device=self.device)
        rays = self.data.rays_train.index_select(0, my)
        loss, pixel, occl, _ = self.loss_on(n_iter, rays[:, 0, :], rays[:, 1, :], self.I0[: hi - lo], rays[:, 2, 0], rays[:, 3, 0],
                                            self.draw_jitter(n_iter), share=(hi - lo) / R)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.world > 1:
            CompositeTrainer.allreduce_grads(self)
        self.opt.step()
        self.sched.step()
        return loss.detach(), pixel.detach(), occl.detach()


def MH_depth(geo, n, device):
    from .data_helpers import create_depth_values
    return create_depth_values(geo["near_thresh"], geo["far_thresh"], n, device)
