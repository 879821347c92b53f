"""TEST INFRASTRUCTURE: a CompositeTrainer whose device operations are injected.

The product trainer (nerf-ca_amd/train/trainer.py) has one path, the HIP library.  What the data-parallel bookkeeping needs to be
tested for -- slices of one global batch, sum- vs mean-type loss terms, the one all-reduce, the fine pass's cross-rank maximum and
ray-0 broadcast (train/run_composite.py:250-312) -- does not depend on who renders, so the CPU / gloo tests run it
with the CPU oracle behind the renderer and the sampler.  This subclass is that seam: renderer and fine sampler as callables, the
batch preparation and the evaluation's loss terms as the reference's torch operations (they run on any device), autograd step only.
"""
import torch

from nerfca_amd import losses as LS
from nerfca_amd.train import model_helpers as MH
from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig  # noqa: F401


def all_terms(static_sigma, temp_sigma, dists, weighted_pixs, run_args):
    """compute_losses (train/model_helpers.py:250-262) assembled in torch from the part functions the package exports: the reference's
    11-tuple on any device.  (The product's compute_losses is the HIP loss kernel; this restatement serves the CPU tests and is itself
    checked against the reference's goldens, tests/test_host_cpu.py::test_losses_match_reference.)"""
    bw, top_s, top_d = LS.blend_weight(static_sigma, temp_sigma)
    favor = LS.binary_entropy_of_blend(bw, skewness=run_args.skewness_val)
    s_ent, s_sum = LS.ray_entropy(static_sigma, dists, mask_threshold=run_args.entro_mask_thre)
    d_ent, d_sum = LS.ray_entropy(temp_sigma, dists, mask_threshold=run_args.entro_mask_thre, use_weighting=run_args.entro_use_weighting,
                                  weighted_pixs=weighted_pixs, weighted_thresh=run_args.entro_weighted_thresh)
    occl = LS.occlusion(temp_sigma, dists, run_args.occl_reg_perc)
    mass = static_sigma * dists
    return bw.mean(), top_s, top_d, favor, s_ent, s_sum, d_ent, d_sum, occl, mass.sum(), (mass ** 2).sum()


def weighted_sq_err(preds, gts, weights):
    """weighted_MSELoss.forward (train/model_helpers.py:284-288) in torch (the caller takes .mean())."""
    return (preds - gts) ** 2 * weights


class InjectedTrainer(CompositeTrainer):
    def __init__(self, *args, render=None, fine_sampler=None, **kw):
        """``render(static_model, temp_model, o, d, phases, I0, z, dists, act=...) -> (pix, sigma_s, sigma_d)``;
        ``fine_sampler(sig_s, sig_d, z, u, reduce_max=...) -> z_all`` (fused.fine_depths's signature).  Either may be None: the
        product's HIP operation then stays in place (a GPU test that injects only a CPU sampler)."""
        kw["fused_loss"] = False                 # the injected operations run under autograd
        super().__init__(*args, **kw)
        self._inj_render, self._inj_sampler = render, fine_sampler

    def _render(self, static_model, temp_model, o, d, phases, I0, z, dists, act):
        if self._inj_render is None:
            return super()._render(static_model, temp_model, o, d, phases, I0, z, dists, act)
        return self._inj_render(static_model, temp_model, o, d, phases, I0, z, dists, act=act)

    def _fine_depths(self, sig_s, sig_d, z, u, reduce_max):
        if self._inj_sampler is None:
            return super()._fine_depths(sig_s, sig_d, z, u, reduce_max)
        return self._inj_sampler(sig_s, sig_d, z, u, reduce_max=reduce_max)

    def _fine_depths_autograd(self, sig_s, sig_d, z, u, reduce_max):
        if self._inj_sampler is None:
            return super()._fine_depths_autograd(sig_s, sig_d, z, u, reduce_max)
        # the reference's own operations under autograd (train/model_helpers.py:135-146)
        tot = sig_s + sig_d
        wts = torch.cat([torch.ones_like(tot[:, :1]) * 1e-10, torch.abs(tot[:, 1:] - tot[:, :-1])], dim=-1)
        wts = wts / MH._BatchMax.apply(wts, reduce_max)
        zrep = z[None, :].repeat(sig_s.shape[0], 1)
        mid = 0.5 * (zrep[..., 1:] + zrep[..., :-1])
        z_pdf = MH.sample_pdf(mid, wts[..., 1:-1], self.n_fine, self.device, u=u)
        return torch.sort(torch.cat([z_pdf, zrep.detach()], -1), -1)[0]

    def _pixel_loss(self, pix, gt, w):
        return weighted_sq_err(pix, gt, w).mean()             # train/model_helpers.py:284-288 as torch operations (any device)

    def _loss_terms(self, sig_s, sig_d, dists, w):
        return all_terms(sig_s, sig_d, dists, w, self.cfg)           # train/model_helpers.py:250-262 as torch operations

    def _draw_ids(self, n_iter):
        on_gpu = torch.device(self.device).type == "cuda"
        return self.draw_ray_ids_device(n_iter) if on_gpu else self.draw_ray_ids(n_iter)

    def _prepare(self, my, t_rand):
        rt = self.data.rays_train
        if rt.is_cuda:
            return super()._prepare(my, t_rand)
        rays = rt.index_select(0, my)                                     # train/run_composite.py:262-273
        phases = self.data.phases_train.index_select(0, my)
        o, d, gt, w = rays[:, 0, :], rays[:, 1, :], rays[:, 2, 0], rays[:, 3, 0]
        z = MH.randomize_depth(self.depth, self.device, t_rand)           # train/model_helpers.py:3-12
        return o, d, gt, w, phases, z, MH._interval_lengths(z, d)

    def _eval_terms(self, pix, gt, ones, sig_s, sig_d, dists, n_iter):
        if pix.is_cuda:
            return super()._eval_terms(pix, gt, ones, sig_s, sig_d, dists, n_iter)
        pixel = weighted_sq_err(pix, gt, ones).mean()
        terms = all_terms(sig_s, sig_d, dists, ones, self.cfg)
        fav_w, ent_w, occ_w, l1_w = self.loss_weights(n_iter)
        test_loss = pixel + fav_w * terms[3] + ent_w * terms[6] + occ_w * terms[8] + l1_w * terms[10] + l1_w * terms[9]
        return test_loss, pixel, terms[3], terms[0], terms[4], terms[6]
