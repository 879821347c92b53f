"""What the reference logs, without wandb: the dictionaries of train/run_composite.py:314-344 (every ``log_every`` steps) and
:393-404 (every ``display_every`` steps) under the reference's own keys, as plain floats, one JSON object per line.  wandb itself
is out of scope; a maintainer who wants it passes ``JsonlLogger.records`` entries to ``wandb.log``."""
import json
import math
import time

from .. import _capi


def train_record(trainer, n_iter: int, terms, start_time: float, fine_terms=None) -> dict:
    """``terms`` = the f64[13] vector a fused / graph step returns (under ray sharding: ``trainer.global_terms(terms)``);
    ``fine_terms`` = ``trainer.last_fine_terms`` when a fine model pair trains.  One host sync (the terms are read)."""
    t = [float(x) for x in terms]
    name = {k: i for i, k in enumerate(_capi.TERM_NAMES)}
    fav_w, ent_w, occ_w, l1_w = trainer.loss_weights(n_iter)
    loss = t[name["loss"]]
    ft = [float(x) for x in fine_terms] if fine_terms is not None else None
    src = ft if ft is not None else t                      # run_composite.py:298-301: the fine pass's regularisers replace the coarse ones
    rec = {"train_loss": loss, "train_psnr": -10.0 * math.log10(loss) if loss > 0 else float("inf"),
           "train_pixel_loss_coarse": t[name["pixel"]], "train_pixel_loss_fine": ft[name["pixel"]] if ft is not None else 0.0,
           "train_blendw": src[name["blendw"]], "train_sigma_s_max": src[name["sigma_s_max"]], "train_sigma_d_max": src[name["sigma_d_max"]],
           "train_favor_s_loss": src[name["favor_s"]], "train_s_entropy_loss": src[name["s_entropy"]], "train_d_entropy_loss": src[name["d_entropy"]],
           "train_s_entropy_sum": src[name["s_entropy_sum"]], "train_d_entropy_sum": src[name["d_entropy_sum"]], "train_d_occl_loss": src[name["d_occl"]],
           "train_s_l1": src[name["s_l1"]], "train_s_l2": src[name["s_l2"]],
           "favor_s_weight": float(fav_w), "dynamic_entro_weight": float(ent_w), "occl_weight": float(occ_w), "l1_weight": float(l1_w),
           "train_time": time.time() - start_time}
    c = trainer.cfg
    if "windowed" in c.static_pos_enc:
        rec["train_static_windowed"] = float(getattr(trainer.s, "windowed_alpha", 0.0))        # (:340-343: the models' current window position)
    if "windowed" in c.temp_pos_enc:
        rec["train_temp_windowed"] = float(getattr(trainer.t, "windowed_alpha", 0.0))
    return rec


def test_record(evaluation: dict) -> dict:
    """The scalar part of the display_every block (:393-404) from ``CompositeTrainer.evaluate``'s result."""
    keep = ("test_loss", "test_psnr", "test_pixel_loss_coarse", "test_favor_s_loss", "test_blendw", "test_s_entropy_loss", "test_d_entropy_loss")
    return {k: float(evaluation[k]) for k in keep}


class JsonlLogger:
    """Appends one JSON object per call to ``path`` (and keeps them in ``records``)."""

    def __init__(self, path=None):
        self.path, self.records = path, []

    def log(self, record: dict, step=None) -> None:
        rec = dict(record) if step is None else dict(record, step=int(step))
        self.records.append(rec)
        if self.path:
            with open(self.path, "a") as fh:
                fh.write(json.dumps(rec) + "\n")
