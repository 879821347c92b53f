"""PSNR gates of the throughput arithmetic ("bf16: PSNR on the held-out view within 0.1 dB of fp32 after equal steps";
BASELINE.json: "... at matched PSNR").  Every leg trains the composite model on the synthetic data set from the same initial
weights, ray batches and depth jitter; f32 is the mode that is within 1e-5 of the reference's arithmetic per step
(tests/test_hip_parity.py), `test_psnr` = -10 log10(test loss) is the reference's own definition (train/run_composite.py:391),
`psnr_mse` = -10 log10(MSE of the held-out view).

Two batch regimes:
  * the bench configuration, 65 536 rays x 192 samples per step (12.6 M samples: the rounding noise of a step averages out);
  * the reference's default batch, 1 024 rays x 500 samples (train/composite.txt:25,40), 5 000 steps, where it does not -- five
    seeds (initial weights, ray batches, jitter), f32 against bf16 with fp8 staging (the planner's choice) and with bf16 staging.
    The f32 trajectory itself is stable (initial weights moved by 1e-6: final PSNR moves by <= 0.002 dB,
    profiles/r03_psnr_small_batch.json), so the per-seed gaps are properties of the arithmetic, not trajectory noise; they scatter
    around their mean with a standard deviation of 0.06 - 0.13 dB, and the gate is on the MEAN over the seeds.
"""
import importlib.util
import os
import statistics
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def psnr_run():
    spec = importlib.util.spec_from_file_location("psnr_run", os.path.join(ROOT, "tools", "psnr_run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bf16_psnr_gate_at_bench_configuration(dev, psnr_run):
    """300 steps of 65 536 rays x 192 samples on the 256^2 data set (40 training images, one held-out view): f32 against the bf16
    mode as the bench runs it (the planner's default: fp8 staging, resident kernels, mode-5 backward) and against bf16 staging."""
    from nerfca_amd import synthetic
    args = SimpleNamespace(steps=300, every=300, rays=65536, samples=192, det=256, graph=False)
    data = synthetic.make_dataset(256, 192, dev, views=synthetic.TRAIN_VIEWS)
    res = {v: psnr_run.run(v, args, dev, data) for v in ("f32", "bf16", "bf16_bf16stage")}
    assert res["bf16"]["stage_fp8_in_effect"] is True and res["bf16_bf16stage"]["stage_fp8_in_effect"] is False
    fin = {v: r["curve"][-1] for v, r in res.items()}
    print("held-out PSNR after 300 steps at 65 536 x 192: untrained %.2f dB; " % res["f32"]["curve"][0]["psnr_mse_db"]
          + "; ".join(f"{v} {c['psnr_mse_db']:.3f} dB (reference's test_psnr {c['test_psnr_reference_def_db']:.3f})" for v, c in fin.items()))
    for v in ("f32", "bf16", "bf16_bf16stage"):
        assert fin[v]["psnr_mse_db"] - res[v]["curve"][0]["psnr_mse_db"] > 25.0, v
    for v in ("bf16", "bf16_bf16stage"):
        assert abs(fin[v]["psnr_mse_db"] - fin["f32"]["psnr_mse_db"]) < 0.1, (v, fin)
        assert abs(fin[v]["test_psnr_reference_def_db"] - fin["f32"]["test_psnr_reference_def_db"]) < 0.1, (v, fin)


@pytest.mark.timeout(900)
def test_psnr_gate_at_reference_default_batch(dev, psnr_run):
    """1 024 rays x 500 samples per step, 5 000 graph-replayed steps, five seeds: the mean gap to f32 of the planner's default
    (fp8 staging) is within 0.1 dB on both PSNR definitions, fp8 staging is not worse than bf16 staging beyond 0.05 dB on the
    mean, and no single run strays by more than 0.4 dB."""
    from nerfca_amd import synthetic
    args = SimpleNamespace(steps=5000, every=5000, rays=1024, samples=500, det=256, graph=True)
    data = synthetic.make_dataset(256, 500, dev, views=synthetic.TRAIN_VIEWS)
    seeds = [0, 1, 2, 3, 4]
    variants = ("f32", "bf16", "bf16_bf16stage")
    runs = {v: [psnr_run.run(v, args, dev, data, seed=sd) for sd in seeds] for v in variants}
    assert all(r["stage_fp8_in_effect"] is True for r in runs["bf16"]), "the planner's default at this batch size is fp8 staging"
    gaps = psnr_run.gap_statistics(runs, list(variants), tail=1)
    for v in ("bf16", "bf16_bf16stage"):
        for key in ("psnr_mse_db", "test_psnr_reference_def_db"):
            g = gaps[v][key]
            print(f"{v:16s} {key:28s} mean gap {g['final_gap_mean']:+.3f} dB, sd {g['final_gap_sd']:.3f}, per seed {[round(x, 3) for x in g['final_gap_per_seed']]}")
    for v in variants:
        for r in runs[v]:
            assert r["curve"][-1]["psnr_mse_db"] - r["curve"][0]["psnr_mse_db"] > 25.0, (v, r["seed"])
    for key in ("psnr_mse_db", "test_psnr_reference_def_db"):
        g8, g16 = gaps["bf16"][key], gaps["bf16_bf16stage"][key]
        assert abs(g8["final_gap_mean"]) < 0.1, (key, g8)
        assert g8["final_gap_mean"] > g16["final_gap_mean"] - 0.05, (key, g8["final_gap_mean"], g16["final_gap_mean"])
        assert max(abs(x) for x in g8["final_gap_per_seed"]) < 0.4, (key, g8)
    assert statistics.mean(r["wall_s_incl_eval"] for r in runs["bf16"]) < statistics.mean(r["wall_s_incl_eval"] for r in runs["f32"])
