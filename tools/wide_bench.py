#!/usr/bin/env python3
"""Time the GENERAL kernels (nets beyond 128 units: nerf-ca_amd/csrc/nca_wide.hpp) on a composite render of synthetic rays:
forward, and forward + backward, per net width.  Prints one JSON line per width with ms, the f32 FLOP rate against the
v_mfma_f32_32x32x2_f32 peak (157.3 TFLOP/s; forward + backward = 3 x the forward's FLOPs) and the library's own per-kernel-class timing.

    python tools/wide_bench.py [--rays 8192] [--samples 192] [--widths 256,512] [--steps 5]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_F32_MFMA = 157.3e12


def measure(widths, rays=8192, samples=192, steps=5):
    """One record per net width: forward and forward + backward of a composite render through the drop-in models."""
    from nerfca_amd import _capi, render_rays, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    dev = torch.device("cuda:0")
    R, S = rays, samples
    gen = torch.Generator().manual_seed(0)
    o = (torch.rand(R, 3, generator=gen) * 0.1 + torch.tensor([3.0, -2.0, 2.5])).double().to(dev)
    d = (torch.rand(R, 3, generator=gen) - 0.5).double().to(dev)
    ph = torch.randint(0, 10, (R,), generator=gen).to(dev)
    z = torch.linspace(3.4, 5.6, S).to(dev)
    dists = torch.cat([z[1:] - z[:-1], torch.tensor([1e-10], device=dev)]).double()
    I0 = torch.full((R,), 2.16, device=dev)
    out = []
    for F in widths:
        torch.manual_seed(1)
        sdef, tdef = synthetic.net_definitions(dev, F=F)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        for m in (s, t):
            m.update_freq_mask_alpha(75000, 150000)
        NL = 1 + sdef["num_early_layers"]
        K0 = 3 * (1 + 2 * sdef["pos_enc_basis"])
        Fp = s._binding.net.F
        flop_fwd = 2.0 * R * S * ((K0 + (K0 + tdef["num_time_dim"])) * Fp + 2 * (NL - 1) * Fp * Fp + 2 * Fp)
        rec = {"F": F, "kernel_width": Fp, "layers": NL, "rays": R, "samples": S, "general": bool(_capi.net_is_general(s._binding.net)), "dtype": "f32",
               "peak_tflops": PEAK_F32_MFMA / 1e12}
        for what in ("fwd", "fwd_bwd"):
            def once():
                if what == "fwd":
                    with torch.no_grad():
                        return render_rays(s, t, o, d, ph, I0, z, dists)
                for m in (s, t):
                    for p in m.parameters():
                        p.grad = None
                pix, a_, b_ = render_rays(s, t, o, d, ph, I0, z, dists)
                (pix.sum() + a_.sum() + b_.sum()).backward()
            once()
            torch.cuda.synchronize()
            _capi.timing_reset()
            _capi.timing_enable(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                once()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / steps
            kms = {}
            for k in _capi.KERNEL_KINDS:
                tot, n = _capi.timing_read(k)
                if n:
                    kms[k] = round(tot / steps, 3)
            _capi.timing_enable(False)
            flop = flop_fwd * (1 if what == "fwd" else 3)          # forward, dgrad, wgrad (the backward runs from the forward's store: nothing recomputed)
            rec[what] = {"ms": round(ms, 3), "tflops": round(flop / ms / 1e9, 1), "frac_of_f32_mfma_peak": round(flop / (ms * 1e-3) / PEAK_F32_MFMA, 3),
                         "kernel_ms": kms}
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--widths", default="256,512")
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    for rec in measure([int(w) for w in a.widths.split(",")], a.rays, a.samples, a.steps):
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
