#!/usr/bin/env python3
"""Headline benchmark: training rays/sec of the composite NeRF-CA step on synthetic
256^2-detector x 192-samples/ray batches (BASELINE.json metric, configs[1]).

    python bench.py --gpus N --steps K --warmup W

For N > 1 and no RANK in the environment this process starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`,
one process per GPU over RCCL) before it touches the GPU, relays their output and exits with their code; started under
torch.distributed.run it is one of the ranks.

One "step" = one pass of the hot path over one batch: ray gather (GPU-resident table), fused
forward, all losses, fused backward, (all-reduce), Adam + LinearLR.  Inputs are resident in HBM when
the timed region starts.  Multi-GPU: STRONG scaling by default -- one global batch of --rays rays per step, split N ways, one
all-reduce of the flat gradient (BASELINE configs[2]); the same run also times the weak-scaled step (--rays per rank) and
reports it beside the headline.  The N = 1 line carries `predicted_scaling`: the measured single-GPU step at rays / N for N = 2, 4, 8
with the step's collective in place (a one-rank RCCL group: two graph segments with the all-reduce between them).

Rank 0 writes the FULL record (every sub-record and kernel table) to gpurun_out/bench_full.json (--full-record) and prints, as the LAST
stdout line, ONE compact JSON object under 4 KB (`headline_line`): the contract's keys, `roofline`, `cpu_baseline`, and three numbers per
extra leg.  `dtype` names the arithmetic of the timed step as the library's planner ran it: "bf16+fp8stage" = bf16 MLP contractions whose
layer inputs / output gradients cross HBM as e4m3 / e5m2 on their way to the weight-gradient kernel (which contracts them on the MX-fp8
matrix path), "bf16" = the same with bf16 staging, "f32" = the 1e-5 parity mode.  `roofline` is for the kernel with the largest share of
the step, measured with HIP events on the launch stream inside the library (nca_timing_*) over an eager pass of the same step (events
cannot be recorded inside a replayed graph: the table decomposes `eager_ms_per_step`); fractions are named by their denominator --
`mfma_frac` (dense MFMA peak of the pipe the contractions run on: 2 500 TFLOP/s bf16; 2 500 / 6 for the f32 mode, whose hidden layers are six
bf16 products per f32 product) and `hbm_frac` (8 TB/s).  `cpu_baseline` is the CPU oracle (reference-equivalent torch CPU ops) on a bounded
sample.  At N = 1 the default run also times: `baseline_config_dtype` (stage_fp8 = 0: BASELINE configs[1] as written -- bf16 operands
everywhere, nothing in 8 bits), `f32`, `latency_regime` (the reference's default batch, 1 024 rays x 500 samples), `configs3` (MAGIX 512^2 x
256, f32), `sustained` and `unfused_gpu_baseline` (the reference-equivalent torch ops run op by op on the same GPU: the denominator of the
north star's >= 10x).  --no-extras drops those; --psnr-steps N and --predicted-scaling add their records to the full record.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work per sample of the default nets (recompute NOT counted)
FLOP_FWD, FLOP_DGRAD, FLOP_WGRAD = 303104, 264704, 303104
# MI355X dense MFMA peaks (MI355X_MICROARCH.md).  The f32 mode's hidden-layer contractions run on the bf16 matrix cores
# as six bf16 products per f32 product (exact 3-way split), so the pipe it really uses peaks at 2500 / 6
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
PEAK_HBM_GBPS = 8000.0
# The weight-gradient kernel under fp8 staging reads, per sample and net, one 8-bit 128-wide block pair per layer (the layer's
# output gradient, e5m2, and its input, e4m3) -- except the last hidden layer's output gradient, which it rebuilds
# from 16 B of mask bits and 4 B: 2 nets x (4 x 256 + 128 + 20) B.  303 104 FLOP over those bytes = 129 FLOP/B, below the ridge
# of 2500 TFLOP/s / 8 TB/s = 312 FLOP/B: that kernel's roofline is the HBM one.
WGRAD_FP8_BYTES_PER_SAMPLE = 2 * (4 * 256 + 128 + 20)
# ... and out of the bf16 store (nothing in 8 bits): per sample and net four bf16 block pairs (2 x 256 B each), the 224-byte bf16 input block
# and, for the last hidden layer's rebuilt block, 16 B of mask bits + 4 B: 2 nets x (4 x 512 + 224 + 20) -- 66 FLOP/B, far below the ridge
WGRAD_BF16_BYTES_PER_SAMPLE = 2 * (4 * 512 + 224 + 20)
PEAK_F32_ON_BF16_PIPE = 2500.0 / 6.0
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02", "r01")          # committed PMC summaries, newest first
FLOP_STEP = FLOP_FWD + FLOP_DGRAD + FLOP_WGRAD            # 870 912 FLOP per sample of a training step
# the pipe a precision's contractions run on: the f32 mode's hidden layers are six bf16 products per f32 product
PIPE_PEAK_TFLOPS = {"f32": PEAK_F32_ON_BF16_PIPE, "bf16": 2500.0}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (SURVEY.md 8d: >= 50 steady-state steps after 10 warm-up)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=65536, help="rays per step per GPU (one full 256^2 detector)")
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--prec", default="bf16", choices=["f32", "bf16"],
                    help="bf16 = BASELINE configs[1] (bf16 MFMA operands, f32 accumulate/master weights, PSNR-gated); f32 = parity mode")
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays per step of the CPU baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU steps (plus one warm-up): ~20 s of CPU work at the defaults")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the f32 sub-record, the unfused GPU baseline and the PSNR record")
    ap.add_argument("--graph", action="store_true", help="(the default) replay the step from a captured HIP graph (library Adam+LinearLR); "
                    "per-kernel timings then come from a short eager pass after the timed region")
    ap.add_argument("--eager", action="store_true", help="launch every kernel of the step from the host instead (CompositeTrainer.step, torch Adam)")
    ap.add_argument("--unfused-gpu-rays", type=int, default=16384, help="rays/step of the unfused torch path on cuda:0 (0: skip)")
    ap.add_argument("--unfused-gpu-steps", type=int, default=3)
    ap.add_argument("--f32-steps", type=int, default=20, help="timed steps of the f32 sub-record (after --f32-warmup; 0: skip)")
    ap.add_argument("--f32-warmup", type=int, default=5)
    ap.add_argument("--pure-steps", type=int, default=20, help="timed steps of the bf16_pure sub-record (bf16 staging; 0: skip)")
    ap.add_argument("--sustained-steps", type=int, default=500, help="further graph steps after the timed region; the last 100 are timed (0: skip)")
    ap.add_argument("--kernel-steps", type=int, default=8, help="steps of the eager pass that times the kernels")
    ap.add_argument("--psnr-steps", type=int, default=0, help="steps of the PSNR record of the FULL record (64^2 detector, 256 rays/step, HIP f32 / HIP bf16 / CPU oracle; "
                    "0 = skip, the default: the PSNR evidence is tests/test_psnr_gates.py)")
    ap.add_argument("--predicted-scaling", action="store_true", help="add `predicted_scaling` to the full record: the measured one-GPU step of a rank's share (rays / N, N = 2, 4, 8) "
                    "under a one-rank RCCL group; the links' time in it is an ESTIMATE")
    ap.add_argument("--full-record", default=None, help="where the full record goes (default: gpurun_out/bench_full.json under the repository root); the LAST stdout "
                    "line is the compact headline record (< 4 KB) whatever this says")
    ap.add_argument("--configs3-steps", type=int, default=3, help="timed steps of the BASELINE configs[3] sub-record (MAGIX geometry, 8 sequences x 10 phases of 512^2, "
                    "256 samples per ray, f32, one full detector = 262 144 rays per step; 0: skip)")
    ap.add_argument("--latency-steps", type=int, default=200, help="timed steps of the latency-regime sub-record (the reference's default batch: 1 024 rays x 500 "
                    "samples per step, train/composite.txt:25,40; bf16 and f32, graph-replayed; 0: skip)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous bookkeeping only: without RANK in the environment print the launch this process would "
                    "make (and make it, so that every rank reports); as a rank print {rank, world, local_rank, device index} and exit BEFORE anything touches the GPU")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--torch-losses", action="store_true", help="losses + autograd in torch ops instead of the fused loss kernel")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default; BASELINE configs[2], SURVEY.md 8e's partitioning): ONE global batch of --rays rays is split N ways -- rank g takes "
                         "ids [g R/N, (g+1) R/N) of the same id vector; weak: every rank renders --rays rays per step.  At N = 1 they are the same run.  "
                         "With N > 1 the line carries the other mode as a sub-record (`weak_scaling` / `strong_scaling`) measured in the same run")
    args = ap.parse_args(argv)
    args.graph = not args.eager            # the whole step as one captured HIP graph unless --eager
    if args.scaling == "strong" and args.rays % max(args.gpus, 1):
        ap.error(f"--scaling strong: --rays {args.rays} does not divide by --gpus {args.gpus}")
    return args


def rays_per_rank(args, world: int) -> int:
    return args.rays // world if args.scaling == "strong" else args.rays


def host_cores() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup quota (the GPU box shows
    256 logical CPUs but grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process tree (this process has not touched
    the GPU and never will), one rank per GPU, rendezvous on 127.0.0.1."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if args.dry_run:
        print(json.dumps({"launch": cmd, "ranks": args.gpus, "HSA_ENABLE_IPC_MODE_LEGACY": env["HSA_ENABLE_IPC_MODE_LEGACY"]}), flush=True)
    return subprocess.run(cmd, env=env).returncode


def source_sha():
    """Hash of the kernel sources this library was built from: a committed PMC summary is replayed only for the sources it
    was taken on."""
    import re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "nerf-ca_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            text = open(os.path.join(csrc, name), encoding="utf-8", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)            # comments and layout do not change a kernel
            text = re.sub(r"//[^\n]*", " ", text)
            h.update(name.encode())
            h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def committed_pmc(prec, rays, samples, stage_fp8=None):
    """The committed PMC summaries of this configuration, newest round first: (traffic record, sq record, source) or Nones.
    These are REPLAYED figures (separate rocprofv3 --pmc passes cannot run inside a bench run): the line labels them with
    the file and commit they come from, and they are dropped when the kernel sources have changed since (source_sha)."""
    sha = source_sha()
    stem = f"{prec}_pure" if (prec == "bf16" and stage_fp8 is False) else prec        # (the bf16-staging passes are kept under <tag>_bf16_pure_*)
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, "profiles", f"{tag}_{stem}_pmc_traffic.json")
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        cfg = rec.get("config", {})
        if cfg.get("rays_per_step") != rays or cfg.get("samples_per_ray") != samples or cfg.get("prec") != prec:
            continue
        if rec.get("source_sha") != sha or (stage_fp8 is not None and cfg.get("stage_fp8", True) != stage_fp8):
            continue          # taken on other kernel sources / another staging: not this build's traffic
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{stem}_pmc_sq.json")))
        except (OSError, ValueError):
            sq = None
        return rec, sq, {"file": f"profiles/{tag}_{stem}_pmc_traffic.json", "taken_at_commit": rec.get("commit"), "source_sha": sha,
                         "kind": "replayed from a committed rocprofv3 --pmc pass over these kernel sources"}
    return None, None, None


def _oracle_leg(samples, data, R, steps, device):
    """`steps` reference-equivalent training steps (the oracle: plain torch ops + autograd + torch Adam) of R rays on
    `device`; returns seconds.  On cuda this is the unfused PyTorch-ROCm path of SURVEY.md 8(d)(ii)."""
    from oracle import nerfca_oracle as O
    S = samples
    gen = torch.Generator().manual_seed(0)
    ss, sd = O.NetSpec(num_filters=128), O.NetSpec(num_filters=128, num_time_dim=8)
    on = lambda t: None if t is None else t.to(device)
    ps = {k: on(v) for k, v in O.init_params(ss, gen).items()}
    pd = {k: on(v) for k, v in O.init_params(sd, gen).items()}

    class Trainer(O.OracleTrainer):
        def windows(self, n_iter):
            return tuple(on(w) for w in super().windows(n_iter))

    ids = torch.randint(0, data.rays_train.shape[0], (R,), generator=gen)
    rays = data.rays_train.cpu().index_select(0, ids)
    ph = data.phases_train.cpu().index_select(0, ids)
    o, d, gt, w = (on(t) for t in (rays[:, 0, :], rays[:, 1, :], rays[:, 2, 0], rays[:, 3, 0]))
    I0 = torch.full((R,), float(data.geo["max_pixel_value"]), device=device)
    z0 = O.depth_values(data.geo["near_thresh"], data.geo["far_thresh"], S)
    phs = on(ph[:, None].repeat(1, S))
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    prev = torch.get_default_device()
    torch.set_default_device(device)           # the oracle builds its small constants on the default device
    try:
        tr = Trainer(ps, ss, pd, sd)

        def one(i):
            zj = on(O.stratified_depths(z0.cpu(), torch.rand(S, generator=gen, device="cpu")))
            tr.step(75000 + i, o, d, phs, I0, zj, gt, w)

        one(0)
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            one(1 + i)
        sync()
        return time.perf_counter() - t0
    finally:
        torch.set_default_device(prev)


def cpu_baseline(args, data):
    """Reference-equivalent CPU path (oracle) on a bounded sample of the same workload."""
    cores = host_cores()
    torch.set_num_threads(cores)
    R, S = args.cpu_rays, args.samples
    dt = _oracle_leg(S, data, R, args.cpu_steps, "cpu")
    return {"value": R * args.cpu_steps / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} full training steps of {R} rays x {S} samples (same nets, losses, Adam) with torch CPU ops, "
                      f"{dt:.1f} s"}


def unfused_gpu_baseline(args, data):
    """The same reference-equivalent torch path run op by op on cuda:0 in f32 -- what the reference itself does on a GPU
    (the chunk loop of the reference is not needed for memory on 288 GB)."""
    R, steps = args.unfused_gpu_rays, args.unfused_gpu_steps
    dt = _oracle_leg(args.samples, data, R, steps, "cuda:0")
    return {"value": R * steps / dt, "unit": "rays/s", "dtype": "f32",
            "sample": f"{steps} full training steps of {R} rays x {args.samples} samples with unfused PyTorch-ROCm ops, {dt:.2f} s"}


def make_trainer(args, prec, data, dev, rank, world, use_pg, plan_opts=None, global_rays=None):
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    torch.manual_seed(1)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    nerfca_amd.set_precision(prec, s, t)
    cfg = TrainConfig(depth_samples_per_ray_coarse=args.samples, img_sample_size=global_rays if global_rays is not None else rays_per_rank(args, world) * world)
    tr = CompositeTrainer(cfg, s, t, data, dev, rank=rank, world=world, seed=0, fused_loss=not args.torch_losses, plan_opts=plan_opts)
    tr.always_allreduce = use_pg
    return tr


def kernel_table(args, timed_steps, plan, world=1):
    """Per-kernel HIP-event times of the span the library's timers covered, with algorithmic TFLOP/s (the recompute backward's
    repeated forward is NOT counted as work: its dgrad kernel is charged the dgrad FLOPs only)."""
    from nerfca_amd import _capi
    n_samp = rays_per_rank(args, world) * args.samples * timed_steps
    kern = {}
    for name, flop in (("fwd", FLOP_FWD), ("bwd_dgrad", FLOP_DGRAD), ("bwd_wgrad", FLOP_WGRAD), ("bwd_reduce", 0), ("loss", 0), ("pack", 0)):
        ms, n = _capi.timing_read(name)
        kern[name] = {"ms_total": ms, "launches": n, "avg_ms": ms / n if n else None, "ms_per_step": ms / max(timed_steps, 1),
                      "tflops": (flop * n_samp / (ms * 1e-3) / 1e12) if ms > 0 and flop else None}
    return kern


def roofline_of(args, prec, kern, eager_dt, plan, ms_per_step, world=1):
    """`eager_dt` = wall seconds of the eager pass the kernel table was taken over (what the table decomposes); `ms_per_step` = the
    timed region's.  The kernel the record is about is the one with the largest time per step, with a FIXED tie-break: the forward
    unless another kernel takes more than 10 % longer (forward and weight gradient are within a few percent of each other and
    trade places from box to box; a reader should not see the record change its subject with them).  `frac` divides by the peak of
    the pipe the contractions run on -- 2 500 TFLOP/s bf16; for f32 2 500 / 6, since its hidden layers are six bf16 products per
    f32 product (the 157.3 TFLOP/s f32-MFMA peak it does NOT use is kept as `frac_of_f32_mfma_peak`).  `step` is the whole step:
    870 912 FLOP per sample over the timed region's ms_per_step -- the one number that does not depend on which kernel leads."""
    big = max(("fwd", "bwd_dgrad", "bwd_wgrad"), key=lambda k: kern[k]["ms_per_step"])
    dom = big if kern[big]["ms_per_step"] > 1.10 * kern["fwd"]["ms_per_step"] else "fwd"
    peak = PIPE_PEAK_TFLOPS[prec]
    n_step = rays_per_rank(args, world) * args.samples
    fp8 = bool(plan.get("stage_fp8"))
    traffic_rec, sq_rec, source = committed_pmc(prec, rays_per_rank(args, world), args.samples, fp8 if prec == "bf16" else None)
    traffic = traffic_rec["kernels"][dom]["hbm_bytes_per_launch"] if traffic_rec and dom in traffic_rec.get("kernels", {}) else None
    issue = None
    if sq_rec and dom in sq_rec.get("kernels", {}):
        k = sq_rec["kernels"][dom]
        issue = {"mfma_busy": k["mfma_busy"], "valu_busy": k.get("valu_busy"), "valu_mfma_coexec": k.get("valu_mfma_coexec")}
    ksum = sum(v["ms_total"] for v in kern.values())
    roof = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
            "frac": kern[dom]["tflops"] / peak if kern[dom]["tflops"] else None,
            "traffic": traffic, "traffic_source": source if traffic is not None else None,
            "avg_launch_ms": kern[dom]["avg_ms"], "launches": kern[dom]["launches"],
            # the staged (non-algorithmic) HBM traffic of that kernel per second of its run time: what it is bound by in practice
            "staging_TBps": (traffic / (kern[dom]["avg_ms"] * 1e-3) / 1e12) if traffic and kern[dom]["avg_ms"] else None,
            "simd_issue_share_pmc": issue,
            # share of the eager pass (the span the table decomposes) spent in the dominant kernel / in all timed kernels
            "kernel_time_share": kern[dom]["ms_total"] / (eager_dt * 1e3) if eager_dt else None,
            "all_kernels_time_share": ksum / (eager_dt * 1e3) if eager_dt else None, "all_kernels": kern}
    step_tflops = FLOP_STEP * n_step / (ms_per_step * 1e-3) / 1e12
    roof["step"] = {"bound": "mfma", "achieved": step_tflops, "peak": peak, "unit": "TFLOP/s", "frac": step_tflops / peak,
                    "flop_per_sample": FLOP_STEP, "samples_per_step_per_gpu": n_step, "ms_per_step": ms_per_step,
                    "note": "whole step (forward + losses + backward + Adam) against the peak of the pipe its contractions run on; per GPU"}
    roof["kernel_choice"] = "largest time per step; the forward unless another kernel is > 10 % slower"
    if prec == "f32":
        roof["frac_of_f32_mfma_peak"] = kern[dom]["tflops"] / PEAK_TFLOPS["f32"] if kern[dom]["tflops"] else None
        roof["peak_note"] = "2 500 / 6 TFLOP/s: the hidden layers run as six bf16 products per f32 product on the bf16 matrix cores (exact 3-way split)"
    wg = kern["bwd_wgrad"]
    bf16_store = prec == "bf16" and not fp8 and (plan.get("fwd_store_format", 0) & 15) == 4
    if prec == "bf16" and (fp8 or bf16_store) and wg["avg_ms"]:
        # the weight-gradient kernel's own roofline is the HBM one (see WGRAD_FP8_BYTES_PER_SAMPLE / WGRAD_BF16_BYTES_PER_SAMPLE)
        nbytes = (WGRAD_FP8_BYTES_PER_SAMPLE if fp8 else WGRAD_BF16_BYTES_PER_SAMPLE) * n_step
        gbps = nbytes / (wg["avg_ms"] * 1e-3) / 1e9
        hbm = {"bound": "hbm", "kernel": "bwd_wgrad", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
               "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": wg["avg_ms"],
               "traffic": traffic_rec["kernels"]["bwd_wgrad"]["hbm_bytes_per_launch"] if traffic_rec and "bwd_wgrad" in traffic_rec.get("kernels", {}) else None}
        if dom == "bwd_wgrad":
            roof.update({k: hbm[k] for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")})
            roof["mfma"] = {"achieved": wg["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": wg["tflops"] / peak}
        else:
            roof["bwd_wgrad_roofline"] = hbm
    # every large kernel against its own bound, whichever is the largest this run (forward and weight gradient trade places from
    # box to box): the fused kernels against the dense MFMA peak, the fp8-staged weight gradient against HBM
    roof["per_kernel"] = {k: {"bound": "mfma", "achieved": kern[k]["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": kern[k]["tflops"] / peak if kern[k]["tflops"] else None,
                              "avg_launch_ms": kern[k]["avg_ms"], "ms_per_step": kern[k]["ms_per_step"],
                              "traffic": traffic_rec["kernels"][k]["hbm_bytes_per_launch"] if traffic_rec and k in traffic_rec.get("kernels", {}) else None}
                          for k in ("fwd", "bwd_dgrad", "bwd_wgrad")}
    if prec == "bf16" and (fp8 or bf16_store) and wg["avg_ms"]:
        roof["per_kernel"]["bwd_wgrad"].update({"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                                                "mfma_frac": wg["tflops"] / peak})
    return roof


BWD_MODES = {1: "mode 1: recompute backward (no forward store)", 3: "mode 3: from the forward's f32 store",
             5: "mode 5: from the forward's store (8-bit staged, or bf16 where nothing may be staged in 8 bits), nothing recomputed"}
STORE_NAMES = {0: "none", 1: "f32", 3: "8-bit staged (e4m3 layer inputs)", 4: "bf16 (bf16 layer inputs)"}


def measure(args, prec, stage_fp8, data, dev, rank, world, use_pg, steps, warmup, sustained_steps=0):
    """Warm-up, `steps` timed steps between barriers (max over ranks), then an eager pass that times the kernels with HIP
    events, then (optionally) the sustained run.  `stage_fp8`: None = the planner's default, 0 / 1 = this record's
    trainer runs with that value of NCA_OPT_STAGE_FP8 as ITS planner option (per trainer: nothing process-wide is touched).  Returns
    the record's fields (value, ms_per_step, dtype label, plan, roofline, ...).  A forward store that cannot be allocated is an
    error here (fused.STRICT_STORE): the bench never silently times the recompute path."""
    from nerfca_amd import fused as fused_mod
    strict0 = fused_mod.STRICT_STORE
    fused_mod.STRICT_STORE = True
    try:
        return _measure(args, prec, stage_fp8, data, dev, rank, world, use_pg, steps, warmup, sustained_steps)
    finally:
        fused_mod.STRICT_STORE = strict0


def _measure(args, prec, stage_fp8, data, dev, rank, world, use_pg, steps, warmup, sustained_steps):
    from nerfca_amd import _capi
    from nerfca_amd import fused as fused_mod
    tr = make_trainer(args, prec, data, dev, rank, world, use_pg, plan_opts=None if stage_fp8 is None else {"stage_fp8": stage_fp8})

    def barrier():
        torch.cuda.synchronize()
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    base_iter = 75000    # steady state: half of the frequency bands open
    step = tr.step_graph if args.graph else tr.step
    fallbacks0 = fused_mod.STORE_FALLBACKS
    if args.graph and use_pg and warmup > 0:
        # The sharded step records its RCCL all-reduce INTO the step graph (one replay per step).  That was verified under a one-rank group on
        # one GPU; no multi-GPU node was available to any round.  Should the capture of the collective fail on a real node, fall back -- on every
        # rank alike -- to round 5's structure (two graph segments, the collective issued from the host between them) rather than lose the run.
        try:
            step(base_iter)
        except Exception as e:          # noqa: BLE001
            print(f"bench.py: capturing the all-reduce into the step graph failed ({type(e).__name__}: {e}); falling back to a host-issued collective between two graph segments", file=sys.stderr, flush=True)
            os.environ["NERFCA_GRAPH_COLLECTIVE"] = "0"
            torch.cuda.synchronize()
            tr = make_trainer(args, prec, data, dev, rank, world, use_pg, plan_opts=None if stage_fp8 is None else {"stage_fp8": stage_fp8})
            step = tr.step_graph
    for i in range(warmup):
        step(base_iter + i)
    barrier()
    _capi.timing_reset()
    _capi.timing_enable(not args.graph)
    t0 = time.perf_counter()
    for i in range(steps):
        loss, _, _ = step(base_iter + warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    loss = float(loss)
    plan = tr.plan()            # of this trainer's last backward (graph: of the capture)
    sustained = None
    if sustained_steps >= 200 and args.graph:
        for i in range(sustained_steps - 100):
            step(base_iter + i)
        barrier()
        ts = time.perf_counter()
        for i in range(100):
            step(base_iter + i)
        barrier()
        sustained = {"ms_per_step": (time.perf_counter() - ts) * 10.0, "over": f"the last 100 of {sustained_steps} further graph-replayed steps (back to back after the timed region)",
                     "rays_per_s": rays_per_rank(args, world) * world * 100 / (time.perf_counter() - ts),
                     "in_kernel_clock": "profiles/r03_clock_probe.txt (round 3's diagnostic build that stamps s_memtime / s_memrealtime -- tools/r03_experiments.sh; no stamp executes in this build)"}
    timed_steps, eager_dt = steps, dt
    eager_ms = None
    if args.graph:       # events cannot be recorded inside a replayed graph: time the same kernels eagerly, outside dt
        timed_steps = max(1, min(args.kernel_steps, steps))
        tr.step(base_iter)          # (untimed: the eager step's first pass allocates its own store, and first-touch page mapping shows in the kernels)
        tr.step(base_iter + 1)
        barrier()
        _capi.timing_reset()
        _capi.timing_enable(True)
        te = time.perf_counter()
        for i in range(timed_steps):
            tr.step(base_iter + i)
        barrier()
        eager_dt = time.perf_counter() - te
        eager_ms = eager_dt / timed_steps * 1e3       # the same step with host-launched kernels and torch's Adam
        plan = tr.plan()
    _capi.timing_enable(False)
    if use_pg:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    kern = kernel_table(args, timed_steps, plan, world)
    _capi.timing_reset()
    fp8 = bool(plan.get("stage_fp8"))
    label = "f32" if prec == "f32" else ("bf16+fp8stage" if fp8 else "bf16")
    rec = {"value": rays_per_rank(args, world) * world * steps / dt, "unit": "rays/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "dtype": label,
           "arithmetic": {"f32": "f32 (hidden layers on the bf16 matrix cores from exact 3-way splits): 1e-5 relative vs the reference's f32 path per step (tests/test_hip_parity.py)",
                          "bf16": "bf16 MFMA operands everywhere (forward, dgrad chain AND weight gradient), f32 accumulation, f32 master weights; nothing is staged in 8 bits: "
                                  "the forward leaves layer inputs as bf16 fragments + ReLU masks + raw outputs (the bf16 store), the backward recomputes nothing and writes bf16 "
                                  "output gradients (or, where the store does not fit, recomputes the layers: see plan.backward)",
                          "bf16+fp8stage": "bf16 MFMA operands for the MLP contractions (forward and dgrad), f32 accumulation, f32 master weights; the layer inputs (e4m3) and output "
                                           "gradients (e5m2, per-tile power-of-two scale) cross HBM in 8 bits and the weight gradient contracts them on the MX-fp8 matrix path; PSNR-gated "
                                           "(tests/test_psnr_gates.py)"}[label],
           "hip_graph": bool(args.graph), "eager_ms_per_step": eager_ms, "kernel_table_steps": timed_steps, "final_loss": loss,
           "plan": {"stage_fp8": fp8, "backward": BWD_MODES.get(plan.get("bwd_kernel_mode"), str(plan.get("bwd_kernel_mode"))),
                    "forward_store": STORE_NAMES.get(plan.get("fwd_store_format", 0) & 15, str(plan.get("fwd_store_format"))),
                    "resident_weight_images": {"fwd": bool(plan.get("fwd_resident")), "bwd": bool(plan.get("bwd_resident"))},
                    "ray_chunks": plan.get("chunks"),
                    "wgrad": {"jobs": plan.get("wgrad_jobs"), "splits": plan.get("wgrad_splits"), "splits_rebuild_jobs": plan.get("wgrad_splits_rebuild")},
                    "overlap": {"cus": plan.get("overlap_cus"), "forked": bool(plan.get("overlap_forked"))},
                    "launches_per_step": {k: (kern[k]["launches"] // max(timed_steps, 1)) for k in ("fwd", "bwd_dgrad", "bwd_wgrad")}},
           "roofline": roofline_of(args, prec, kern, eager_dt, plan, dt / steps * 1e3, world),
           "store_fallbacks": fused_mod.STORE_FALLBACKS - fallbacks0}       # > 0: some backward ran on the recompute path (store did not fit)
    if sustained:
        rec["sustained"] = sustained
    tr.check_ray_ids()          # (one device read: a ray id outside the table would have been clamped by nca_prepare_batch -- never silently)
    del tr
    torch.cuda.empty_cache()
    return rec


def quick_step_ms(args, prec, data, dev, rank, world, use_pg, global_rays, steps=30, warmup=5):
    """ms per graph-replayed step of a trainer whose GLOBAL batch is `global_rays` (this rank renders its 1 / world of it), max over ranks."""
    tr = make_trainer(args, prec, data, dev, rank, world, use_pg, global_rays=global_rays)
    tr.always_allreduce = use_pg
    for i in range(warmup):
        tr.step_graph(75000 + i)
    torch.cuda.synchronize()
    if use_pg:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step_graph(75000 + warmup + i)
    torch.cuda.synchronize()
    if use_pg:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_pg and world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    tr.check_ray_ids()
    del tr
    torch.cuda.empty_cache()
    return dt / steps * 1e3


def predicted_scaling(args, data, dev, step_ms_1gpu):
    """What one node's strong-scaled run can be PREDICTED from on a single GPU (no 8-GPU node was available to any round): the
    measured step of ONE rank's share of the global batch, rays / N for N = 2, 4, 8, with everything a rank does per step in place --
    including the step's collective, as a one-rank RCCL process group (two captured graph segments with the all-reduce of the 152 916
    floats between them; on one rank the collective moves no bytes over xGMI, so the links' own time is an estimate, stated apart)."""
    import socket
    import torch.distributed as dist
    out = {"basis": "measured on ONE MI355X: graph-replayed step of a rank's share (rays / N) of the global batch under a one-rank RCCL process group "
                    "(graph segment -> all_reduce(SUM) of the flat gradient + the early-stop pair -> graph segment with the library Adam)",
           "global_rays": args.rays, "step_ms_1gpu_no_collective": step_ms_1gpu}
    own_pg = False
    try:
        if not dist.is_initialized():
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            own_pg = True
        steps = {}
        for n in (1, 2, 4, 8):
            if args.rays % n:
                continue
            steps[n] = quick_step_ms(args, args.prec, data, dev, 0, 1, True, args.rays // n)
        out["step_ms_with_one_rank_collective"] = {str(n): v for n, v in steps.items()}
        out["collective_overhead_ms_1gpu"] = steps[1] - step_ms_1gpu            # graph split + one-rank all-reduce launch, measured
        # 152 916 f32 = 0.61 MB: on 8 ranks a direct reduce-scatter + all-gather moves 7/8 of it each way over 7 links of ~153 GB/s -- ~1 us of
        # bandwidth; the collective is latency-bound (two hops of a few us each plus RCCL's launch), estimated, NOT measured here
        out["xgmi_allreduce_estimate_ms"] = 0.04
        out["predicted_speedup_vs_1gpu"] = {str(n): steps[1] / (steps[n] + (out["xgmi_allreduce_estimate_ms"] if n > 1 else 0.0)) for n in steps}
        out["predicted_rays_per_s"] = {str(n): args.rays / ((steps[n] + (out["xgmi_allreduce_estimate_ms"] if n > 1 else 0.0)) * 1e-3) for n in steps}
        out["note"] = ("a rank's kernels see rays / N rays: at N = 8 that is 8 192 rays x 192 samples = 24 576 wave tiles, 12 per wave of the persistent grids "
                       "(resident kernels: threshold 16 384); what does not shrink with N is the per-step fixed part -- batch preparation, weight packing, "
                       "loss finish, reductions over the split slabs, Adam: ~0.15 ms -- and the collective")
    except Exception as e:          # (a box without RCCL for one rank: the record says so instead of failing the bench line)
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        if own_pg:
            dist.destroy_process_group()
    return out


def general_kernels_record():
    """Nets beyond the fused kernels' range (more than 128 units per layer): a composite render of 8 192 rays x 192 samples with two
    256-unit nets on the general f32 kernels, forward and forward + backward (tools/wide_bench.py).  In the full record only."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wide_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "wide_bench.py"))
    wb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wb)
    rec = wb.measure([256], 8192, 192, 3)[0]
    rec["note"] = ("f32 GEMM per layer on v_mfma_f32_32x32x2_f32, activations in HBM; frac = algorithmic FLOPs (forward + backward = 3 x forward; the backward runs from the forward's store) / time / 157.3 TFLOP/s; "
                   "not part of the headline (BASELINE configs use 128 units)")
    return rec


def configs3_record(args, dev):
    """BASELINE configs[3] -- "MAGIX 4D phantom, 8 angiogram sequences, 512^2 x 256 samples, fp32, 1 x MI355X" -- as a short leg of the
    default line, so that it is timed by whoever runs the bench: synthetic data of that shape (MAGIX cone beam DSD 2000 / DSO 600 mm,
    8 views x 10 phases of 512^2, 256 samples per ray), the f32 parity mode, one full detector (262 144 rays) per graph-replayed step.
    The forward store of the whole batch would be ~380 GB: the step runs as ray micro-batches under the 96 GB store limit."""
    import argparse
    from nerfca_amd import synthetic
    a3 = argparse.Namespace(**vars(args))
    a3.det, a3.samples, a3.rays, a3.views, a3.prec, a3.kernel_steps, a3.scaling = 512, 256, 262144, 8, "f32", 1, "strong"
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    data = synthetic.make_dataset(a3.det, a3.samples, dev, views=synthetic.TRAIN_VIEWS_8, geometry="magix")
    build_s = time.perf_counter() - t0
    rec = measure(a3, "f32", None, data, dev, 0, 1, False, args.configs3_steps, 1)
    out = {"workload": "BASELINE configs[3]: MAGIX geometry, 8 views x 10 phases of 512^2, 256 samples/ray, 262 144 rays/step (one full detector), f32, fwd+losses+bwd+Adam",
           "value": rec["value"], "unit": "rays/s", "ms_per_step": rec["ms_per_step"], "steps": rec["steps"], "warmup": rec["warmup"], "dtype": rec["dtype"],
           "hip_graph": rec["hip_graph"], "final_loss": rec["final_loss"], "plan": rec["plan"], "roofline_step": rec["roofline"]["step"],
           "per_kernel": rec["roofline"]["per_kernel"], "store_fallbacks": rec["store_fallbacks"], "dataset_build_s": build_s,
           "ray_table_rows": int(data.rays_train.shape[0])}
    del data
    torch.cuda.empty_cache()
    return out


def latency_record(args, dev):
    """BASELINE.md 3.3's second batch size: the reference's DEFAULT batch, 1 024 rays x 500 samples per step (train/composite.txt:25,40) --
    0.45 TFLOP per step, where launch count and per-step fixed work matter as much as the kernels.  Graph-replayed step, bf16 (the
    planner's default: 8-bit staged store, streaming kernels below the resident threshold) and f32."""
    import argparse
    from nerfca_amd import synthetic
    a = argparse.Namespace(**vars(args))
    a.rays, a.samples, a.scaling = 1024, 500, "strong"
    data = synthetic.make_dataset(a.det, a.samples, dev, views=synthetic.TRAIN_VIEWS)
    out = {"workload": "run_composite defaults: 1 024 rays x 500 samples per step (train/composite.txt:25,40), 256^2 detector, 4 views x 10 phases, fwd+losses+bwd+Adam, graph-replayed",
           "rays_per_step": a.rays, "samples_per_ray": a.samples, "steps": args.latency_steps}
    for prec in ("bf16", "f32"):
        ms = quick_step_ms(a, prec, data, dev, 0, 1, False, a.rays, steps=args.latency_steps, warmup=20)
        out[prec] = {"ms_per_step": ms, "rays_per_s": a.rays / (ms * 1e-3), "step_tflops": FLOP_STEP * a.rays * a.samples / (ms * 1e-3) / 1e12}
    del data
    torch.cuda.empty_cache()
    return out


def psnr_record(args, dev):
    """Held-out-view PSNR after `--psnr-steps` steps from identical initial weights, ray batches and depth jitter: HIP f32,
    HIP bf16 and the CPU oracle (reference-equivalent torch ops).  64^2 detector x --samples, 256 rays per step: a size the
    oracle affords inside a bench run.  All three parameter sets are evaluated by the same (HIP f32) renderer."""
    import nerfca_amd
    from oracle import nerfca_oracle as O
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    S, R, steps, det = args.samples, 256, args.psnr_steps, 64
    data = synthetic.make_dataset(det, S, dev, views=synthetic.TRAIN_VIEWS, n_phases=10, F=64)
    cfg = TrainConfig(depth_samples_per_ray_coarse=S, img_sample_size=R, static_pos_enc_window_decay_steps=steps,
                      temp_pos_enc_window_decay_steps=steps, lr_decay_steps=steps)

    def fresh(prec):
        torch.manual_seed(1)
        sdef, tdef = synthetic.net_definitions(dev)
        s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
        nerfca_amd.set_precision(prec, s, t)
        return CompositeTrainer(cfg, s, t, data, dev, seed=0)

    def psnr_of(tr, it):
        e = tr.evaluate(it)
        return {"psnr_mse_db": float(e["test_psnr_mse"]), "test_psnr_db": float(e["test_psnr"])}

    out = {}
    for prec in ("f32", "bf16"):
        tr = fresh(prec)
        tr.update_windows(0)
        if prec == "f32":
            out["untrained"] = psnr_of(tr, 0)
        for it in range(steps):
            tr.step(it)
        tr.update_windows(steps)
        out["hip_" + prec] = psnr_of(tr, steps)
        if prec == "f32":
            out["_hip_f32_params"] = torch.cat([p.detach().flatten().cpu() for p in list(tr.t.parameters()) + list(tr.s.parameters())])
    # the oracle on the host cores, fed the SAME ray ids and jitter draws
    tr = fresh("f32")
    ss, sd = O.NetSpec(num_filters=128), O.NetSpec(num_filters=128, num_time_dim=8)
    ps = {k: v.detach().cpu().clone() for k, v in tr.s.state_dict().items()}
    pd = {k: v.detach().cpu().clone() for k, v in tr.t.state_dict().items()}
    torch.set_num_threads(host_cores())
    ot = O.OracleTrainer(ps, ss, pd, sd, lr=cfg.lr, lr_end_factor=cfg.lr_end_factor, lr_decay_steps=steps, window_decay_steps=steps)
    table, phases = data.rays_train.cpu(), data.phases_train.cpu()
    I0 = torch.full((R,), float(data.geo["max_pixel_value"]))
    z0 = tr.depth.cpu()
    t0 = time.perf_counter()
    for it in range(steps):
        ids = tr.draw_ray_ids_device(it).cpu()
        rays, ph = table.index_select(0, ids), phases.index_select(0, ids)
        zj = O.stratified_depths(z0, tr.draw_jitter(it).cpu())
        ot.step(it, rays[:, 0, :], rays[:, 1, :], ph[:, None].repeat(1, S), I0, zj, rays[:, 2, 0], rays[:, 3, 0])
    cpu_s = time.perf_counter() - t0
    hip_f32 = out.pop("_hip_f32_params")
    cpu = torch.cat([v.detach().flatten() for v in list(ot.pd.values()) + list(ot.ps.values())])
    out["max_param_diff_hip_f32_vs_oracle"] = float((hip_f32 - cpu).abs().max() / cpu.abs().max())
    tr.s.load_state_dict({k: v.detach() for k, v in ot.ps.items()})
    tr.t.load_state_dict({k: v.detach() for k, v in ot.pd.items()})
    tr.update_windows(steps)
    out["cpu_oracle"] = psnr_of(tr, steps)
    out["cpu_oracle"]["wall_s"] = cpu_s
    out["config"] = f"{det}^2 detector x {S} samples/ray, 4 views x 10 phases + 1 held-out view, {R} rays/step, {steps} steps, schedules compressed to the run"
    out["gap_f32_vs_oracle_db"] = out["hip_f32"]["psnr_mse_db"] - out["cpu_oracle"]["psnr_mse_db"]
    out["gap_bf16_vs_f32_db"] = out["hip_bf16"]["psnr_mse_db"] - out["hip_f32"]["psnr_mse_db"]
    return out


def _r(x, sig=6):
    """Floats of the compact line with `sig` significant digits (the full record keeps every digit)."""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _kernel_fracs(roof):
    """{kernel: {bound, mfma_frac | hbm_frac, avg_launch_ms}}: ONE key per denominator -- `mfma_frac` divides algorithmic FLOP/s by the dense
    MFMA peak of the pipe the contractions run on, `hbm_frac` divides algorithmic bytes/s by the 8 TB/s HBM peak; never one under the other's name."""
    out = {}
    for k, v in (roof.get("per_kernel") or {}).items():
        e = {"bound": v["bound"], "avg_launch_ms": v["avg_launch_ms"]}
        if v["bound"] == "hbm":
            e["hbm_frac"] = v["frac"]
            e["mfma_frac"] = v.get("mfma_frac")
        else:
            e["mfma_frac"] = v["frac"]
        out[k] = e
    return out


def precision_summary(rec):
    """Three numbers per precision for the compact line: throughput, step time, and the whole step's fraction of the MFMA peak."""
    return {"rays_per_s": rec["value"], "ms_per_step": rec["ms_per_step"], "step_mfma_frac": rec["roofline"]["step"]["frac"]}


def write_full_record(args, out):
    """The full record (every sub-record, every kernel table) goes to a FILE; stdout ends with the compact line the driver parses."""
    path = args.full_record or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f)
            f.write("\n")
        return os.path.relpath(path, ROOT)
    except OSError as e:          # (a read-only tree: the compact line still goes out)
        print(f"bench.py: full record not written ({e})", file=sys.stderr)
        return None


HEADLINE_LIMIT_BYTES = 4000


def headline_line(out, full_path=None):
    """The LAST stdout line: the bench contract's keys plus `roofline` and `cpu_baseline`, under HEADLINE_LIMIT_BYTES (round 5's 24.6 KB line was
    not parsed by the driver).  Everything else is in the full record (`full_record`).  tests/test_host_cpu.py checks size, keys and json.loads."""
    cfg = out["config"]
    roof = out["roofline"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg["workload"], "rays_per_step_per_gpu": cfg["rays_per_step_per_gpu"], "global_rays_per_step": cfg["global_rays_per_step"],
                      "samples_per_ray": cfg["samples_per_ray"], "parallelism": cfg["parallelism"], "hip_graph": cfg["hip_graph"], "stage_fp8": cfg["stage_fp8"],
                      "launches_per_step": cfg["launches_per_step"]}
    line["roofline"] = {"bound": roof["bound"], "kernel": roof["kernel"], "achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"], "frac": roof["frac"],
                        "traffic": roof.get("traffic"), "avg_launch_ms": roof["avg_launch_ms"], "launches": roof["launches"],
                        "step": {k: roof["step"][k] for k in ("achieved", "peak", "unit", "frac")}, "kernels": _kernel_fracs(roof)}
    if roof.get("traffic_source"):
        line["roofline"]["traffic_source"] = roof["traffic_source"]["file"]
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {k: out["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample")}
    b = out.get("baseline_config_dtype")
    if b:
        line["baseline_config_dtype"] = {"dtype": b["dtype"], "value": b["value"], "unit": b["unit"], "ms_per_step": b["ms_per_step"], "steps": b["steps"],
                                         "step_mfma_frac": b["roofline_step"]["frac"],
                                         "kernels": {k: {kk: vv for kk, vv in v.items() if kk != "avg_launch_ms"} for k, v in _kernel_fracs({"per_kernel": b["per_kernel"]}).items()}}
    if out.get("precisions"):
        line["precisions"] = out["precisions"]
    lat = out.get("latency_regime")
    if lat:
        line["latency_regime"] = {"rays_per_step": lat["rays_per_step"], "samples_per_ray": lat["samples_per_ray"], "steps": lat["steps"],
                                  **{p: {"ms_per_step": lat[p]["ms_per_step"], "rays_per_s": lat[p]["rays_per_s"]} for p in ("bf16", "f32") if p in lat}}
    c3 = out.get("configs3")
    if c3:
        line["configs3"] = {"dtype": c3["dtype"], "rays_per_step": 262144, "samples_per_ray": 256, "value": c3["value"], "unit": c3["unit"], "ms_per_step": c3["ms_per_step"],
                            "steps": c3["steps"], "step_mfma_frac": c3["roofline_step"]["frac"]}
    for k in ("vs_unfused_gpu", "rccl_ranks", "store_fallbacks"):
        if out.get(k) is not None:
            line[k] = out[k]
    for k in ("weak_scaling", "strong_scaling"):
        if k in out:
            line[k] = {kk: out[k][kk] for kk in ("scaling", "ms_per_step", "value", "unit", "global_rays_per_step", "rays_per_step_per_gpu")}
    if out.get("sustained"):
        line["sustained_ms_per_step"] = out["sustained"]["ms_per_step"]
    line["full_record"] = full_path
    text = json.dumps(_r(line), separators=(",", ":"))
    if len(text) > HEADLINE_LIMIT_BYTES:          # (never let the line outgrow the driver's parser again: shed the optional parts, largest first)
        for k in ("baseline_config_dtype", "latency_regime", "configs3", "precisions", "weak_scaling", "strong_scaling"):
            if k in line:
                line[k] = "see full_record"
                text = json.dumps(_r(line), separators=(",", ":"))
                if len(text) <= HEADLINE_LIMIT_BYTES:
                    break
    return text


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before the first HIP call of this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: start one rank per GPU (or run `python bench.py --gpus N`, which does)")
    if not (0 <= rank < world and 0 <= local < world):
        raise SystemExit(f"RANK={rank} / LOCAL_RANK={local} outside a world of {world}")
    if args.dry_run:       # nothing below this line has run: no HIP call, no process group
        # (ONE write per rank, newline included: eight ranks share the launcher's pipe, and print()'s separate newline let two records land on one line)
        os.write(1, (json.dumps({"rank": rank, "world": world, "local_rank": local, "device": f"cuda:{local}", "backend": "nccl" if world > 1 else None,
                                 "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}", "scaling": args.scaling,
                                 "rays_per_rank": rays_per_rank(args, world), "global_rays_per_step": rays_per_rank(args, world) * world}) + "\n").encode())
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or os.environ.get("NERFCA_FORCE_PG") == "1"   # the env switch lets a 1-GPU box exercise RCCL
    rccl_ranks = None
    if use_pg:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        rccl_ranks = dist.get_world_size()
        if rccl_ranks != world or dist.get_rank() != rank:
            raise SystemExit(f"process group reports rank {dist.get_rank()} of {rccl_ranks}, the environment said {rank} of {world}")
        # one eager collective before anything is captured: the communicator exists when the step graph records its all-reduce
        warm = torch.zeros(8, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()

    from nerfca_amd import _capi, synthetic
    _capi.lib()   # fail loudly if the HIP library is missing

    views = synthetic.TRAIN_VIEWS if args.views == 4 else synthetic.TRAIN_VIEWS_8[: args.views]
    data = synthetic.make_dataset(args.det, args.samples, dev, views=views)
    main_rec = measure(args, args.prec, None, data, dev, rank, world, use_pg, args.steps, args.warmup,
                       sustained_steps=args.sustained_steps if (world == 1 and not args.no_extras) else 0)
    other_rec = None
    if (world > 1 or os.environ.get("NERFCA_FORCE_PG") == "1") and not args.no_extras:          # (the env switch: the multi-rank legs on a one-GPU box)
        # the other scaling mode in the same run, every rank taking part: strong = one global batch of --rays split N ways, weak = --rays per rank
        other = "weak" if args.scaling == "strong" else "strong"
        g_rays = args.rays * world if other == "weak" else args.rays
        if g_rays % world == 0:
            ms = quick_step_ms(args, args.prec, data, dev, rank, world, use_pg, g_rays)
            other_rec = {"scaling": other, "ms_per_step": ms, "value": g_rays / (ms * 1e-3), "unit": "rays/s", "global_rays_per_step": g_rays,
                         "rays_per_step_per_gpu": g_rays // world, "steps": 30, "warmup": 5}

    if rank == 0:
        out = {"metric": f"training rays/sec ({args.det}^2 det, {args.samples} samples/ray)", "value": main_rec["value"], "unit": "rays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_rec["ms_per_step"],
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": main_rec["dtype"], "data": "synthetic",
               "config": {"workload": f"run_composite XCAT {args.views}-view x 10 phases, {args.det}^2 detector x {args.samples} samples/ray, "
                                      f"{rays_per_rank(args, world)} rays/step/GPU ({'one full detector' if args.scaling == 'weak' else f'a global batch of {args.rays} rays split {world} ways'}), "
                                      f"F=128 x 4 hidden layers x 2 nets, L=12, fwd+losses+bwd+Adam",
                          "rays_per_step_per_gpu": rays_per_rank(args, world), "global_rays_per_step": rays_per_rank(args, world) * world, "samples_per_ray": args.samples, "parallelism": f"ray-sharded dp{world}", "hip_graph": bool(args.graph),
                          "arithmetic": main_rec["arithmetic"], "stage_fp8": main_rec["plan"]["stage_fp8"], "backward": main_rec["plan"]["backward"],
                          "launches_per_step": main_rec["plan"]["launches_per_step"],
                          "plan": main_rec["plan"], "library": _capi.build_info()},
               "rccl_ranks": rccl_ranks, "roofline": main_rec["roofline"], "final_loss": main_rec["final_loss"], "eager_ms_per_step": main_rec["eager_ms_per_step"],
               "kernel_table_steps": main_rec["kernel_table_steps"], "store_fallbacks": main_rec["store_fallbacks"]}
        if "sustained" in main_rec:
            out["sustained"] = main_rec["sustained"]
        if other_rec is not None:
            out[other_rec["scaling"] + "_scaling"] = other_rec
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, data)
        if world == 1 and not args.no_extras and args.prec == "bf16" and args.predicted_scaling:
            out["predicted_scaling"] = predicted_scaling(args, data, dev, main_rec["ms_per_step"])
        if world == 1 and not args.no_extras:
            if args.unfused_gpu_rays > 0:
                out["unfused_gpu_baseline"] = unfused_gpu_baseline(args, data)
                out["vs_unfused_gpu"] = out["value"] / out["unfused_gpu_baseline"]["value"]
            # the other precisions side by side, each through the same (graph-replayed) step
            if args.prec == "bf16" and args.pure_steps > 0 and main_rec["plan"]["stage_fp8"]:
                out["bf16_pure"] = measure(args, "bf16", 0, data, dev, 0, 1, False, args.pure_steps, args.warmup)
                out["bf16_pure"]["note"] = ("BASELINE configs[1] as written: bf16 operands in every contraction, nothing staged in 8 bits (stage_fp8 = 0: the bf16 store of round 5 -- "
                                            "NCA_STORE_BF16 -- and the mode-5 backward from it; DESIGN.md 4.7)")
                bp = out["bf16_pure"]
                # BASELINE.json configs[1] says "bf16": the record of the arithmetic that is bf16 and nothing narrower, at the top level beside
                # the headline (whose `dtype` says "bf16+fp8stage": PSNR-gated, but 8-bit on its way to the weight-gradient kernel)
                out["baseline_config_dtype"] = {"dtype": "bf16", "config": "BASELINE.json configs[1] as written (run_composite XCAT 4-view, 256^2 x 192 samples, bf16, 1 x MI355X)",
                                                "value": bp["value"], "unit": "rays/s", "ms_per_step": bp["ms_per_step"], "steps": bp["steps"], "plan": bp["plan"],
                                                "roofline": {k: bp["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "mfma", "traffic", "traffic_source", "avg_launch_ms")},
                                                "roofline_step": bp["roofline"]["step"], "per_kernel": bp["roofline"]["per_kernel"], "record": "bf16_pure"}
            if args.prec != "f32" and args.f32_steps > 0:
                out["f32"] = measure(args, "f32", None, data, dev, 0, 1, False, args.f32_steps, args.f32_warmup)
            out["precisions"] = {k: precision_summary(r) for k, r in (("f32", out.get("f32")), ("bf16", out.get("bf16_pure")), (main_rec["dtype"], main_rec)) if r}
            if args.psnr_steps > 0:
                out["psnr"] = psnr_record(args, dev)
            if args.configs3_steps > 0 and args.prec == "bf16":
                out["configs3"] = configs3_record(args, dev)
            if args.latency_steps > 0 and args.prec == "bf16":
                out["latency_regime"] = latency_record(args, dev)
            if args.prec == "bf16":
                out["general_kernels"] = general_kernels_record()
        path = write_full_record(args, out)
        print(headline_line(out, path), flush=True)
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
