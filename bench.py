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
the timed region starts.  Weak scaling: every rank renders --rays rays per step.

Rank 0 prints ONE JSON line.  `roofline` is for the kernel with the largest share of the timed
region, measured with HIP events on the launch stream inside the library (nca_timing_*);
`cpu_baseline` is the CPU oracle (reference-equivalent torch CPU ops) on a bounded sample.  At N = 1 the line
also carries `f32` (the same step in the 1e-5 parity mode, a few steps), `unfused_gpu_baseline` (the
reference-equivalent torch ops run op by op on the same GPU: the denominator of the north star's >= 10x) and
`psnr` (held-out PSNR of HIP f32, HIP bf16 and the CPU oracle after equal steps from identical weights and
batches; train/run_composite.py:391 defines test_psnr).  --no-extras drops those three.
"""
import argparse
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
PEAK_F32_ON_BF16_PIPE = 2500.0 / 6.0
PROFILE_TAGS = ("r02", "r01")          # committed PMC summaries, newest first


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
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
    ap.add_argument("--f32-steps", type=int, default=4)
    ap.add_argument("--psnr-steps", type=int, default=100, help="steps of the PSNR record (64^2 detector, 256 rays/step; 0: skip)")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--torch-losses", action="store_true", help="losses + autograd in torch ops instead of the fused loss kernel")
    args = ap.parse_args(argv)
    args.graph = not args.eager            # the whole step as one captured HIP graph unless --eager
    return args


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
    return subprocess.run(cmd, env=env).returncode


def committed_pmc(prec, rays, samples):
    """The committed PMC summaries of this configuration, newest round first: (traffic record, sq record, source) or Nones.
    These are REPLAYED figures (separate rocprofv3 --pmc passes cannot run inside a bench run): the line labels them with
    the file and commit they come from."""
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, "profiles", f"{tag}_{prec}_pmc_traffic.json")
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        cfg = rec.get("config", {})
        if cfg.get("rays_per_step") != rays or cfg.get("samples_per_ray") != samples or cfg.get("prec") != prec:
            continue
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{prec}_pmc_sq.json")))
        except (OSError, ValueError):
            sq = None
        return rec, sq, {"file": f"profiles/{tag}_{prec}_pmc_traffic.json", "taken_at_commit": rec.get("commit"), "kind": "replayed from a committed rocprofv3 --pmc pass"}
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


def make_trainer(args, prec, data, dev, rank, world, use_pg):
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    torch.manual_seed(1)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    nerfca_amd.set_precision(prec, s, t)
    cfg = TrainConfig(depth_samples_per_ray_coarse=args.samples, img_sample_size=args.rays * world)
    tr = CompositeTrainer(cfg, s, t, data, dev, rank=rank, world=world, seed=0, fused_loss=not args.torch_losses)
    tr.always_allreduce = use_pg
    return tr


def kernel_table(args, prec, timed_steps):
    """Per-kernel HIP-event times of the span the library's timers covered, with algorithmic TFLOP/s."""
    from nerfca_amd import _capi
    n_samp = args.rays * args.samples * timed_steps
    kern = {}
    # bf16 with BF16 staging at this size (NCA_STAGE_FP8=0): the weight gradient of the last hidden layer of both nets (2 x 2 x 128 x
    # 128 FLOP per sample) is accumulated inside the dgrad kernel (mode 4), not by the wgrad kernel.  The default, fp8 staging,
    # runs mode 5 (nothing recomputed; also one dgrad launch per net when the weight images are resident) and all of it in the wgrad
    dg_ms, dg_n = _capi.timing_read("bwd_dgrad")
    wg_ms, wg_n = _capi.timing_read("bwd_wgrad")
    fp8 = prec == "bf16" and _capi.get_option(_capi.OPT_STAGE_FP8) != 0
    onchip = prec == "bf16" and not fp8 and wg_n > 0 and dg_n == 2 * wg_n
    moved = 2 * 2 * 128 * 128 if onchip else 0
    for name, flop in (("fwd", FLOP_FWD), ("bwd_dgrad", FLOP_DGRAD + moved), ("bwd_wgrad", FLOP_WGRAD - moved), ("bwd_reduce", 0), ("loss", 0), ("pack", 0)):
        ms, n = _capi.timing_read(name)
        kern[name] = {"ms_total": ms, "launches": n, "avg_ms": ms / n if n else None,
                      "tflops": (flop * n_samp / (ms * 1e-3) / 1e12) if ms > 0 and flop else None}
    return kern, onchip


def roofline_of(args, prec, kern, dt):
    dom = max(("fwd", "bwd_dgrad", "bwd_wgrad"), key=lambda k: kern[k]["ms_total"])
    peak = PEAK_TFLOPS[prec]
    traffic_rec, sq_rec, source = committed_pmc(prec, args.rays, args.samples)
    traffic = traffic_rec["kernels"][dom]["hbm_bytes_per_launch"] if traffic_rec and dom in traffic_rec.get("kernels", {}) else None
    issue = None
    if sq_rec and dom in sq_rec.get("kernels", {}):
        k = sq_rec["kernels"][dom]
        issue = {"mfma_busy": k["mfma_busy"], "valu_busy": k.get("valu_busy")}
    roof = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
            "frac": kern[dom]["tflops"] / peak if kern[dom]["tflops"] else None,
            "traffic": traffic, "traffic_source": source if traffic is not None else None,
            "avg_launch_ms": kern[dom]["avg_ms"], "launches": kern[dom]["launches"],
            # the staged (non-algorithmic) HBM traffic of that kernel per second of its run time: what it is bound by in practice
            "staging_TBps": (traffic / (kern[dom]["avg_ms"] * 1e-3) / 1e12) if traffic and kern[dom]["avg_ms"] else None,
            "simd_issue_share_pmc": issue,
            "kernel_time_share": kern[dom]["ms_total"] / (dt * 1e3) if dt else None, "all_kernels": kern}
    from nerfca_amd import _capi
    wg = kern["bwd_wgrad"]
    if prec == "bf16" and _capi.get_option(_capi.OPT_STAGE_FP8) != 0 and wg["avg_ms"]:
        # the weight-gradient kernel's own roofline is the HBM one (see WGRAD_FP8_BYTES_PER_SAMPLE)
        nbytes = WGRAD_FP8_BYTES_PER_SAMPLE * args.rays * args.samples
        gbps = nbytes / (wg["avg_ms"] * 1e-3) / 1e9
        hbm = {"bound": "hbm", "kernel": "bwd_wgrad", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
               "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": wg["avg_ms"],
               "traffic": traffic_rec["kernels"]["bwd_wgrad"]["hbm_bytes_per_launch"] if traffic_rec and "bwd_wgrad" in traffic_rec.get("kernels", {}) else None}
        if dom == "bwd_wgrad":
            roof.update({k: hbm[k] for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")})
            roof["mfma"] = {"achieved": wg["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": wg["tflops"] / peak}
        else:
            roof["bwd_wgrad_roofline"] = hbm
    if prec == "f32":      # the pipe the f32 mode's hidden-layer contractions really run on
        roof["frac_of_bf16_pipe_div_6"] = kern[dom]["tflops"] / PEAK_F32_ON_BF16_PIPE if kern[dom]["tflops"] else None
    return roof


def f32_record(args, data, dev):
    """The same step in the parity mode (1e-5 vs the reference per step): a few steps, own kernel table and roofline."""
    from nerfca_amd import _capi
    tr = make_trainer(args, "f32", data, dev, 0, 1, False)
    tr.step(75000)
    torch.cuda.synchronize()
    _capi.timing_reset()
    _capi.timing_enable(True)
    t0 = time.perf_counter()
    for i in range(args.f32_steps):
        tr.step(75001 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern, _ = kernel_table(args, "f32", args.f32_steps)
    _capi.timing_enable(False)
    _capi.timing_reset()
    return {"value": args.rays * args.f32_steps / dt, "unit": "rays/s", "steps": args.f32_steps, "warmup": 1, "ms_per_step": dt / args.f32_steps * 1e3,
            "dtype": "f32", "parity": "1e-5 relative vs the reference's f32 path per step (tests/test_hip_parity.py)",
            "roofline": roofline_of(args, "f32", kern, dt)}


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
        zj = O.stratified_depths(z0, tr.draw_jitter(it))
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
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or os.environ.get("NERFCA_FORCE_PG") == "1"   # the env switch lets a 1-GPU box exercise RCCL
    rccl_ranks = None
    if use_pg:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        rccl_ranks = dist.get_world_size()

    from nerfca_amd import _capi, synthetic
    from nerfca_amd import fused as fused_mod
    _capi.lib()   # fail loudly if the HIP library is missing

    views = synthetic.TRAIN_VIEWS if args.views == 4 else synthetic.TRAIN_VIEWS_8[: args.views]
    data = synthetic.make_dataset(args.det, args.samples, dev, views=views)
    tr = make_trainer(args, args.prec, data, dev, rank, world, use_pg)

    def barrier():
        torch.cuda.synchronize()
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    base_iter = 75000    # steady state: half of the frequency bands open
    step = tr.step_graph if args.graph else tr.step
    for i in range(args.warmup):
        step(base_iter + i)
    barrier()
    _capi.timing_reset()
    _capi.timing_enable(not args.graph)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _, _ = step(base_iter + args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    loss = float(loss)
    timed_steps = args.steps
    eager_ms = None
    if args.graph:       # events cannot be recorded inside a replayed graph: time the same kernels eagerly, outside dt
        timed_steps = min(args.steps, 4)
        tr.step(base_iter)          # (untimed: the eager step's first pass allocates its own store, and first-touch page mapping shows in the kernels)
        barrier()
        _capi.timing_reset()
        _capi.timing_enable(True)
        te = time.perf_counter()
        for i in range(timed_steps):
            tr.step(base_iter + i)
        barrier()
        eager_ms = (time.perf_counter() - te) / timed_steps * 1e3       # the same step with host-launched kernels and torch's Adam
    _capi.timing_enable(False)
    if use_pg:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        kern, onchip = kernel_table(args, args.prec, timed_steps)
        roof = roofline_of(args, args.prec, kern, dt)
        out = {"metric": f"training rays/sec ({args.det}^2 det, {args.samples} samples/ray)", "value": args.rays * world * args.steps / dt, "unit": "rays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
               "config": {"workload": f"run_composite XCAT {args.views}-view x 10 phases, {args.det}^2 detector x {args.samples} samples/ray, "
                                      f"{args.rays} rays/step/GPU (one full detector), F=128 x 4 hidden layers x 2 nets, L=12, fwd+losses+bwd+Adam",
                          "rays_per_step_per_gpu": args.rays, "samples_per_ray": args.samples, "parallelism": f"ray-sharded dp{world}", "hip_graph": bool(args.graph),
                          "stage_fp8": bool(args.prec == "bf16" and _capi.get_option(_capi.OPT_STAGE_FP8) != 0),
                          "backward": ("mode 5: from the forward's fp8-staged store, nothing recomputed" if args.prec == "bf16" and _capi.get_option(_capi.OPT_STAGE_FP8) != 0
                                       else ("mode 4: from the store, last hidden layer's wgrad on chip" if onchip else "from the store / recompute")),
                          "launches_per_step": {k: (kern[k]["launches"] // max(timed_steps, 1)) for k in ("fwd", "bwd_dgrad", "bwd_wgrad")},
                          "onchip_last_layer_wgrad": bool(onchip)},
               "rccl_ranks": rccl_ranks, "roofline": roof, "final_loss": float(loss), "eager_ms_per_step": eager_ms,
               "store_fallbacks": fused_mod.STORE_FALLBACKS}       # > 0: some backward ran on the recompute path (store did not fit)
        del tr
        torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, data)
        if world == 1 and not args.no_extras:
            if args.unfused_gpu_rays > 0:
                out["unfused_gpu_baseline"] = unfused_gpu_baseline(args, data)
                out["vs_unfused_gpu"] = out["value"] / out["unfused_gpu_baseline"]["value"]
            if args.prec != "f32" and args.f32_steps > 0:
                out["f32"] = f32_record(args, data, dev)
            if args.psnr_steps > 0:
                out["psnr"] = psnr_record(args, dev)
        print(json.dumps(out))
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
