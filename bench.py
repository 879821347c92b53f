#!/usr/bin/env python3
"""Headline benchmark: training rays/sec of the composite NeRF-CA step on synthetic
256^2-detector x 192-samples/ray batches (BASELINE.json metric, configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: ray gather (GPU-resident table), fused
forward, all losses, fused backward, (all-reduce), Adam + LinearLR.  Inputs are resident in HBM when
the timed region starts.  Weak scaling: every rank renders --rays rays per step.

Rank 0 prints ONE JSON line.  `roofline` is for the kernel with the largest share of the timed
region, measured with HIP events on the launch stream inside the library (nca_timing_*);
`cpu_baseline` is the CPU oracle (reference-equivalent torch CPU ops) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work per sample of the default nets (recompute NOT counted)
FLOP_FWD, FLOP_DGRAD, FLOP_WGRAD = 303104, 264704, 303104
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}       # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def parse():
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
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (library Adam+LinearLR); "
                    "per-kernel timings then come from a short eager pass after the timed region")
    ap.add_argument("--unfused-gpu-rays", type=int, default=0, help="also time the unfused torch path on cuda:0 with this many rays/step")
    ap.add_argument("--unfused-gpu-steps", type=int, default=5)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--torch-losses", action="store_true", help="losses + autograd in torch ops instead of the fused loss kernel")
    return ap.parse_args()


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


def measured_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_*_pmc_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE doubled as the
    gfx950 guide prescribes).  Only reported for the exact configuration those passes were taken on."""
    path = os.path.join(ROOT, "profiles", f"r01_{args.prec}_pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    cfg = rec.get("config", {})
    if cfg.get("rays_per_step") != args.rays or cfg.get("samples_per_ray") != args.samples or cfg.get("prec") != args.prec:
        return None
    k = rec.get("kernels", {}).get(kernel)
    return k["hbm_bytes_per_launch"] if k else None


def measured_issue_share(args, kernel):
    """(mfma_busy, valu_busy) of `kernel` from the committed SQ counter pass (profiles/r01_*_pmc_sq.json): the shares of
    SIMD time spent in MFMA and in other vector-ALU instructions -- they do not overlap on gfx950, so
    their sum is the issue-side utilisation.  Same configuration gate as the traffic figure."""
    if measured_traffic(args, kernel) is None:
        return None
    try:
        k = json.load(open(os.path.join(ROOT, "profiles", f"r01_{args.prec}_pmc_sq.json")))["kernels"][kernel]
        return {"mfma_busy": k["mfma_busy"], "valu_busy": k.get("valu_busy")}
    except (OSError, ValueError, KeyError):
        return None


def _oracle_leg(args, data, R, steps, device):
    """`steps` reference-equivalent training steps (the oracle: plain torch ops + autograd + torch Adam) of R rays on
    `device`; returns seconds.  On cuda this is the unfused PyTorch-ROCm path of SURVEY.md 8(d)(ii)."""
    from oracle import nerfca_oracle as O
    S = args.samples
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


def cpu_baseline(args, data, cfg_kwargs):
    """Reference-equivalent CPU path (oracle) on a bounded sample of the same workload."""
    cores = host_cores()
    torch.set_num_threads(cores)
    R, S = args.cpu_rays, args.samples
    dt = _oracle_leg(args, data, R, args.cpu_steps, "cpu")
    return {"value": R * args.cpu_steps / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} full training steps of {R} rays x {S} samples (same nets, losses, Adam) with torch CPU ops, "
                      f"{dt:.1f} s"}


def unfused_gpu_baseline(args, data):
    """Opt-in (--unfused-gpu-rays): the same reference-equivalent torch path run op by op on cuda:0 in f32 — what
    the reference itself does on a GPU.  The chunk loop of the reference is not needed for memory on 288 GB."""
    R, steps = args.unfused_gpu_rays, args.unfused_gpu_steps
    dt = _oracle_leg(args, data, R, steps, "cuda:0")
    return {"value": R * steps / dt, "unit": "rays/s", "dtype": "f32",
            "sample": f"{steps} full training steps of {R} rays x {args.samples} samples with unfused PyTorch-ROCm ops, {dt:.2f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or os.environ.get("NERFCA_FORCE_PG") == "1"   # the env switch lets a 1-GPU box exercise RCCL
    if use_pg:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import nerfca_amd
    from nerfca_amd import _capi, synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    _capi.lib()   # fail loudly if the HIP library is missing

    views = synthetic.TRAIN_VIEWS if args.views == 4 else synthetic.TRAIN_VIEWS_8[: args.views]
    data = synthetic.make_dataset(args.det, args.samples, dev, views=views)
    torch.manual_seed(1)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    nerfca_amd.set_precision(args.prec, s, t)
    cfg = TrainConfig(depth_samples_per_ray_coarse=args.samples, img_sample_size=args.rays * world)
    tr = CompositeTrainer(cfg, s, t, data, dev, rank=rank, world=world, seed=0, fused_loss=not args.torch_losses)
    tr.always_allreduce = use_pg

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
    if args.graph:       # events cannot be recorded inside a replayed graph: time the same kernels eagerly, outside dt
        timed_steps = min(args.steps, 4)
        _capi.timing_enable(True)
        for i in range(timed_steps):
            tr.step(base_iter + i)
        barrier()
    _capi.timing_enable(False)
    if use_pg:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        n_samp = args.rays * args.samples * timed_steps               # per GPU over the span the kernel timers covered
        kern = {}
        # bf16 with a forward store: the weight gradient of the last hidden layer of both nets (2 x 2 x 128 x 128 FLOP per sample)
        # is accumulated inside the dgrad kernel, not by the wgrad kernel
        moved = 2 * 2 * 128 * 128 if (args.prec == "bf16" and os.environ.get("NCA_ONCHIP", "1") != "0") else 0
        for name, flop in (("fwd", FLOP_FWD), ("bwd_dgrad", FLOP_DGRAD + moved), ("bwd_wgrad", FLOP_WGRAD - moved), ("bwd_reduce", 0), ("loss", 0), ("pack", 0)):
            ms, n = _capi.timing_read(name)
            kern[name] = {"ms_total": ms, "launches": n, "avg_ms": ms / n if n else None,
                          "tflops": (flop * n_samp / (ms * 1e-3) / 1e12) if ms > 0 and flop else None}
        dom = max(("fwd", "bwd_dgrad", "bwd_wgrad"), key=lambda k: kern[k]["ms_total"])
        peak = PEAK_TFLOPS[args.prec]
        roof = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                "frac": kern[dom]["tflops"] / peak if kern[dom]["tflops"] else None, "traffic": measured_traffic(args, dom),
                "avg_launch_ms": kern[dom]["avg_ms"], "launches": kern[dom]["launches"],
                # the staged (non-algorithmic) HBM traffic of that kernel per second of its run time: what it is bound by in practice
                "staging_TBps": (measured_traffic(args, dom) / (kern[dom]["avg_ms"] * 1e-3) / 1e12) if measured_traffic(args, dom) and kern[dom]["avg_ms"] else None,
                "simd_issue_share_pmc": measured_issue_share(args, dom),
                "kernel_time_share": kern[dom]["ms_total"] / (dt * 1e3), "all_kernels": kern}
        out = {"metric": f"training rays/sec ({args.det}^2 det, {args.samples} samples/ray)", "value": args.rays * world * args.steps / dt, "unit": "rays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
               "config": {"workload": f"run_composite XCAT {args.views}-view x 10 phases, {args.det}^2 detector x {args.samples} samples/ray, "
                                      f"{args.rays} rays/step/GPU (one full detector), F=128 x 4 hidden layers x 2 nets, L=12, fwd+losses+bwd+Adam",
                          "rays_per_step_per_gpu": args.rays, "samples_per_ray": args.samples, "parallelism": f"ray-sharded dp{world}", "hip_graph": bool(args.graph)},
               "roofline": roof, "final_loss": float(loss)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, data, None)
        if world == 1 and args.unfused_gpu_rays > 0:
            out["unfused_gpu_baseline"] = unfused_gpu_baseline(args, data)
        print(json.dumps(out))
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
