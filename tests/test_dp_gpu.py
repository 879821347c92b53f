"""Ray-sharded data parallelism with the REAL kernels: two ranks that share the one GPU of the test box
(process group over gloo -- RCCL refuses two ranks on one device; the collective is the same single all-reduce of the flat
gradient) against one rank: fused forward / loss / backward steps (`step_fused`), the autograd step, and the hierarchical
pass with the reference's through-depth gradient (cross-rank maximum and ray-0 backward).  tests/test_dp_gloo.py covers
the same logic on the CPU with the oracle injected as the renderer; bench.py's N>1 path is this trainer over RCCL.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _run(rank, world, mode, prec, device_index=0):
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    dev = torch.device("cuda", device_index)
    torch.cuda.set_device(dev)
    S, R = 48, 192
    data = synthetic.make_dataset(16, S, dev, views=synthetic.TRAIN_VIEWS[:2], n_phases=3, F=32)
    torch.manual_seed(3)
    sdef, tdef = synthetic.net_definitions(dev, F=64)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    kw = {}
    n_fine = 8 if mode in ("fine", "graphfine") else 0
    if n_fine:
        fs, ft = synthetic.net_definitions(dev, F=32)
        kw = dict(static_model_fine=CPPN(fs).to(dev), temp_model_fine=Temporal(ft).to(dev))
    nerfca_amd.set_precision(prec, s, t, *kw.values())
    cfg = TrainConfig(depth_samples_per_ray_coarse=S, depth_samples_per_ray_fine=n_fine, img_sample_size=R, favor_s_weight_delay_steps=0,
                      l1_weight_start=1e-3, l1_weight_end=1e-3, occl_weight_start=1e-2, dynamic_entro_weight_start=1e-3,
                      favor_s_weight_start=1e-3, entro_mask_thre=1e-6)
    tr = CompositeTrainer(cfg, s, t, data, dev, rank=rank, world=world, seed=11, fused_loss=(mode in ("fused", "graph", "graphfine")), **kw)
    grads = None
    for it in range(2):
        if mode == "graphfine":    # step_graph with a fine pass: one rank replays a graph, sharded ranks run the host-launched step
            tr.step_graph(2000 + it)
            if it == 0:
                grads = (tr._graph_out["flat"][:-13] if getattr(tr, "_graphs", None) else torch.cat([p.grad.flatten() for p in tr.params])).clone()
        elif mode == "graph":      # bench.py's default step: two captured graphs with the gradient all-reduce between them
            tr.step_graph(2000 + it)
            if it == 0:
                grads = tr._graph_out["flat"][:-13].clone()      # [dynamic | static] = the order of tr.params (behind them: the 13 loss terms as f32)
        else:
            tr.step(2000 + it)
            if it == 0:
                grads = torch.cat([p.grad.flatten() for p in tr.params]).clone()
    params = torch.cat([p.detach().flatten() for p in tr.params]).clone()
    torch.cuda.synchronize()
    return grads.cpu(), params.cpu()


def _worker(rank, world, port, outdir, mode, prec, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":          # RCCL: one device per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, p = _run(rank, world, mode, prec, device_index=rank if backend == "nccl" else 0)
        torch.save({"g": g, "p": p}, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,prec,gtol", [("fused", "f32", 1e-5), ("autograd", "f32", 1e-5), ("fused", "bf16", 1e-3), ("fine", "f32", 1e-3),
                                            ("graph", "f32", 1e-5), ("graph", "bf16", 1e-3), ("graphfine", "f32", 1e-3)])
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path, mode, prec, gtol):
    """Gradient of the first step: both ranks hold the same all-reduced buffer, equal to the single-process gradient up to
    f32 summation order (1e-5; bf16 rounds the per-rank partial sums differently, and the fine pass's through-depth term is
    ill-conditioned: 1e-3).  Parameters after two steps only guard against a wrong step (Adam divides by sqrt(v))."""
    g1, p1 = _run(0, 1, mode, prec)
    assert float(g1.abs().max()) > 0
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), mode, prec), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["p"], r1["p"])
    gerr = float((r0["g"] - g1).abs().max() / g1.abs().max())
    perr = float((r0["p"] - p1).abs().max() / p1.abs().max())
    assert gerr < gtol, gerr
    assert perr < 1e-2, perr


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,prec,gtol", [("fused", "f32", 1e-5), ("fused", "bf16", 1e-3), ("fine", "f32", 1e-3)])
def test_two_ranks_over_rccl_equal_one_rank(tmp_path, mode, prec, gtol):
    """The same comparison with backend `nccl` (= RCCL), one GPU per rank: runs where the box has two GPUs (the driver's
    multi-GPU node); a one-GPU box skips it -- RCCL refuses two ranks on one device."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL takes one device per rank)")
    g1, p1 = _run(0, 1, mode, prec)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), mode, prec, "nccl"), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["p"], r1["p"])
    assert float((r0["g"] - g1).abs().max() / g1.abs().max()) < gtol


def _graph_steps(n_steps, prec, allreduce):
    """`n_steps` graph-replayed steps (bench.py's default step) of one rank; with `allreduce` the gradient all-reduce runs between
    the two captured graphs as it does under ray sharding.  Returns the bits of every step's loss and the final parameters."""
    import nerfca_amd
    from nerfca_amd import synthetic
    from nerfca_amd.model.CPPN import CPPN
    from nerfca_amd.model.Temporal import Temporal
    from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    S, R = 64, 2048
    data = synthetic.make_dataset(32, S, dev, views=synthetic.TRAIN_VIEWS[:2], n_phases=3, F=32)
    torch.manual_seed(5)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    nerfca_amd.set_precision(prec, s, t)
    tr = CompositeTrainer(TrainConfig(depth_samples_per_ray_coarse=S, img_sample_size=R), s, t, data, dev, rank=0, world=1, seed=4)
    tr.always_allreduce = allreduce
    losses = []
    for it in range(n_steps):
        loss, _, _ = tr.step_graph(75000 + it)
        losses.append(loss.detach().clone().view(1).view(torch.int64 if loss.dtype == torch.float64 else torch.int32).cpu())
    torch.cuda.synchronize()
    return torch.cat(losses), torch.cat([p.detach().flatten() for p in tr.params]).cpu()


def _rccl_one_rank_worker(rank, port, outdir, n_steps, prec):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        l, p = _graph_steps(n_steps, prec, True)
        torch.save({"l": l, "p": p}, os.path.join(outdir, "rccl1.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_one_rank_rccl_graph_step_is_bit_identical_to_no_process_group(tmp_path, prec):
    """The graph-replayed step with an RCCL process group of ONE rank (what `NERFCA_FORCE_PG=1 bench.py` runs, and the N = 1 end of the
    driver's scaling run): the all-reduce between the two captured graphs must be a no-op on values -- ten steps, every loss and the
    final parameters bit-identical to the step without a process group."""
    l0, p0 = _graph_steps(10, prec, False)
    mp.spawn(_rccl_one_rank_worker, args=(_free_port(), str(tmp_path), 10, prec), nprocs=1, join=True)
    r = torch.load(tmp_path / "rccl1.pt")
    assert torch.equal(r["l"], l0), (r["l"], l0)
    assert torch.equal(r["p"].view(torch.int32), p0.view(torch.int32))


def _bench(args, env_extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_over_rccl_with_one_rank():
    """bench.py's N > 1 code path on the one-GPU box: RCCL process group of one rank (NERFCA_FORCE_PG=1), gradient all-reduce
    every step, barriers, max-over-ranks time -- in a child process, as the driver runs it."""
    port = str(_free_port())
    line = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--rays", "8192", "--no-cpu-baseline", "--no-extras"],
                  {"NERFCA_FORCE_PG": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1 and line["value"] > 0


@pytest.mark.timeout(1200)
def test_bench_self_launch_two_gpus():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks itself (before it touches the
    GPU), they train over RCCL, rank 0's line comes back.  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    line = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {})
    assert line["rccl_ranks"] == 2 and line["n_gpus"] == 2 and line["config"]["parallelism"] == "ray-sharded dp2"
