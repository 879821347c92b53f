"""BASELINE.json's configurations beyond the single-GPU bench line: configs[2] (ray-sharded over 8 GPUs with an RCCL gradient
all-reduce) as far as it can go without the node -- the rendezvous bookkeeping of bench.py's own launcher, 8 ranks, stopped
before anything touches a GPU -- and configs[4] (the hyper-parameter sweep of train/sweep-composite.yaml as independent
single-GPU jobs) run for real on the one GPU of the test box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(text):
    """Every JSON object of the output, also when two ranks' records share a line (the ranks write to one pipe)."""
    out, dec = [], json.JSONDecoder()
    for l in text.splitlines():
        i = l.find("{")
        while i >= 0:
            try:
                obj, end = dec.raw_decode(l, i)
            except json.JSONDecodeError:
                break
            out.append(obj)
            i = l.find("{", end)
    return out


@pytest.mark.timeout(300)
def test_bench_self_launch_eight_ranks_dry():
    """`python bench.py --gpus 8 --dry-run`: the launcher starts 8 ranks through torch.distributed.run on 127.0.0.1; every rank
    sees world 8, its own rank / local rank / device index and the common rendezvous, and leaves before the first HIP call."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-800:]
    recs = _json_lines(r.stdout)
    launch = [x for x in recs if "launch" in x]
    ranks = [x for x in recs if "rank" in x]
    assert len(launch) == 1 and launch[0]["ranks"] == 8 and launch[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    cmd = launch[0]["launch"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1"][-6:]
    assert sorted(x["rank"] for x in ranks) == list(range(8))
    assert sorted(x["local_rank"] for x in ranks) == list(range(8))
    assert all(x["world"] == 8 and x["backend"] == "nccl" and x["device"] == f"cuda:{x['local_rank']}" for x in ranks)
    assert len({x["master"] for x in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")
    # the default is STRONG scaling (BASELINE configs[2]: the same composite config, ray-sharded across 8 GPUs): one global batch of 65 536 rays
    assert all(x["global_rays_per_step"] == 65536 and x["rays_per_rank"] == 8192 and x["scaling"] == "strong" for x in ranks)


@pytest.mark.timeout(300)
def test_bench_weak_scaling_eight_ranks_dry():
    """`--scaling weak`: every rank renders one full detector (65 536 rays) per step, the global batch is 8 x that and the line
    would say "weak"; under the default (strong) a ray count that does not divide by the ranks is refused before anything starts."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--scaling", "weak"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-800:]
    ranks = [x for x in _json_lines(r.stdout) if "rank" in x]
    assert sorted(x["rank"] for x in ranks) == list(range(8))
    assert all(x["scaling"] == "weak" and x["rays_per_rank"] == 65536 and x["global_rays_per_step"] == 8 * 65536 for x in ranks)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--rays", "1001"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "does not divide" in r.stderr


def test_bench_refuses_a_world_that_does_not_match():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 8" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sweep_launcher_runs_two_grid_points_on_one_gpu():
    """configs[4]: tools/sweep_launch.py starts one independent bench.py process per grid point, each pinned to a GPU before it
    touches it (HIP_VISIBLE_DEVICES), no communication; with one GPU the points run one after the other."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_launch.py"), "--gpus", "1", "--grid", "rays=8192,16384", "--",
                        "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, (r.stdout[-600:], r.stderr[-600:])
    recs = _json_lines(r.stdout)
    assert [x["point"] for x in recs] == [{"rays": "8192"}, {"rays": "16384"}]
    for x in recs:
        assert x["gpu"] == 0 and x["unit"] == "rays/s" and x["value"] > 1e5 and x["ms_per_step"] > 0 and x["final_loss"] == x["final_loss"]
    # larger batches amortise the fixed costs: more rays per second
    assert recs[1]["value"] > recs[0]["value"]
