"""World-size-2 `gloo` test of the batch-sharded path + observation gather (CPU, emulated kernel)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import emu_backend

emu_backend.build()   # once, in the parent: the ranks below must not all start compiling the emulator library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from predpreygrass_amd import _abi
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env
    from predpreygrass_amd.distributed import ObservationGatherer, shard_range
    from tests.emu_backend import library

    total = 5
    lo, hi = shard_range(total, rank, world)
    env = BatchedPredPreyGrass(config_env, batch_size=hi - lo, _library=library(), seed=100 + lo)
    env.reset()
    g = ObservationGatherer(env)
    for _ in range(25):
        env.step(random_actions=True, auto_reset=True)
    res = g.gather()
    # every rank holds every shard; compare with what the owning rank sees locally
    local = g.pack_local()
    np.savez(os.path.join(tmpdir, f"local{rank}.npz"), **{k: v.numpy() for k, v in local.items()})
    dist.barrier()
    for r in range(world):
        want = np.load(os.path.join(tmpdir, f"local{r}.npz"))
        for k in want.files:
            assert np.array_equal(res[k][r].numpy(), want[k]), (rank, r, k)
    # shards are the same envs a single process would own: seeds 100 + global index
    if rank == 0:
        ref = BatchedPredPreyGrass(config_env, batch_size=total, _library=library(), seed=100)
        ref.reset()
        for _ in range(25):
            ref.step(random_actions=True, auto_reset=True)
        es = torch.cat([res["env_state"][r] for r in range(world)])
        assert torch.equal(es[:, :13], ref.env_state[:, :13])
        cat = torch.cat([res["obs_prey"][r] for r in range(world)])
        nQ = ref.env_state[:, _abi.ENV_N_PREY_ROWS]
        mask = torch.arange(ref.prey_capacity)[None, :] < nQ[:, None]
        assert torch.equal(cat, ref.obs_prey[mask])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_observation_gather(tmp_path):
    port = 29500 + os.getpid() % 500
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_shard_range_partitions_exactly():
    from predpreygrass_amd.distributed import shard_range
    for total in (1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _run_bench(args, nproc):
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT)
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200),
               os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_contract_single_process_dry_run():
    """bench.py's control flow and JSON contract on CPU (emulated kernel; numbers meaningless)."""
    d = _run_bench(["--dry-run-cpu", "--gpus", "1", "--steps", "6", "--warmup", "2", "--envs", "6", "--streams", "2"], 1)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "workload" in d["config"] and "model" not in d["config"]


def test_bench_two_ranks_dry_run_with_gather_leg():
    """`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as the driver launches it (gloo on CPU):
    barrier + max-over-ranks timing, rank-0 JSON, the obs_gather leg."""
    d = _run_bench(["--dry-run-cpu", "--gpus", "2", "--steps", "5", "--warmup", "2", "--envs", "5", "--streams", "2",
                    "--gather-steps", "3"], 2)
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 5
    assert "obs_gather" in d and "error" not in d["obs_gather"], d.get("obs_gather")
    assert d["obs_gather"]["steps"] == 3 and d["obs_gather"]["gathered_bytes_per_step_per_rank"] > 0
    assert "error" not in d["obs_gather_overlapped"], d["obs_gather_overlapped"]
    assert d["obs_gather_overlapped"]["gathered_bytes_per_step_per_rank"] > 0
