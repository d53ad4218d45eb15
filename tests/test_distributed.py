"""World-size-2 / -4 `gloo` tests of the batch-sharded path + the single-collective observation gather (CPU, emulated kernel)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from predpreygrass_amd import _abi as _ABI
from tests import emu_backend

emu_backend.build()   # once, in the parent: the ranks below must not all start compiling the emulator library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def local_rows_reference(env):
    """What `ppg_pack` must produce for one env object, computed with plain torch indexing (tests)."""
    es = env.env_state
    cp = env.pred_capacity
    mp = torch.arange(env.pred_capacity, device=env.device)[None, :] < es[:, _ABI.ENV_N_PRED_ROWS:_ABI.ENV_N_PRED_ROWS + 1]
    mq = torch.arange(env.prey_capacity, device=env.device)[None, :] < es[:, _ABI.ENV_N_PREY_ROWS:_ABI.ENV_N_PREY_ROWS + 1]
    return {
        "env_state": es.clone(),
        "id_pred": env.row_id[:, :cp][mp], "id_prey": env.row_id[:, cp:][mq],
        "reward_pred": env.row_reward[:, :cp][mp], "reward_prey": env.row_reward[:, cp:][mq],
        "flags_pred": env.row_flags[:, :cp][mp], "flags_prey": env.row_flags[:, cp:][mq],
        "obs_pred": env.obs_pred[mp].reshape(int(mp.sum()), -1), "obs_prey": env.obs_prey[mq].reshape(int(mq.sum()), -1),
    }


def _worker(rank, world, port, tmpdir, n_sub, wire_f32):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from predpreygrass_amd import _abi
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env
    from predpreygrass_amd.distributed import ObservationGatherer, shard_range
    from tests.emu_backend import library

    total = 2 * world + 1
    lo, hi = shard_range(total, rank, world)
    # the shard as n_sub sub-batches (handles) -- ppg_pack concatenates them in handle order
    spans = [shard_range(hi - lo, k, n_sub) for k in range(n_sub)]
    envs = [BatchedPredPreyGrass(config_env, batch_size=b - a, _library=library(), seed=100 + lo + a) for a, b in spans if b > a]
    for e in envs:
        e.reset()
    # deliberately tiny image: the first gather overflows on every rank and grow() has to enlarge it everywhere
    g = ObservationGatherer(envs, wire_dtype=torch.float32 if wire_f32 else None, rows_per_env=(1, 1))
    for _ in range(25):
        for e in envs:
            e.step(random_actions=True, auto_reset=True)
    slot = g.gather()                      # ONE collective
    assert all(h.overflow == 1 for h in g.headers(slot))
    assert g.grow(slot)
    res = g.gather_dict()
    want = [local_rows_reference(e) for e in envs]
    mine = {k: torch.cat([w[k] for w in want]) for k in want[0]}
    if wire_f32:
        mine["obs_pred"], mine["obs_prey"] = mine["obs_pred"].float(), mine["obs_prey"].float()
    np.savez(os.path.join(tmpdir, f"local{rank}.npz"), **{k: v.numpy() for k, v in mine.items()})
    dist.barrier()
    for r in range(world):
        ref = np.load(os.path.join(tmpdir, f"local{r}.npz"))
        for k in ref.files:
            assert np.array_equal(res[k][r].numpy(), ref[k]), (rank, r, k)
        # row_off = exclusive prefix sums of the row counts
        es = res["env_state"][r]
        for col, w in ((0, _abi.ENV_N_PRED_ROWS), (1, _abi.ENV_N_PREY_ROWS)):
            c = es[:, w].to(torch.int64)
            assert torch.equal(res["row_off"][r][:, col].to(torch.int64), torch.cumsum(c, 0) - c)
    # shards are the same envs a single process would own: seeds 100 + global index
    if rank == 0:
        ref = BatchedPredPreyGrass(config_env, batch_size=total, _library=library(), seed=100)
        ref.reset()
        for _ in range(25):
            ref.step(random_actions=True, auto_reset=True)
        es = torch.cat([res["env_state"][r] for r in range(world)])
        assert torch.equal(es[:, :13], ref.env_state[:, :13])
        cat = torch.cat([res["obs_prey"][r] for r in range(world)])
        want_all = local_rows_reference(ref)["obs_prey"]
        assert torch.equal(cat, want_all.float() if wire_f32 else want_all)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_sub,wire_f32", [(2, 1, False), (2, 3, True), (4, 2, False)])
def test_sharding_and_single_collective_observation_gather(tmp_path, world, n_sub, wire_f32):
    port = 29500 + (os.getpid() * 7 + world * 3 + n_sub) % 500
    mp.spawn(_worker, args=(world, port, str(tmp_path), n_sub, wire_f32), nprocs=world, join=True)


def _worker_modes(rank, world, port, tmpdir):
    """The image without observations (PPG_PACK_NO_OBS), gather-to-root and the all-pairs spelling of the all-gather, `fit()`
    sizing from bytes_used, shards of very different sizes (rank 0 holds 5 envs, the others 1)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from predpreygrass_amd import _abi
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env
    from predpreygrass_amd.distributed import ObservationGatherer
    from tests.emu_backend import library

    env = BatchedPredPreyGrass(config_env, batch_size=5 if rank == 0 else 1, _library=library(), seed=40 + 10 * rank)
    env.reset()
    for _ in range(20):
        env.step(random_actions=True, auto_reset=True)
    want = local_rows_reference(env)
    np.savez(os.path.join(tmpdir, f"m{rank}.npz"), **{k: v.numpy() for k, v in want.items()})
    dist.barrier()
    refs = [np.load(os.path.join(tmpdir, f"m{r}.npz")) for r in range(world)]

    # 1. no observations on the wire: sized for the LARGEST shard (5 envs) on every rank, a few hundred bytes per env
    g = ObservationGatherer(env, include_obs=False)
    res = g.gather_dict()
    full = ObservationGatherer(env)
    assert g.capacity * 20 < full.capacity
    for r in range(world):
        for k in ("env_state", "id_pred", "id_prey", "reward_pred", "reward_prey", "flags_pred", "flags_prey"):
            assert np.array_equal(res[k][r].numpy(), refs[r][k]), (rank, r, k)
        assert res["obs_prey"][r].shape[1] == 0 and res["obs_pred"][r].shape[1] == 0
        assert g.headers()[r].blk_pred == 0

    # 2. gather-to-root: rank 1 receives everything, the others nothing
    g = ObservationGatherer(env, mode="gather", dst=1)
    slot = g.gather()
    assert g.holds_all == (rank == 1)
    assert [int(h.n_envs) for h in g.headers(slot)] == [5] + [1] * (world - 1)     # (headers are exchanged on every rank)
    if rank == 1:
        for r in range(world):
            v = g.views(r, slot)
            assert np.array_equal(v["obs_prey"].numpy(), refs[r]["obs_prey"]) and np.array_equal(v["id_pred"].numpy(), refs[r]["id_pred"])
    else:
        assert np.array_equal(g.views(rank, slot)["obs_prey"].numpy(), refs[rank]["obs_prey"])     # its own image only
        with pytest.raises(RuntimeError):
            g.image((rank + 1) % world, slot)

    # 3. the all-gather as grouped point-to-point copies; fit(): grows after an overflow, shrinks to what is used
    g = ObservationGatherer(env, mode="all_pairs", rows_per_env=(1, 1))
    slot = g.gather()
    assert g.fit(slot)                                  # overflowed -> grown, gather again
    slot = g.gather()
    for r in range(world):
        assert np.array_equal(g.views(r, slot)["obs_pred"].numpy(), refs[r]["obs_pred"])
    big = ObservationGatherer(env, mode="all_pairs", rows_per_env=(60, 120))
    slot = big.gather()
    cap0 = big.capacity
    used = max(int(h.bytes_used) for h in big.headers(slot))
    assert not big.fit(slot) and big.capacity == cap0   # nothing to redo, and nothing reallocated yet:
    for r in range(world):                              # the images of that gather are still there to be consumed
        assert np.array_equal(big.views(r, slot)["obs_prey"].numpy(), refs[r]["obs_prey"])
    slot = big.gather()                                 # the NEXT pack shrinks the buffers
    assert used <= big.capacity <= int(used * 1.16) + 256 and big.capacity < cap0
    for r in range(world):
        assert np.array_equal(big.views(r, slot)["obs_prey"].numpy(), refs[r]["obs_prey"])
    dist.barrier()
    dist.destroy_process_group()


def _worker_subgroup(rank, world, port, tmpdir):
    """A gatherer on a SUB-GROUP of the job (ranks 1..world-1): its ranks, its root and its point-to-point peers are group-relative
    and have to be translated to global ranks for torch.distributed (a hang or a wrong destination otherwise)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    from predpreygrass_amd.config import config_env
    from predpreygrass_amd.distributed import ObservationGatherer
    from tests.emu_backend import library
    members = list(range(1, world))
    sub = dist.new_group(members)          # (every rank of the job takes part in creating it)
    if rank in members:
        env = BatchedPredPreyGrass(config_env, batch_size=2, _library=library(), seed=70 + 10 * rank)
        env.reset()
        for _ in range(12):
            env.step(random_actions=True, auto_reset=True)
        want = local_rows_reference(env)
        np.savez(os.path.join(tmpdir, f"s{rank}.npz"), **{k: v.numpy() for k, v in want.items()})
        dist.barrier(group=sub)
        refs = [np.load(os.path.join(tmpdir, f"s{r}.npz")) for r in members]
        for mode in ("all_gather", "all_pairs", "gather"):
            g = ObservationGatherer(env, group=sub, mode=mode, dst=len(members) - 1)    # root = the group's LAST rank
            assert g.world == len(members) and g.rank == rank - 1
            slot = g.gather()
            if g.holds_all:
                for r in range(g.world):
                    assert np.array_equal(g.views(r, slot)["obs_prey"].numpy(), refs[r]["obs_prey"]), (mode, rank, r)
            else:
                assert np.array_equal(g.views(g.rank, slot)["id_prey"].numpy(), refs[g.rank]["id_prey"])
        dist.barrier(group=sub)
    dist.barrier()
    dist.destroy_process_group()


def test_gatherer_on_a_sub_group_translates_ranks(tmp_path):
    world = 3
    port = 30600 + (os.getpid() * 3) % 300
    mp.spawn(_worker_subgroup, args=(world, port, str(tmp_path)), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_gather_modes_no_obs_image_and_fit(tmp_path, world):
    port = 30100 + (os.getpid() * 5 + world) % 400
    mp.spawn(_worker_modes, args=(world, port, str(tmp_path)), nprocs=world, join=True)


def test_shard_range_partitions_exactly():
    from predpreygrass_amd.distributed import shard_range
    for total in (1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _run_bench(args, nproc, launcher=True):
    """tests/bench_dry.py = bench.main() on the CPU stand-in (emulated kernel, gloo); same flags as bench.py."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    script = os.path.join(ROOT, "tests", "bench_dry.py")
    if nproc == 1 or not launcher:
        cmd = [sys.executable, script] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200), script] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_contract_single_process_dry_run():
    """bench.py's control flow and JSON contract on CPU (emulated kernel; numbers meaningless)."""
    d = _run_bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--envs", "6", "--streams", "2",
                    "--preroll-min", "64", "--preroll-max", "128"], 1)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_survey_formula"}
    assert "workload" in d["config"] and "model" not in d["config"]
    assert 64 <= d["config"]["preroll_steps"] <= 128 and d["config"]["mean_agents_per_env"] > 0
    d = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--envs", "4", "--streams", "1", "--workload", "red_queen",
                    "--preroll-max", "0"], 1)
    assert d["dtype"] == "f32" and d["config"]["preroll_steps"] == 0


@pytest.mark.parametrize("launcher", [True, False])
def test_bench_two_ranks_dry_run_with_gather_leg(launcher):
    """`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as the driver launches it, and plain
    `bench.py --gpus 2` (which starts its two ranks itself), gloo on CPU: barrier + max-over-ranks timing, rank-0 JSON,
    the single-collective obs_gather legs."""
    d = _run_bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--envs", "5", "--streams", "2",
                    "--gather-steps", "3", "--preroll-min", "64", "--preroll-max", "64"], 2, launcher=launcher)
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 5
    for leg in ("obs_gather", "obs_gather_overlapped", "obs_gather_to_root", "obs_all_pairs", "ids_rewards_gather"):
        assert leg in d and "error" not in d[leg], d.get(leg)
        assert d[leg]["steps"] == 3 and d[leg]["collectives_per_step"] == 1 and d[leg]["image_overflows"] == 0
        assert d[leg]["wire_bytes_per_step_per_rank"] > 0 and len(d[leg]["image_bytes_used_last_step"]) == 2
    # the image without observations is two orders of magnitude smaller; images are sized from what a step really used
    assert d["ids_rewards_gather"]["wire_bytes_per_step_per_rank"] * 20 < d["obs_gather"]["wire_bytes_per_step_per_rank"]
    assert d["obs_gather"]["wire_bytes_per_step_per_rank"] < 1.25 * max(d["obs_gather"]["image_bytes_used_last_step"]) + 512
    # the N-GPU line explains itself: every rank's own pace and GPU next to the max-over-ranks `value`
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    for r in d["per_rank"]:
        assert set(r) >= {"ms_per_step", "kernel_ms", "placement_probe_us_min", "gpu_uuid"} and r["ms_per_step"] > 0
    assert d["ms_per_step"] >= max(r["ms_per_step"] for r in d["per_rank"]) * 0.999   # (`value` is the slowest rank's)
    assert d["value_if_every_rank_were_median"] >= d["value"] * 0.999 and d["slowest_over_median_rank"] >= 1.0
    for leg in ("obs_gather", "obs_all_pairs", "ids_rewards_gather"):
        assert len(d[leg]["per_rank_ms_per_step"]) == 2


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry.py"), "--gpus", "4", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_device_state_is_parsed_from_the_side_samplers_file(tmp_path):
    """bench_support.sampled_state: the last rocm-smi block taken entirely inside the window, parsed into clocks / power / temperatures /
    listed KFD processes (the sampler itself needs a GPU box; its file format does not)."""
    import bench_support
    block = """============================ ROCm System Management Interface ============================
GPU[0]		: Temperature (Sensor junction) (C): 51.0
GPU[0]		: Temperature (Sensor memory) (C): 68.0
GPU[0]		: fclk clock level: 0: (1250Mhz)
GPU[0]		: mclk clock level: 0: (2000Mhz)
GPU[0]		: sclk clock level: 1: (2393Mhz)
GPU[0]		: Current Socket Graphics Package Power (W): 1081.0
KFD process information:
PID   	PROCESS NAME	GPU(s)	VRAM USED	SDMA USED    	CU OCCUPANCY	
523673	UNKNOWN     	0     	0        	0            	UNKNOWN     	
517951	python3     	1     	188780544	1313156012359	0           	
"""
    p = tmp_path / "state.txt"
    p.write_text("@@ 100.000 100.100\n" + block.replace("2393", "500") + "\n@@ 105.000 105.100\n" + block + "\n@@ 109.950 110.200\n" + block.replace("2393", "777") + "\n")
    assert bench_support.sampled_state(str(p), 200.0, 300.0) is None              # nothing inside the window
    s = bench_support.sampled_state(str(p), 104.0, 110.0)                           # the third block ends after the window
    assert s["sclk_MHz"] == 2393 and s["mclk_MHz"] == 2000 and s["fclk_MHz"] == 1250 and s["power_W"] == "1081.0"
    assert s["temp_junction_C"] == "51.0" and s["temp_memory_C"] == "68.0" and s["kfd_processes_listed"] == 2
    assert bench_support.sampled_state(str(tmp_path / "missing.txt"), 0.0, 1e12) is None
