#!/usr/bin/env python3
"""Diagnostic: where does a wavefront spend its cycles?  Builds libppg_hip_prof.so
(-DPPG_PROFILE_PHASES: s_memtime stamps at phase boundaries, written to a buffer nothing else
reads) and prints per-phase cycle shares.  Never quote this build's run time; read the SHARES."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from predpreygrass_amd import _abi
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env

CSRC = os.path.join(ROOT, "predpreygrass_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "_build", "libppg_hip_prof.so")  # diagnostic build, git-ignored
os.makedirs(os.path.dirname(LIB), exist_ok=True)
import __graft_entry__ as graft
MOVE = "--move-detail" in sys.argv   # (non-cooperative plans only: slots 13-15 of the stamp buffer)
if MOVE:
    LIB = LIB.replace("_prof.so", "_profmove.so")
graft.build_hip(force=not os.path.exists(LIB), extra_flags=["-DPPG_PROFILE_PHASES"] + (["-DPPG_PROFILE_MOVE"] if MOVE else []), out=LIB)
lib = _abi.bind(ctypes.CDLL(LIB))
_abi._lib = lib  # this process only
lib.ppg_debug_set_profile_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
RQ = "--rq" in sys.argv   # second-generation env on its reference config ("engage_pred" = type-1 classes, "engage_prey" = type-2)
WALLS = "--walls" in sys.argv   # walls variant, zigzag layout with every line-of-sight option (bench.py --workload walls)
DRIVE = "--drive" in sys.argv   # drive-conditioned variant of the default config
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if len(args) > 0 else 4096
warm = int(args[1]) if len(args) > 1 else 400
if WALLS:
    from predpreygrass_amd.red_queen import BatchedRedQueen
    from predpreygrass_amd.walls_occlusion import config_env_zigzag_walls
    env = BatchedRedQueen(config_env_zigzag_walls, batch_size=B, device="cuda:0", walls=True, obs_dtype=torch.float32)
    env.set_walls(config_env_zigzag_walls["manual_wall_positions"])
elif RQ:
    from predpreygrass_amd.red_queen import BatchedRedQueen, config_env_base
    env = BatchedRedQueen(config_env_base, batch_size=B, device="cuda:0", obs_dtype=torch.float32)
elif "--c4" in sys.argv:   # BASELINE.json configs[3]: 64x64 grid, 16 predators / 32 prey, 7x7 windows
    env = BatchedPredPreyGrass({**config_env, "grid_size": 64, "n_initial_active_predator": 16, "n_initial_active_prey": 32,
                                "predator_obs_range": 7, "prey_obs_range": 7}, batch_size=B, device="cuda:0")
elif DRIVE:
    env = BatchedPredPreyGrass({**config_env, "enable_drive_channels": True}, batch_size=B, device="cuda:0")
else:
    env = BatchedPredPreyGrass(config_env, batch_size=B, device="cuda:0")
PLAN = [a for a in sys.argv if a.startswith("--plan=")]
if PLAN:   # e.g. --plan=4,0,2: force a wave plan (ppg_set_wave_plan)
    env.set_wave_plan(*[int(v) for v in PLAN[0][7:].split(",")])
print("kernel:", env.step_kernel_name(), env.wave_plan())
env.reset()
prof = torch.zeros((B, 16), dtype=torch.int64, device="cuda:0")
for _ in range(warm):
    env.step(random_actions=True, auto_reset=True)
lib.ppg_debug_set_profile_buffer(env._handle, ctypes.c_void_p(prof.data_ptr()))
names = ["load_env+rows", "actions", "decay", "grass", "move", "sort", "build_maps", "engage_pred", "engage_prey",
         "reproduce", "obs", "store"]
COOP = env.wave_plan()[2] > 0   # cooperative kernels: "obs" = publishing the row lists; then the workgroup barrier and the shared write phase
if COOP:
    names += ["wait_barrier", "coop_write"]
NS = len(names)
acc = np.zeros((NS,)); accmax = np.zeros((NS,)); tot = []; rows = []
N = 20
for it in range(N):
    prof.zero_()
    env.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    es = env.env_state.cpu().numpy()
    ok = (p[:, 12] > 0) & (p[:, 1] > 0)   # envs that took the normal path (not reset / truncation)
    d = np.diff(p[ok][:, :NS + 1], axis=1)
    acc += d.mean(axis=0)
    whole = p[ok][:, NS] - p[ok][:, 0]
    slow = np.argsort(whole)[-max(1, len(whole) // 100):]
    accmax += d[slow].mean(axis=0)
    tot.append((whole.mean(), whole.max(), np.percentile(whole, 99)))
    n = (es[ok][:, 0] + es[ok][:, 1])
    rows.append((n.mean(), n.max(), np.corrcoef(n, whole)[0, 1]))
    span = p[ok][:, NS].max() - p[ok][:, 0].min()
    if MOVE and not COOP:
        mv = p[ok][:, 13:16]
        n_ord = np.bitwise_and(mv[:, 2].astype(np.int64), 0xFFFF); n_all = mv[:, 2].astype(np.int64) >> 16
        movedetail = locals().get("movedetail", [])
        movedetail.append((mv[:, 0].mean(), mv[:, 1].mean(), n_ord.mean(), n_all.mean(), (mv[:, 1] / np.maximum(n_ord, 1)).mean(),
                           mv[slow, 0].mean(), mv[slow, 1].mean(), n_ord[slow].mean()))
print("phase               mean cyc   share | slowest-1%% cyc  share")
for k, n in enumerate(names):
    print(f"{n:18s} {acc[k]/N:9.0f}  {acc[k]/acc.sum():6.1%} | {accmax[k]/N:9.0f}  {accmax[k]/accmax.sum():6.1%}")
t = np.array(tot); r = np.array(rows)
print(f"wave cycles: mean {t[:,0].mean():.0f}  p99 {t[:,2].mean():.0f}  max {t[:,1].mean():.0f};  first-start..last-end span {span:.0f}")
print(f"rows/env: mean {r[:,0].mean():.1f} max {r[:,1].mean():.0f}; corr(rows, cycles) {r[:,2].mean():.3f}")
if MOVE and not COOP:
    m = np.array(movedetail).mean(axis=0)
    print(f"move detail: lane-parallel part {m[0]:.0f} cyc, ordered loop {m[1]:.0f} cyc for {m[2]:.2f} of {m[3]:.1f} acting agents ({m[4]:.0f} cyc per ordered agent);"
          f" slowest 1 %: parallel {m[5]:.0f}, ordered {m[6]:.0f} for {m[7]:.1f} agents")
