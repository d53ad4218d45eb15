"""BatchedPredPreyGrass: B independent PredPreyGrass grids stepped in lockstep on one MI355X.

Host code is plumbing only: it owns the PyTorch-ROCm tensors (state, actions, observations),
hands their raw pointers to libppg_hip.so through the C ABI of include/ppg.h and launches one
HIP kernel per call on torch's current stream.  All environment logic
(predpreygrass_rllib_env.py:219-473 of the reference) runs in predpreygrass_amd/csrc/ppg_kernel.h.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import torch

from . import _abi
from .config import resolve_config

PREDATOR, PREY = 0, 1
PRED_CAPACITY = 64


def lexkey(ids) -> np.ndarray:
    """Sort key of the decimal id strings: same order as list.sort() on "prey_<id>"
    (predpreygrass_rllib_env.py:468); mirrors ppg_lexkey() in include/ppg.h."""
    ids = np.asarray(ids, dtype=np.int64)
    out = np.zeros(ids.shape, dtype=np.int64)
    flat, o = ids.reshape(-1), out.reshape(-1)
    for n, v in enumerate(flat):
        s = str(int(v))
        o[n] = sum((ord(ch) - 48 + 1) * 11 ** (5 - i) for i, ch in enumerate(s))
    return out.astype(np.int32)


def agent_name(type_: int, id_: int) -> str:
    return ("predator_%d" if type_ == PREDATOR else "prey_%d") % int(id_)


class BatchedPredPreyGrass:
    """Tensor API.

    Rows: ``[0, 64)`` predators, ``[64, 64 + prey_capacity)`` prey.  Within a type the rows of
    the last call are ordered like the dict the reference's ``step()`` returns:
    survivors (incl. agents that died in that call) followed by newborns.  ``actions[b, row]``
    answers the observation in the same row of the previous call; rows that are dead, unused or
    deliberately left out of the action dict hold ``ACTION_NONE`` (-1).
    """

    def __init__(self, config=None, batch_size=1, device=None, obs_dtype=torch.float64,
                 prey_capacity=128, seed=0, _library=None, obs_spread=0):
        """obs_spread = N > 1: the two observation tensors live in memory from `ppg_alloc_spread` (include/ppg.h) -- physical pages
        picked at random from a stretch of device memory N times their size, which is what HBM wants for the step's scattered writes
        (profiles/EXPERIMENTS.md, round 3; N = 32 costs a few seconds and N x the tensors' size of transient device memory).  They stay valid
        until close()."""
        cfg = resolve_config(config)
        self.config = cfg
        self._obs_spread = int(obs_spread)
        self.batch_size = int(batch_size)
        self._init_device(device, obs_dtype, _library)
        self.grid_size = int(cfg["grid_size"])
        self.Rp, self.Rq = int(cfg["predator_obs_range"]), int(cfg["prey_obs_range"])
        self.P0, self.Q0 = int(cfg["n_initial_active_predator"]), int(cfg["n_initial_active_prey"])
        self.n_grass = int(cfg["initial_num_grass"])
        # drive-conditioned variant (drive_conditioned_environment/predpreygrass_rllib_env.py:54-89): extra observation
        # channels filled with per-agent scalars; absent / False = the base environment
        drive = bool(cfg.get("enable_drive_channels", False))
        drive_lists = (list(cfg.get("predator_drive_channels", _abi.DEFAULT_PREDATOR_DRIVES)) if drive else [],
                       list(cfg.get("prey_drive_channels", _abi.DEFAULT_PREY_DRIVES)) if drive else [])
        for lst in drive_lists:
            for name in lst:
                if name not in _abi.DRIVE_KINDS:
                    raise ValueError(f"Unknown drive feature: {name!r}")   # its :610
            if len(lst) > 4:
                raise ValueError("at most 4 drive channels per species")
        self.obs_channels_pred, self.obs_channels_prey = 4 + len(drive_lists[0]), 4 + len(drive_lists[1])
        self._alloc_buffers(prey_capacity)
        NG = self.grass_capacity

        c = _abi.PpgConfig()
        for t in range(2):
            c.n_drive[t] = len(drive_lists[t])
            for k, name in enumerate(drive_lists[t]):
                c.drive_kind[t][k] = _abi.DRIVE_KINDS[name]
        c.hunger_safe_energy[0] = float(cfg.get("predator_hunger_safe_energy", cfg["initial_energy_predator"]))
        c.hunger_safe_energy[1] = float(cfg.get("prey_hunger_safe_energy", cfg["initial_energy_prey"]))
        c.prey_opportunity_normalizer = float(cfg.get("prey_opportunity_normalizer", cfg["initial_energy_prey"] * 3))
        c.predator_danger_normalizer = float(cfg.get("predator_danger_normalizer", cfg["initial_energy_predator"] * 2))
        c.grass_opportunity_normalizer = float(cfg.get("grass_opportunity_normalizer", cfg["initial_energy_grass"] * 5))
        c.abi_version = _abi.ABI_VERSION
        c.grid_size, c.predator_obs_range, c.prey_obs_range = self.grid_size, self.Rp, self.Rq
        c.max_steps = int(cfg["max_steps"])
        c.n_possible_predators, c.n_possible_prey = int(cfg["n_possible_predators"]), int(cfg["n_possible_prey"])
        c.n_initial_predators, c.n_initial_prey, c.n_grass = self.P0, self.Q0, self.n_grass
        c.pred_capacity, c.prey_capacity, c.grass_capacity = self.pred_capacity, self.prey_capacity, NG
        c.obs_dtype = self._abi_obs_dtype()
        for k_cfg, k_abi in [
            ("reward_predator_catch_prey",) * 2, ("reward_prey_eat_grass",) * 2, ("reward_predator_step",) * 2,
            ("reward_prey_step",) * 2, ("penalty_prey_caught",) * 2, ("reproduction_reward_predator",) * 2,
            ("reproduction_reward_prey",) * 2, ("energy_loss_per_step_predator",) * 2,
            ("energy_loss_per_step_prey",) * 2, ("predator_creation_energy_threshold",) * 2,
            ("prey_creation_energy_threshold",) * 2, ("initial_energy_predator",) * 2,
            ("initial_energy_prey",) * 2, ("initial_energy_grass",) * 2, ("energy_gain_per_step_grass",) * 2,
        ]:
            setattr(c, k_abi, float(cfg[k_cfg]))
        # seasonal variant keys (base_environment_seasonal/config_env.py:35-40); absent -> base environment
        c.season_length_steps = int(cfg.get("season_length_steps", 0) or 0)
        c.season_high_multiplier = float(cfg.get("season_high_multiplier", 1.0))
        c.season_low_multiplier = float(cfg.get("season_low_multiplier", 1.0))
        modes = {"sparse": 0, "dense_energy_delta": 1, "dense_energy_delta_plus_reproduction": 2}
        if cfg.get("reward_mode", "sparse") not in modes:
            raise ValueError(f"reward_mode must be one of {sorted(modes)}")
        c.reward_mode = modes[cfg.get("reward_mode", "sparse")]
        # kickback variant keys (project_reward_shaping/base_environment_sparse_rewards_plus_kickback/config_env.py:24-25)
        c.kickback = int("kickback_reward_predator" in cfg or "kickback_reward_prey" in cfg)
        c.kickback_reward_predator = float(cfg.get("kickback_reward_predator", 10.0))
        c.kickback_reward_prey = float(cfg.get("kickback_reward_prey", 10.0))
        if c.kickback and c.reward_mode != 0:
            raise ValueError("the kickback rewards are defined on top of the sparse reward mode only")
        self._create_handle(c, self._lib.ppg_create)
        self.set_seeds(seed)

    # ------------------------------------------------------------------
    def _init_device(self, device, obs_dtype, _library):
        self._emulated = _library is not None
        if _library is None:
            # the product path: HIP on a real GPU, or an exception
            self._lib = _abi.load_hip_library()
            if not torch.cuda.is_available():
                raise RuntimeError("predpreygrass_amd needs a ROCm GPU (gfx950); torch.cuda.is_available() is False")
            self.device = torch.device(device if device is not None else "cuda:0")
            if self.device.type != "cuda":
                raise RuntimeError(f"device must be a cuda (ROCm) device, got {self.device}")
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
        else:
            # test hook (tests/wave_emu): the same kernel source compiled for the CPU wave emulator
            self._lib = _library
            self.device = torch.device("cpu")
        # float64 = the reference's observations bit for bit; float32; bfloat16 = compact rows for a policy that runs next to the env
        # (FusedPolicy stages them without conversion: its logits are bit-identical to those from the float64 rows)
        if obs_dtype not in (torch.float64, torch.float32, torch.bfloat16):
            raise ValueError("obs_dtype must be torch.float64, torch.float32 or torch.bfloat16")
        self.obs_dtype = obs_dtype

    def _abi_obs_dtype(self):
        return {torch.float64: 0, torch.float32: 1, torch.bfloat16: 2}[self.obs_dtype]

    def _alloc_buffers(self, prey_capacity):
        self.pred_capacity = PRED_CAPACITY
        self.prey_capacity = int(prey_capacity)
        if self.n_grass > 255 and self.prey_capacity < 256:
            self.prey_capacity = 256   # up to 128 prey rows the kernels use 8-bit cell maps: at most 255 grass patches
        self.S = self.pred_capacity + self.prey_capacity
        self.grass_capacity = max(64, (self.n_grass + 63) // 64 * 64)
        B, S, NG, dev = self.batch_size, self.S, self.grass_capacity, self.device

        def z(shape, dtype):
            return torch.zeros(shape, dtype=dtype, device=dev)

        self.row_xy = z((B, S), torch.int16)
        self.row_energy = z((B, S), torch.float64)
        self.row_id = z((B, S), torch.int32)
        self.row_key = z((B, S), torch.int32)
        self.row_cumrew = z((B, S), torch.float64)
        self.row_flags = z((B, S), torch.uint8)
        self.row_reward = z((B, S), torch.float64)
        self.row_parent = torch.full((B, S), -1, dtype=torch.int32, device=dev)
        self.row_lastrep = z((B, S), torch.int32)  # second generation only (agent_last_reproduction)
        self.wall_bits = z((B, (self.grid_size * self.grid_size + 31) // 32), torch.int32)  # walls variant only
        self.row_info = z((B, S), torch.uint8)                                               # walls variant only
        self.env_state = z((B, _abi.ENV_WORDS), torch.int32)
        self.env_seed = z((B,), torch.int64)
        self.grass_xy = z((B, NG), torch.int16)
        self.grass_energy = z((B, NG), torch.float64)
        nc = getattr(self, "obs_channels", 4)
        ncp, ncq = getattr(self, "obs_channels_pred", nc), getattr(self, "obs_channels_prey", nc)
        self.obs_pred = self._obs_tensor((B, self.pred_capacity, ncp, self.Rp, self.Rp), seed_salt=1)
        self.obs_prey = self._obs_tensor((B, self.prey_capacity, ncq, self.Rq, self.Rq), seed_salt=2)
        self.actions = torch.full((B, S), _abi.ACTION_NONE, dtype=torch.int8, device=dev)

    def _obs_tensor(self, shape, seed_salt=0):
        """An observation tensor: from torch's allocator, or (obs_spread > 1, HIP library) on spread physical pages."""
        spread = getattr(self, "_obs_spread", 0)
        if spread <= 1 or self.device.type != "cuda" or not hasattr(self._lib, "ppg_alloc_spread"):
            return torch.zeros(shape, dtype=self.obs_dtype, device=self.device)
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=self.obs_dtype).element_size()
        ptr = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self._lib.ppg_alloc_spread(dev_index, nbytes, spread, (id(self) << 4) ^ seed_salt, C.byref(ptr))
        if rc != 0:
            raise RuntimeError(f"ppg_alloc_spread failed ({rc}): {self._lib.ppg_spread_last_error().decode()}")

        class _Holder:   # what torch.as_tensor needs to see device memory it does not own
            pass

        holder = _Holder()
        holder.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr.value, False), "version": 2}
        # THE MAPPING LIVES AS LONG AS THE MEMORY IS REFERENCED: torch keeps `holder` alive until the last tensor (or view, or
        # tensor returned by reset_batch / step_batch) over this storage is gone, and only then are the pages unmapped -- never in
        # close(), which a caller may well outlive with an observation tensor in hand (touching unmapped memory is a GPU fault)
        weakref.finalize(holder, self._lib.ppg_free_spread, C.c_void_p(ptr.value))
        with torch.cuda.device(self.device):
            t = torch.as_tensor(holder, device=self.device).view(self.obs_dtype).view(shape)
            t.zero_()
        return t

    def _create_handle(self, c, create_fn):
        bufs = _abi.PpgBuffers()
        for name in _abi._BUF_FIELDS:
            setattr(bufs, name, getattr(self, name).data_ptr())
        self._handle = C.c_void_p()
        dev_index = self.device.index if self.device.type == "cuda" else 0
        rc = create_fn(C.byref(c), self.batch_size, dev_index, C.byref(bufs), C.byref(self._handle))
        if rc != 0:
            msg = self._lib.ppg_last_error(None).decode()
            self._handle = None
            if rc == -1:
                raise ValueError(msg)  # e.g. "Cannot place more unique positions than grid cells."
            raise RuntimeError(f"ppg_create failed ({rc}): {msg}")
        self.lds_bytes = int(self._lib.ppg_lds_bytes(self._handle))

    # ------------------------------------------------------------------
    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.ppg_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self, stream=None):
        """hipStream_t for a launch: an explicit torch.cuda.Stream / raw handle, else torch's current stream."""
        if self.device.type != "cuda":
            return None
        if stream is None:
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.ppg_last_error(self._handle).decode()}")

    # ------------------------------------------------------------------
    def set_seeds(self, seed):
        """Env b gets Philox key ``seed + b`` (or the given per-env array)."""
        if np.isscalar(seed):
            seeds = (np.arange(self.batch_size, dtype=np.uint64) + np.uint64(int(seed) & (2 ** 64 - 1)))
        else:
            seeds = np.asarray(seed).astype(np.uint64).reshape(self.batch_size)
        self.env_seed.copy_(torch.from_numpy(seeds.view(np.int64)))

    def reset(self, seed=None, episode=0):
        """reset() of every env with on-device Philox placement (predpreygrass_rllib_env.py:129-217)."""
        if seed is not None:
            self.set_seeds(seed)
        self._check(self._lib.ppg_reset(self._handle, None, int(episode), self._stream()), "ppg_reset")
        return self

    def set_placement(self, pred_xy, prey_xy, grass_xy, episode=0):
        """reset() with a given placement (arrays [B,P0,2], [B,Q0,2], [B,n_grass,2] of (x, y)):
        writes the initial state (predpreygrass_rllib_env.py:138-212) and computes the observations."""
        B, G = self.batch_size, self.grid_size
        pred = np.asarray(pred_xy, dtype=np.int64).reshape(B, self.P0, 2)
        prey = np.asarray(prey_xy, dtype=np.int64).reshape(B, self.Q0, 2)
        grass = np.asarray(grass_xy, dtype=np.int64).reshape(B, self.n_grass, 2)
        for a in (pred, prey, grass):
            if a.size and (a.min() < 0 or a.max() >= G):
                raise ValueError("position outside the grid")
        gcell = grass[..., 0] * G + grass[..., 1]
        for b in range(B):
            if len(np.unique(gcell[b])) != self.n_grass:
                raise ValueError("grass positions must be unique")
        # the tables are built by the library (`ppg_reset_from_state`: row tables, OWNS bits in id order, grass table, env words,
        # then one ppg_observe launch)
        arrays = [np.ascontiguousarray((a[..., 0] << 8 | a[..., 1]).astype(np.uint16)) for a in (pred, prey, grass)]
        init = _abi.PpgInitState()
        init.pred_xy, init.prey_xy, init.grass_xy = (a.ctypes.data for a in arrays)
        init.episode = int(episode)
        rc = self._lib.ppg_reset_from_state(self._handle, C.byref(init), self._stream())
        if rc == -1:
            raise ValueError(self._lib.ppg_last_error(self._handle).decode())
        self._check(rc, "ppg_reset_from_state")
        return self

    def set_wave_plan(self, waves=0, helper_min_rows=0, coop_envs=0):
        """Scheduling override (tests, A/B tools; results never depend on it): wavefronts per workgroup of the step kernel
        (0 = automatic), the row count from which helper wavefronts stay, envs per workgroup of the cooperative kernels
        (`ppg_set_wave_plan`)."""
        self._check(self._lib.ppg_set_wave_plan(self._handle, int(waves), int(helper_min_rows), int(coop_envs)), "ppg_set_wave_plan")
        return self

    def wave_plan(self):
        """(wavefronts per workgroup, helper_min_rows, envs per workgroup) ppg_step uses right now."""
        w, m, e = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.ppg_get_wave_plan(self._handle, C.byref(w), C.byref(m), C.byref(e)), "ppg_get_wave_plan")
        return w.value, m.value, e.value

    def step_kernel_name(self) -> str:
        return self._lib.ppg_step_kernel_name(self._handle).decode()

    def rebalance(self, stream=None):
        """Scheduling only (results are unaffected): let envs with many agents start first so that the observation
        writing is spread evenly over the CUs; call every few dozen steps (`ppg_rebalance`)."""
        self._check(self._lib.ppg_rebalance(self._handle, self._stream(stream)), "ppg_rebalance")
        return self

    def observe(self):
        """Recompute the observations of all live rows from the current state tensors."""
        self._check(self._lib.ppg_observe(self._handle, self._stream()), "ppg_observe")
        return self

    def step(self, actions=None, random_actions=False, auto_reset=False, act_rank=None, stream=None):
        """One transition of every env (predpreygrass_rllib_env.py:219-473).

        act_rank (optional uint8 [B,S]): position of each acting row within its type's action
        sequence, for action dicts whose order differs from row order.
        stream (optional): launch on this torch.cuda.Stream instead of torch's current stream."""
        flags = 0
        ptr = None
        if random_actions:
            flags |= _abi.STEP_RANDOM_ACTIONS
        else:
            if actions is None:
                actions = self.actions
            if actions.dtype != torch.int8 or tuple(actions.shape) != (self.batch_size, self.S) or \
                    actions.device != self.device or not actions.is_contiguous():
                raise ValueError(f"actions must be a contiguous int8 tensor [{self.batch_size},{self.S}] on {self.device}")
            ptr = C.c_void_p(actions.data_ptr())
        if auto_reset:
            flags |= _abi.STEP_AUTO_RESET
        if act_rank is not None:
            if random_actions:
                raise ValueError("act_rank cannot be combined with random_actions")
            if act_rank.dtype != torch.uint8 or tuple(act_rank.shape) != (self.batch_size, self.S) or \
                    act_rank.device != self.device or not act_rank.is_contiguous():
                raise ValueError("act_rank must be a contiguous uint8 tensor [B,S] on the env's device")
            self._check(self._lib.ppg_step_ordered(self._handle, ptr, C.c_void_p(act_rank.data_ptr()), flags,
                                                   self._stream(stream)), "ppg_step_ordered")
        else:
            self._check(self._lib.ppg_step(self._handle, ptr, flags, self._stream(stream)), "ppg_step")
        return self

    # SURVEY.md section 8(b)'s spelling of the tensor API: the same calls, returning views of the buffers the kernel wrote
    def reset_batch(self, seeds=None, episode=0):
        """reset() of every env (env b seeded `seeds + b`, or one seed per env); returns (obs_pred, obs_prey)."""
        self.reset(seed=seeds, episode=episode)
        return self.obs_pred, self.obs_prey

    def step_batch(self, actions=None, **kw):
        """step() of every env; returns (obs_pred[B,Sp,C,Rp,Rp], obs_prey[B,Sq,C,Rq,Rq], reward[B,S], terminated[B,S],
        truncated[B,S], live_mask[B,S], agent_id[B,S]).  Row s of env b is in use iff live_mask[b,s]; a row that terminated in
        this call is still in use (it carries its last observation and reward) and is dropped by the next call, like the
        reference keeps a dead agent in `agents` until the next step (predpreygrass_rllib_env.py:222-225,459-461)."""
        self.step(actions, **kw)
        n_pred = self.env_state[:, _abi.ENV_N_PRED_ROWS:_abi.ENV_N_PRED_ROWS + 1]
        n_prey = self.env_state[:, _abi.ENV_N_PREY_ROWS:_abi.ENV_N_PREY_ROWS + 1]
        slot = torch.arange(self.S, device=self.device, dtype=torch.int32).unsqueeze(0)
        live = torch.where(slot < self.pred_capacity, slot < n_pred, slot - self.pred_capacity < n_prey)
        flags = self.row_flags
        return (self.obs_pred, self.obs_prey, self.row_reward, ((flags & _abi.ROW_DIED) != 0) & live,
                ((flags & _abi.ROW_TRUNC) != 0) & live, live, self.row_id)

    def rollout(self, n_steps, actions=None, random_actions=False, auto_reset=False, stream=None):
        """`n_steps` transitions in ONE kernel launch (state stays on chip between steps): the same result
        as `n_steps` calls of step().  Actions come from the device-side uniform random policy
        (`random_actions=True`) or from an action tape `actions` int8 [n_steps, B, S]."""
        flags = (_abi.STEP_RANDOM_ACTIONS if random_actions else 0) | (_abi.STEP_AUTO_RESET if auto_reset else 0)
        ptr = None
        if not random_actions:
            if actions is None or actions.dtype != torch.int8 or tuple(actions.shape) != (n_steps, self.batch_size, self.S) \
                    or actions.device != self.device or not actions.is_contiguous():
                raise ValueError(f"actions must be a contiguous int8 tensor [{n_steps},{self.batch_size},{self.S}] on {self.device}")
            ptr = C.c_void_p(actions.data_ptr())
        self._check(self._lib.ppg_rollout(self._handle, int(n_steps), ptr, flags, self._stream(stream)), "ppg_rollout")
        return self

    def export_grid(self):
        """grid_world_state of every env: float64 [B,4,G,G] (predpreygrass_rllib_env.py:124)."""
        G = self.grid_size
        out = torch.empty((self.batch_size, 4, G, G), dtype=torch.float64, device=self.device)
        self._check(self._lib.ppg_export_grid(self._handle, C.c_void_p(out.data_ptr()), self._stream()), "ppg_export_grid")
        return out

    def export_state(self, b=0) -> bytes:
        """Versioned image of env b's state (`ppg_export_state`): row tables, env words, Philox key, grass table --
        what get_state_snapshot of the reference captures (predpreygrass_rllib_env.py:768-786), as one POD blob."""
        size = C.c_uint64(int(self._lib.ppg_state_bytes(self._handle)))
        blob = C.create_string_buffer(size.value)
        self._check(self._lib.ppg_export_state(self._handle, int(b), blob, C.byref(size), self._stream()), "ppg_export_state")
        return blob.raw[: size.value]

    def import_state(self, blob: bytes, b=0):
        """Write an image from `export_state` into env b (`ppg_import_state`, predpreygrass_rllib_env.py:788-804); the
        geometry of the handle it came from must match.  Call `observe()` afterwards for the observations."""
        buf = C.create_string_buffer(bytes(blob), len(blob))
        rc = self._lib.ppg_import_state(self._handle, int(b), buf, len(blob), self._stream())
        if rc == -1:
            raise ValueError(self._lib.ppg_last_error(self._handle).decode())
        self._check(rc, "ppg_import_state")
        return self

    # ------------------------------------------------------------------
    # the dict classes' host traffic: ONE copy each way per call (`ppg_fetch`, include/ppg.h)
    def _fetch_fields(self):
        """(name, numpy dtype, elements) of an env's record in a fetch image: the state image's field order (include/ppg.h)."""
        S, NG = self.S, self.grass_capacity
        f = [("row_xy", "<i2", S), ("row_energy", "<f8", S), ("row_id", "<i4", S), ("row_key", "<i4", S), ("row_cumrew", "<f8", S),
             ("row_flags", "u1", S), ("row_reward", "<f8", S), ("row_parent", "<i4", S), ("env_state", "<i4", _abi.ENV_WORDS),
             ("env_seed", "<i8", 1), ("grass_xy", "<i2", NG), ("grass_energy", "<f8", NG)]
        if hasattr(self, "walls"):   # BatchedRedQueen (ppg_create_gen2)
            f.append(("row_lastrep", "<i4", S))
            if self.walls:
                f += [("row_info", "u1", S), ("wall_bits", "<i4", int(self.wall_bits.shape[1]))]
        return f

    def _host_buffer(self, nbytes, dtype=torch.uint8):
        """Host memory the library copies into / out of: pinned on a GPU (the copy is then asynchronous), plain for the CPU test build."""
        if self.device.type == "cuda":
            return torch.empty((nbytes,), dtype=dtype).pin_memory()
        return torch.empty((nbytes,), dtype=dtype)

    def fetch(self, env0=0, n=None):
        """What the last call returned for envs [env0, env0 + n), on the host after ONE gather launch, ONE device->host copy and ONE
        stream synchronisation (`ppg_fetch`): (tables, obs_pred, obs_prey) -- tables[name] is the [n, ...] slice of every state tensor
        (a copy), obs_pred[i] / obs_prey[i] the observation blocks IN USE of env env0 + i as arrays [rows, C, R, R] (views into the
        staging buffer: valid until the next fetch)."""
        if self.obs_dtype not in (torch.float64, torch.float32):
            raise RuntimeError("fetch() hands observations to numpy: float64 or float32 rows")
        n = self.batch_size - env0 if n is None else int(n)
        if getattr(self, "_fetch_host", None) is None:
            guess = int(self._lib.ppg_fetch_bytes(self._handle, n, n * min(self.pred_capacity, 24), n * min(self.prey_capacity, 96)))
            self._fetch_host = self._host_buffer(guess)
        for _ in range(3):
            cap = self._fetch_host.numel()
            rc = self._lib.ppg_fetch(self._handle, int(env0), n, C.c_void_p(self._fetch_host.data_ptr()), cap, self._stream())
            if rc == -1 and cap < int(self._lib.ppg_fetch_bytes(self._handle, n, 0, 0)):   # (more envs than the buffer was sized for)
                self._fetch_host = self._host_buffer(int(self._lib.ppg_fetch_bytes(self._handle, n, n * 24, n * 96)))
                continue
            self._check(rc, "ppg_fetch")
            buf = self._fetch_host.numpy()
            H = _abi.PpgFetchHeader.from_buffer(buf[:64])
            if H.magic != _abi.FETCH_MAGIC or H.version != _abi.FETCH_VERSION:
                raise RuntimeError("ppg_fetch returned an image of another version")
            if not H.overflow:
                break
            self._fetch_host = self._host_buffer(int(H.bytes_used) * 3 // 2)
        else:
            raise RuntimeError("ppg_fetch: the image keeps overflowing")
        rec_bytes, bp, bq = int(H.record_bytes), int(H.blk_pred_bytes), int(H.blk_prey_bytes)
        rec = np.array(buf[64:64 + n * rec_bytes]).reshape(n, rec_bytes)   # (a copy: the tables outlive the staging buffer's next use)
        if getattr(self, "_fetch_layout", None) is None:   # (name, dtype, byte offset, bytes) of every field of a record, once
            lay, off = [], 0
            for name, dt, count in self._fetch_fields():
                nb = np.dtype(dt).itemsize * count
                lay.append((name, dt, off, nb))
                off += (nb + 7) // 8 * 8
            self._fetch_layout = lay
        tables = {name: rec[:, off:off + nb].view(dt) for name, dt, off, nb in self._fetch_layout}
        if "row_info" not in tables:
            tables["row_info"] = np.zeros((n, self.S), dtype=np.uint8)
        es = tables["env_state"]
        odt = np.float64 if self.obs_dtype == torch.float64 else np.float32
        pshape, qshape = tuple(self.obs_pred.shape[2:]), tuple(self.obs_prey.shape[2:])
        obs_p, obs_q, o = [], [], 64 + n * rec_bytes
        for kp, kq in zip(es[:, _abi.ENV_N_PRED_ROWS].tolist(), es[:, _abi.ENV_N_PREY_ROWS].tolist()):
            obs_p.append(buf[o:o + kp * bp].view(odt).reshape((kp,) + pshape))
            o += (kp * bp + 15) // 16 * 16
            obs_q.append(buf[o:o + kq * bq].view(odt).reshape((kq,) + qshape))
            o += (kq * bq + 15) // 16 * 16
        return tables, obs_p, obs_q

    def stage_actions(self, env=None):
        """The pinned host mirror of the action tensor (int8 numpy [B,S]) and its upload: fill `stage_actions()` rows, then
        `upload_actions()` -- ONE asynchronous host->device copy for all envs."""
        if getattr(self, "_act_host", None) is None:
            self._act_host = self._host_buffer(self.batch_size * self.S, torch.int8).view(self.batch_size, self.S)
            self._act_host.fill_(_abi.ACTION_NONE)
        a = self._act_host.numpy()
        return a if env is None else a[env]

    def upload_actions(self):
        self.actions.copy_(self._act_host, non_blocking=True)
        return self

    # ------------------------------------------------------------------
    # host views (one device->host copy each; used by tests and by white-box state surgery)
    def host_tables(self, b=None):
        """Small per-env tables copied to the host as numpy arrays."""
        sl = slice(None) if b is None else slice(b, b + 1)
        names = ["row_xy", "row_energy", "row_id", "row_cumrew", "row_flags", "row_reward", "row_parent", "env_state",
                 "grass_xy", "grass_energy"]
        return {n: getattr(self, n)[sl].cpu().numpy() for n in names}

    def records(self, b, tables=None):
        """The returned dicts of env b's last call as an ordered list of
        (name, type, row, reward, terminated, truncated) in the reference's dict order:
        predator survivors, prey survivors, predator newborns, prey newborns."""
        t = tables if tables is not None else self.host_tables(b)
        i = 0 if tables is None else b
        es = t["env_state"][i].tolist()
        nP, nQ = es[_abi.ENV_N_PRED_ROWS], es[_abi.ENV_N_PREY_ROWS]
        newP, newQ = es[_abi.ENV_N_PRED_NEW], es[_abi.ENV_N_PREY_NEW]
        cp = self.pred_capacity
        # (whole rows as Python lists once: indexing numpy scalars one by one costs more than the rest of the dict assembly)
        ids, fls, rws = t["row_id"][i].tolist(), t["row_flags"][i].tolist(), t["row_reward"][i].tolist()
        out = []
        for ty, lo, hi in ((PREDATOR, 0, nP - newP), (PREY, 0, nQ - newQ), (PREDATOR, nP - newP, nP), (PREY, nQ - newQ, nQ)):
            base, fmt = (0, "predator_%d") if ty == PREDATOR else (cp, "prey_%d")
            for r in range(lo, hi):
                fl = fls[base + r]
                out.append((fmt % ids[base + r], ty, r, rws[base + r], bool(fl & _abi.ROW_DIED), bool(fl & _abi.ROW_TRUNC)))
        return out
