"""ctypes view of include/ppg.h and the loader of libppg_hip.so.

The product path has exactly one implementation: the HIP library built for gfx950.
If it is missing or cannot be loaded this module raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libppg_hip.so")

ABI_VERSION = 5

# row_flags bits
ROW_DIED, ROW_OWNS, ROW_NEWBORN, ROW_ATE, ROW_TRUNC, ROW_GRID_E0 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
# env_state words
ENV_WORDS = 20
(ENV_N_PRED_ROWS, ENV_N_PREY_ROWS, ENV_N_PRED_NEW, ENV_N_PREY_NEW, ENV_NEXT_PRED_ID, ENV_NEXT_PREY_ID,
 ENV_STEP, ENV_N_PRED_ALIVE, ENV_N_PREY_ALIVE, ENV_FLAGS, ENV_STATUS, ENV_EPISODE, ENV_FALLBACK_SPAWNS,
 ENV_CALLS, ENV_OBS_PRED, ENV_OBS_PREY, ENV_NEXT_PRED_ID_T2, ENV_NEXT_PREY_ID_T2, ENV_DRAWS) = range(19)
ENVF_TERM_ALL, ENVF_TRUNC_ALL, ENVF_DONE, ENVF_WAS_RESET, ENVF_LIST_IS_ROW_ORDER = 0x01, 0x02, 0x04, 0x08, 0x10
(STATUS_PRED_OVERFLOW, STATUS_PREY_OVERFLOW, STATUS_FALLBACK_SPAWN, STATUS_FAILED_SPAWN,
 STATUS_BAD_ACTION, STATUS_KICK_OVERFLOW, STATUS_UNIFORMS_DRY) = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40
KEY_TYPE2 = 1771561  # row_key offset of second-generation type-2 agents
STEP_RANDOM_ACTIONS, STEP_AUTO_RESET = 0x1, 0x2
ACTION_NONE = -1

_INT_FIELDS = [
    "abi_version", "grid_size", "predator_obs_range", "prey_obs_range", "max_steps", "n_possible_predators",
    "n_possible_prey", "n_initial_predators", "n_initial_prey", "n_grass", "pred_capacity", "prey_capacity",
    "grass_capacity", "obs_dtype",
]
_DBL_FIELDS = [
    "reward_predator_catch_prey", "reward_prey_eat_grass", "reward_predator_step", "reward_prey_step",
    "penalty_prey_caught", "reproduction_reward_predator", "reproduction_reward_prey",
    "energy_loss_per_step_predator", "energy_loss_per_step_prey", "predator_creation_energy_threshold",
    "prey_creation_energy_threshold", "initial_energy_predator", "initial_energy_prey", "initial_energy_grass",
    "energy_gain_per_step_grass",
]
_BUF_FIELDS = [
    "row_xy", "row_energy", "row_id", "row_key", "row_cumrew", "row_flags", "row_reward", "env_state", "env_seed",
    "grass_xy", "grass_energy", "obs_pred", "obs_prey", "row_parent", "row_lastrep", "wall_bits", "row_info",
]
MOVE_REASONS = {1: "wall", 2: "occupied", 3: "corner_cut", 4: "los"}  # row_info - 1 -> move_blocked_reason (WO:466-488)


class PpgConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in _INT_FIELDS] + [(n, C.c_double) for n in _DBL_FIELDS] + [
        ("season_length_steps", C.c_int32), ("season_high_multiplier", C.c_double), ("season_low_multiplier", C.c_double),
        ("reward_mode", C.c_int32), ("kickback", C.c_int32), ("kickback_reward_predator", C.c_double),
        ("kickback_reward_prey", C.c_double),
        # drive-conditioned variant (drive_conditioned_environment/predpreygrass_rllib_env.py:54-89)
        ("n_drive", C.c_int32 * 2), ("drive_kind", (C.c_int32 * 4) * 2), ("hunger_safe_energy", C.c_double * 2),
        ("prey_opportunity_normalizer", C.c_double), ("predator_danger_normalizer", C.c_double),
        ("grass_opportunity_normalizer", C.c_double)]

DRIVE_KINDS = {"hunger_pressure": 0, "reproductive_readiness": 1, "prey_opportunity": 2, "predator_danger_pressure": 3,
               "grass_opportunity": 4}
DEFAULT_PREDATOR_DRIVES = ["hunger_pressure", "reproductive_readiness", "prey_opportunity"]                  # its :56-63
DEFAULT_PREY_DRIVES = ["hunger_pressure", "reproductive_readiness", "predator_danger_pressure", "grass_opportunity"]  # :64-72


# include/ppg.h: struct ppg_config_gen2 (red_queen/predpreygrass_rllib_env.py:28-86 and the config.get calls in step())
GEN2_TYPED = ["reward_predator_catch_prey", "reward_prey_eat_grass", "reward_predator_step", "reward_prey_step",
              "penalty_prey_caught", "reproduction_reward_predator", "reproduction_reward_prey"]
GEN2_SCALARS = [
    "energy_loss_per_step_predator", "energy_loss_per_step_prey", "predator_creation_energy_threshold",
    "prey_creation_energy_threshold", "initial_energy_predator", "initial_energy_prey", "initial_energy_grass",
    "energy_gain_per_step_grass", "move_energy_cost_factor", "max_energy_gain_per_prey", "max_energy_gain_per_grass",
    "max_energy_predator", "max_energy_prey", "max_energy_grass", "energy_transfer_efficiency",
    "reproduction_energy_efficiency", "reproduction_chance_predator", "reproduction_chance_prey",
    "mutation_rate_predator", "mutation_rate_prey",
]


class PpgConfigGen2(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "grid_size", "predator_obs_range", "prey_obs_range", "max_steps")] + [
        ("n_possible", C.c_int32 * 4), ("n_initial", C.c_int32 * 4)] + [
        (n, C.c_int32) for n in ("n_grass", "pred_capacity", "prey_capacity", "grass_capacity", "obs_dtype",
                                 "type_1_action_range", "type_2_action_range", "reproduction_cooldown_steps")] + [
        (n, C.c_double * 2) for n in GEN2_TYPED] + [(n, C.c_double) for n in GEN2_SCALARS] + [
        (n, C.c_int32) for n in ("walls", "include_visibility_channel", "respect_los_for_movement",
                                 "mask_observation_with_visibility")]


PACK_MAGIC, PACK_VERSION, PACK_F32, PACK_MAX_HANDLES = 0x4B475050, 1, 0x1, 8
PACK_NO_OBS = 0x2
STATE_MAGIC, STATE_VERSION = 0x53475050, 1
FETCH_MAGIC, FETCH_VERSION = 0x46475050, 1


class PpgFetchHeader(C.Structure):
    """include/ppg.h: struct ppg_fetch_header (64 bytes, little endian)."""
    _fields_ = [(n, C.c_uint32) for n in ("magic", "version", "env0", "n_envs", "record_bytes", "blk_pred_bytes", "blk_prey_bytes",
                                          "overflow")] + [("bytes_used", C.c_uint64), ("capacity", C.c_uint64),
                                                          ("reserved", C.c_uint32 * 4)]


class PpgPackHeader(C.Structure):
    """include/ppg.h: struct ppg_pack_header (64 bytes, little endian)."""
    _fields_ = [(n, C.c_uint32) for n in ("magic", "version", "n_envs", "n_pred_rows", "n_prey_rows", "obs_elem_bytes",
                                          "blk_pred", "blk_prey")] + [("bytes_used", C.c_uint64), ("capacity", C.c_uint64),
                                                                      ("overflow", C.c_uint32), ("env_words", C.c_uint32),
                                                                      ("reserved", C.c_uint32 * 2)]


def pack_layout(n_envs, n_pred, n_prey, blk_pred, blk_prey, elem):
    """Byte offsets of the sections of a packed observation image (mirror of ppg::pack_layout, csrc/ppg_pack.h)."""
    def al(v):
        return (v + 15) & ~15
    L, o = {}, C.sizeof(PpgPackHeader)
    for name, size in (("env_state", n_envs * ENV_WORDS * 4), ("row_off", n_envs * 8), ("id_pred", n_pred * 4),
                       ("id_prey", n_prey * 4), ("reward_pred", n_pred * 8), ("reward_prey", n_prey * 8),
                       ("flags_pred", n_pred), ("flags_prey", n_prey), ("obs_pred", n_pred * blk_pred * elem),
                       ("obs_prey", n_prey * blk_prey * elem)):
        L[name] = o
        o = al(o + size)
    L["total"] = o
    return L


class PpgBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _BUF_FIELDS]


class PpgInitState(C.Structure):
    """include/ppg.h: struct ppg_init_state (host uint16 arrays, cells as (x << 8) | y)."""
    _fields_ = [("pred_xy", C.c_void_p), ("prey_xy", C.c_void_p), ("grass_xy", C.c_void_p), ("episode", C.c_uint32), ("reserved_", C.c_uint32)]


class PpgPolicyWeights(C.Structure):
    """include/ppg.h: struct ppg_policy_weights (host float32 pointers, PyTorch layouts)."""
    _fields_ = [("conv_w", C.c_void_p * 3), ("conv_b", C.c_void_p * 3), ("fc_w", C.c_void_p * 3), ("fc_b", C.c_void_p * 3)]


POLICY_MAX_CONV, POLICY_MAX_FC = 6, 3


class PpgPolicySpec(C.Structure):
    """include/ppg.h: struct ppg_policy_spec (the general network description of ppg_policy_create_spec)."""
    _fields_ = [("obs_channels", C.c_int32), ("obs_range", C.c_int32), ("n_actions", C.c_int32), ("layout", C.c_int32),
                ("flatten", C.c_int32), ("n_conv", C.c_int32), ("conv_out", C.c_int32 * POLICY_MAX_CONV), ("n_fc", C.c_int32),
                ("fc_out", C.c_int32 * POLICY_MAX_FC),
                ("conv_w", C.c_void_p * POLICY_MAX_CONV), ("conv_b", C.c_void_p * POLICY_MAX_CONV),
                ("fc_w", C.c_void_p * POLICY_MAX_FC), ("fc_b", C.c_void_p * POLICY_MAX_FC)]


POLICY_ARGMAX, POLICY_SAMPLE, POLICY_SEED_ON_DEVICE = 0x0, 0x1, 0x2
POLICY_LAYOUT_CHW, POLICY_LAYOUT_HWC = 0, 1
POLICY_FLATTEN_NCHW, POLICY_FLATTEN_NHWC = 0, 1
POLICY_SYMBOLS = ["ppg_policy_create", "ppg_policy_create_layout", "ppg_policy_create_spec", "ppg_policy_destroy", "ppg_policy_act",
                  "ppg_policy_macs_per_observation", "ppg_policy_last_error", "ppg_policy_describe", "ppg_policy_pack"]
POLICY_PACK_CONV1X, POLICY_PACK_HEAD, POLICY_PACK_SLOTS = 100, 200, 300


SPREAD_SYMBOLS = ["ppg_alloc_spread", "ppg_free_spread", "ppg_spread_stats", "ppg_spread_last_error"]   # HIP library only, like the policy symbols
EXPORTED_SYMBOLS = [
    "ppg_abi_version", "ppg_create", "ppg_destroy", "ppg_get_buffers", "ppg_reset", "ppg_reset_from_state", "ppg_observe", "ppg_step", "ppg_step_many",
    "ppg_rollout", "ppg_step_ordered", "ppg_create_gen2", "ppg_step_uniforms", "ppg_set_envs_in_flight", "ppg_set_wave_plan",
    "ppg_get_wave_plan", "ppg_rebalance",
    "ppg_export_grid", "ppg_walls_changed", "ppg_state_bytes", "ppg_export_state", "ppg_import_state", "ppg_pack_bytes", "ppg_pack",
    "ppg_fetch_bytes", "ppg_fetch",
    "ppg_lexkey", "ppg_lds_bytes", "ppg_step_kernel_name", "ppg_last_error",
] + POLICY_SYMBOLS + SPREAD_SYMBOLS


def bind(lib: C.CDLL) -> C.CDLL:
    """Declare the prototypes of include/ppg.h on a loaded library."""
    lib.ppg_abi_version.restype = C.c_int
    lib.ppg_create.restype = C.c_int
    lib.ppg_create.argtypes = [C.POINTER(PpgConfig), C.c_int32, C.c_int32, C.POINTER(PpgBuffers), C.POINTER(C.c_void_p)]
    lib.ppg_destroy.restype = C.c_int
    lib.ppg_destroy.argtypes = [C.c_void_p]
    lib.ppg_reset.restype = C.c_int
    lib.ppg_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ppg_reset_from_state.restype = C.c_int
    lib.ppg_reset_from_state.argtypes = [C.c_void_p, C.POINTER(PpgInitState), C.c_void_p]
    lib.ppg_observe.restype = C.c_int
    lib.ppg_observe.argtypes = [C.c_void_p, C.c_void_p]
    lib.ppg_step.restype = C.c_int
    lib.ppg_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ppg_step_many.restype = C.c_int
    lib.ppg_step_many.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ppg_rollout.restype = C.c_int
    lib.ppg_rollout.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ppg_step_ordered.restype = C.c_int
    lib.ppg_step_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ppg_create_gen2.restype = C.c_int
    lib.ppg_create_gen2.argtypes = [C.POINTER(PpgConfigGen2), C.c_int32, C.c_int32, C.POINTER(PpgBuffers), C.POINTER(C.c_void_p)]
    lib.ppg_step_uniforms.restype = C.c_int
    lib.ppg_step_uniforms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p]
    lib.ppg_rebalance.restype = C.c_int
    lib.ppg_rebalance.argtypes = [C.c_void_p, C.c_void_p]
    lib.ppg_get_buffers.restype = C.c_int
    lib.ppg_get_buffers.argtypes = [C.c_void_p, C.c_void_p]
    lib.ppg_set_envs_in_flight.restype = C.c_int
    lib.ppg_set_envs_in_flight.argtypes = [C.c_void_p, C.c_int32]
    lib.ppg_set_wave_plan.restype = C.c_int
    lib.ppg_set_wave_plan.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.ppg_get_wave_plan.restype = C.c_int
    lib.ppg_get_wave_plan.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ppg_export_grid.restype = C.c_int
    lib.ppg_export_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ppg_walls_changed.restype = C.c_int
    lib.ppg_walls_changed.argtypes = [C.c_void_p, C.c_void_p]
    lib.ppg_state_bytes.restype = C.c_uint64
    lib.ppg_state_bytes.argtypes = [C.c_void_p]
    lib.ppg_export_state.restype = C.c_int
    lib.ppg_export_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    lib.ppg_import_state.restype = C.c_int
    lib.ppg_import_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.ppg_pack_bytes.restype = C.c_uint64
    lib.ppg_pack_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_uint32]
    lib.ppg_pack.restype = C.c_int
    lib.ppg_pack.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.ppg_fetch_bytes.restype = C.c_uint64
    lib.ppg_fetch_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64]
    lib.ppg_fetch.restype = C.c_int
    lib.ppg_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p]
    if hasattr(lib, "ppg_alloc_spread"):
        lib.ppg_alloc_spread.restype = C.c_int
        lib.ppg_alloc_spread.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]
        lib.ppg_free_spread.restype = C.c_int
        lib.ppg_free_spread.argtypes = [C.c_void_p]
        if hasattr(lib, "ppg_spread_stats"):
            lib.ppg_spread_stats.restype = C.c_int
            lib.ppg_spread_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
        lib.ppg_spread_last_error.restype = C.c_char_p
    if hasattr(lib, "ppg_policy_create"):   # (the MFMA kernels exist in the HIP library only, not in the CPU test build)
        lib.ppg_policy_create.restype = C.c_int
        lib.ppg_policy_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(PpgPolicyWeights), C.POINTER(C.c_void_p)]
        lib.ppg_policy_create_layout.restype = C.c_int
        lib.ppg_policy_create_layout.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PpgPolicyWeights), C.POINTER(C.c_void_p)]
        lib.ppg_policy_create_spec.restype = C.c_int
        lib.ppg_policy_create_spec.argtypes = [C.c_int32, C.POINTER(PpgPolicySpec), C.POINTER(C.c_void_p)]
        lib.ppg_policy_destroy.restype = C.c_int
        lib.ppg_policy_destroy.argtypes = [C.c_void_p]
        lib.ppg_policy_act.restype = C.c_int
        lib.ppg_policy_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ppg_policy_describe.restype = C.c_int
        lib.ppg_policy_describe.argtypes = [C.POINTER(PpgPolicySpec), C.POINTER(C.c_int32), C.c_int32]
        lib.ppg_policy_pack.restype = C.c_int
        lib.ppg_policy_pack.argtypes = [C.POINTER(PpgPolicySpec), C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.ppg_policy_macs_per_observation.restype = C.c_uint64
        lib.ppg_policy_macs_per_observation.argtypes = [C.c_void_p]
        lib.ppg_policy_last_error.restype = C.c_char_p
        lib.ppg_policy_last_error.argtypes = [C.c_void_p]
    lib.ppg_lexkey.restype = C.c_uint32
    lib.ppg_lexkey.argtypes = [C.c_uint32]
    lib.ppg_step_kernel_name.restype = C.c_char_p
    lib.ppg_step_kernel_name.argtypes = [C.c_void_p]
    lib.ppg_lds_bytes.restype = C.c_int32
    lib.ppg_lds_bytes.argtypes = [C.c_void_p]
    lib.ppg_last_error.restype = C.c_char_p
    lib.ppg_last_error.argtypes = [C.c_void_p]
    if lib.ppg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libppg ABI {lib.ppg_abi_version()} != {ABI_VERSION}: rebuild with __graft_entry__.build()")
    return lib


_lib = None


def load_hip_library() -> C.CDLL:
    """Load libppg_hip.so (HIP, gfx950).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        override = os.environ.get("PPG_HIP_LIB")  # A/B experiments with another build of the SAME HIP library
        if override:
            _lib = bind(C.CDLL(override))
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib
