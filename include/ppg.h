/*
 * ppg.h -- C ABI of libppg_hip.so: a batched Predator-Prey-Grass environment
 * whose per-step transition runs as hand-written HIP on MI355X (gfx950).
 *
 * The reference (doesburg11/PredPreyGrass) is pure Python and has no FFI; this
 * is the boundary a maintainer would bind with ctypes to put
 * the HIP path behind the reference's own class
 *   predpreygrass/non_evolutionary/base_environment/predpreygrass_rllib_env.py
 * ("BASE" below).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *  - extern "C", POD structs, fixed-width ints, no torch / C++ types.
 *  - every function returns 0 on success or a negative PPG_E* code and never
 *    throws; ppg_last_error() gives the message for the last failure.
 *  - all device buffers are allocated and owned by the caller (PyTorch-ROCm
 *    tensors); the library borrows the raw pointers for the handle's lifetime
 *    and never frees them.  Library-owned scratch is freed by ppg_destroy().
 *  - all work is asynchronous on the caller's HIP stream (`stream` is a
 *    hipStream_t passed as void*, e.g. torch.cuda.current_stream().cuda_stream).
 *  - one handle belongs to one device and one host thread.
 *
 * State layout (one "row" = one agent slot, kept in the order of the dict that
 * the reference's step() returns, per type):
 *   rows [0, pred_capacity)                       predators
 *   rows [pred_capacity, pred_capacity+prey_cap)  prey
 * Within a type the rows are [survivors in self.agents order..., newborns of
 * the last call in birth order...]; agents that died in the last call keep
 * their row (flag PPG_ROW_DIED) until the next call, exactly like
 * self.agents / _pending_removal in the reference (BASE:222-225,383).
 */
#ifndef PPG_H
#define PPG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPG_ABI_VERSION 5

/* error codes */
#define PPG_OK 0
#define PPG_EINVAL (-1)   /* bad argument / unsupported configuration */
#define PPG_ENOMEM (-2)
#define PPG_EHIP (-3)     /* a HIP runtime call failed */
#define PPG_ENODEV (-4)   /* no usable gfx950 device */

/* row_flags bits (uint8 per row) */
#define PPG_ROW_DIED 0x01u     /* terminations[agent] == True in the last call (BASE:289,332) */
#define PPG_ROW_OWNS 0x02u     /* grid[type, pos] currently holds this agent's energy */
#define PPG_ROW_NEWBORN 0x04u  /* created by the last call (BASE:396-414) */
#define PPG_ROW_ATE 0x08u      /* agent in self.agents_just_ate (BASE:319,362) */
#define PPG_ROW_TRUNC 0x10u    /* truncations[agent] == True in the last call (BASE:232) */
#define PPG_ROW_GRID_E0 0x20u  /* second generation: grid[type, pos] still shows the full initial energy written at birth
                                * (RQ:760) although the agent holds initial_energy * reproduction_energy_efficiency */

/* env_state words (int32 per env) */
#define PPG_ENV_WORDS 20
#define PPG_ENV_N_PRED_ROWS 0   /* rows in use incl. agents that died in the last call */
#define PPG_ENV_N_PREY_ROWS 1
#define PPG_ENV_N_PRED_NEW 2    /* trailing rows that are newborns of the last call */
#define PPG_ENV_N_PREY_NEW 3
#define PPG_ENV_NEXT_PRED_ID 4  /* _next_predator_idx */
#define PPG_ENV_NEXT_PREY_ID 5  /* _next_prey_idx */
#define PPG_ENV_STEP 6          /* current_step */
#define PPG_ENV_N_PRED_ALIVE 7  /* current_num_predators */
#define PPG_ENV_N_PREY_ALIVE 8  /* current_num_prey */
#define PPG_ENV_FLAGS 9
#define PPG_ENV_STATUS 10       /* sticky anomaly bits, see PPG_STATUS_* */
#define PPG_ENV_EPISODE 11
#define PPG_ENV_FALLBACK_SPAWNS 12 /* count of BASE:759-764 events this episode */
#define PPG_ENV_CALLS 13        /* step calls served (diagnostic) */
#define PPG_ENV_OBS_PRED 14     /* predator observations written so far (wraps; for bandwidth accounting) */
#define PPG_ENV_OBS_PREY 15     /* prey observations written so far */
#define PPG_ENV_NEXT_PRED_ID_T2 16 /* second generation: _next_idx[("predator", 2)] (words 4/5 hold type 1) */
#define PPG_ENV_NEXT_PREY_ID_T2 17 /* _next_idx[("prey", 2)] */
#define PPG_ENV_DRAWS 18        /* second generation: uniforms the last call consumed (RQ:701,708) */

/* env_state[PPG_ENV_FLAGS] bits */
#define PPG_ENVF_TERM_ALL 0x01   /* terminations["__all__"] of the last call (BASE:466) */
#define PPG_ENVF_TRUNC_ALL 0x02  /* truncations["__all__"] of the last call (BASE:236) */
#define PPG_ENVF_DONE 0x04       /* episode over: the next auto-reset call resets */
#define PPG_ENVF_WAS_RESET 0x08  /* the last call was a reset, not a transition */
#define PPG_ENVF_LIST_IS_ROW_ORDER 0x10 /* self.agents == row order, not yet sorted (after reset, BASE:143) */

/* env_state[PPG_ENV_STATUS] bits */
#define PPG_STATUS_PRED_OVERFLOW 0x01  /* a birth was dropped: predator rows exhausted */
#define PPG_STATUS_PREY_OVERFLOW 0x02
#define PPG_STATUS_FALLBACK_SPAWN 0x04 /* BASE:759-764 reached (reference is non-deterministic there) */
#define PPG_STATUS_FAILED_SPAWN 0x08   /* BASE:766 reached (reference raises TypeError) */
#define PPG_STATUS_BAD_ACTION 0x10     /* action outside -1..8 (reference: KeyError at BASE:502) */
#define PPG_STATUS_KICK_OVERFLOW 0x20  /* more than 15 kickback rewards for one agent in one step (not representable) */
#define PPG_STATUS_UNIFORMS_DRY 0x40   /* ppg_step_uniforms: an env needed more uniforms than were supplied */

/* ppg_step flags */
#define PPG_STEP_RANDOM_ACTIONS 0x1u /* ignore `actions`; draw uniform actions with Philox4x32-10 on device */
#define PPG_STEP_AUTO_RESET 0x2u     /* envs whose last call ended the episode are reset instead of stepped */

#define PPG_ACTION_NONE (-1) /* agent absent from the action dict: no decay, no move (BASE:244,259) */

/* The config dict of the reference (BASE:20-61; defaults CFG = config_env.py:1-38)
 * plus the capacities of this implementation. */

typedef struct ppg_config {
    int32_t abi_version;          /* PPG_ABI_VERSION */
    int32_t grid_size;            /* BASE:53, >= 2; bounded by 64 KiB of LDS per wavefront (about 80 with 7x7/9x9 windows) */
    int32_t predator_obs_range;   /* BASE:55, 1..15 */
    int32_t prey_obs_range;       /* BASE:56, 1..15 */
    int32_t max_steps;            /* BASE:26 */
    int32_t n_possible_predators; /* BASE:44, <= 999999 */
    int32_t n_possible_prey;      /* BASE:45, <= 999999 */
    int32_t n_initial_predators;  /* BASE:46 */
    int32_t n_initial_prey;       /* BASE:47 */
    int32_t n_grass;              /* BASE:59 */
    int32_t pred_capacity;        /* predator rows per env: 64 */
    int32_t prey_capacity;        /* prey rows per env: 64, 128 or 256 */
    int32_t grass_capacity;       /* >= n_grass, multiple of 64 */
    int32_t obs_dtype;            /* 0: float64 (bit-exact with the reference), 1: float32, 2: bfloat16 (the float64 value rounded to
                                   * float32, then to bfloat16, both to nearest even: compact rows for ppg_policy_act, which stages them
                                   * without conversion -- same logits as from the float64 rows, a quarter of the bytes) */
    double reward_predator_catch_prey;  /* BASE:29 */
    double reward_prey_eat_grass;       /* BASE:30 */
    double reward_predator_step;        /* BASE:31 */
    double reward_prey_step;            /* BASE:32 */
    double penalty_prey_caught;         /* BASE:33 */
    double reproduction_reward_predator;/* BASE:34 */
    double reproduction_reward_prey;    /* BASE:35 */
    double energy_loss_per_step_predator; /* BASE:38 */
    double energy_loss_per_step_prey;     /* BASE:39 */
    double predator_creation_energy_threshold; /* BASE:40 */
    double prey_creation_energy_threshold;     /* BASE:41 */
    double initial_energy_predator;     /* BASE:49 */
    double initial_energy_prey;         /* BASE:50 */
    double initial_energy_grass;        /* BASE:60 (also the regrowth cap, BASE:254) */
    double energy_gain_per_step_grass;  /* BASE:61 */
    /* seasonal grass regrowth (base_environment_seasonal/predpreygrass_rllib_env.py:63-66,224-234,268-271):
     * gain x high for season_length_steps steps, x low for the next, repeating; <= 0 disables (base env) */
    int32_t season_length_steps;
    double season_high_multiplier;
    double season_low_multiplier;
    /* rewards: 0 = base env (BASE:288,322,328,341,365,375,409,438); 1 = net energy delta of the step
     * (project_reward_shaping/base_environment_dense_rewards/predpreygrass_rllib_env.py:242-245,291-292,328-329,440-449);
     * 2 = the same plus the reproduction reward for a parent (.../base_environment_dense_rewards_additive:470) */
    int32_t reward_mode;
    /* kickback variant (project_reward_shaping/base_environment_sparse_rewards_plus_kickback/predpreygrass_rllib_env.py
     * :48-52,86-91,325,364,434,439-449): a grandparent that is still alive is rewarded every time its child reproduces */
    int32_t kickback;                   /* 0 = base env */
    double kickback_reward_predator;
    double kickback_reward_prey;
    /* drive-conditioned variant (drive_conditioned_environment/predpreygrass_rllib_env.py, "DRV"): n_drive[species] extra
     * observation channels, each filled with one per-agent scalar (DRV:54-89,551-616); both 0 = base env.  The observation
     * tensors then have 4 + n_drive[0] (predators) / 4 + n_drive[1] (prey) channels. */
    int32_t n_drive[2];                 /* <= 4 each */
    int32_t drive_kind[2][4];           /* PPG_DRIVE_* per channel, in the order of the reference's *_drive_channels lists */
    double hunger_safe_energy[2];       /* DRV:76-77 */
    double prey_opportunity_normalizer;    /* DRV:78 */
    double predator_danger_normalizer;     /* DRV:79-82 */
    double grass_opportunity_normalizer;   /* DRV:83-86 */
} ppg_config;

#define PPG_DRIVE_HUNGER_PRESSURE 0          /* clip01(1 - energy / hunger_safe_energy), DRV:591-593 */
#define PPG_DRIVE_REPRODUCTIVE_READINESS 1   /* clip01(energy / creation threshold), DRV:595-599 */
#define PPG_DRIVE_PREY_OPPORTUNITY 2         /* clip01(np.sum(window prey channel) / normalizer), DRV:601-602 */
#define PPG_DRIVE_PREDATOR_DANGER 3          /* DRV:604-605 */
#define PPG_DRIVE_GRASS_OPPORTUNITY 4        /* DRV:607-608 */

/* Caller-owned device buffers.  B = batch, S = pred_capacity + prey_capacity,
 * NG = grass_capacity, Rp/Rq = predator/prey obs range. */
typedef struct ppg_buffers {
    uint16_t *row_xy;      /* [B,S]  (x << 8) | y          agent_positions (BASE:111) */
    double *row_energy;    /* [B,S]                        agent_energies (BASE:116) */
    int32_t *row_id;       /* [B,S]  k of "predator_k"/"prey_k" (BASE:71-72) */
    uint32_t *row_key;     /* [B,S]  ppg_lexkey(row_id): order of list.sort() on the id strings (BASE:468) */
    double *row_cumrew;    /* [B,S]                        cumulative_rewards (BASE:63) */
    uint8_t *row_flags;    /* [B,S]  PPG_ROW_* */
    double *row_reward;    /* [B,S]  rewards[agent] of the last call */
    int32_t *env_state;    /* [B,PPG_ENV_WORDS] */
    uint64_t *env_seed;    /* [B]    Philox key of each env (reset placement, random actions, spawn fallback) */
    uint16_t *grass_xy;    /* [B,NG] grass_positions (BASE:114), static after reset */
    double *grass_energy;  /* [B,NG] grass_energies (BASE:117) */
    void *obs_pred;        /* [B,pred_capacity,4,Rp,Rp] float64|float32: observations (BASE:511-526) */
    void *obs_prey;        /* [B,prey_capacity,4,Rq,Rq] */
    int32_t *row_parent;   /* [B,S]  id of the agent's (same-type) parent, -1 = none: agent_parent of the kickback variant */
    int32_t *row_lastrep;  /* [B,S]  second generation: agent_last_reproduction (RQ:115,737,999); may be NULL for ppg_create */
    uint32_t *wall_bits;   /* [B,W]  walls variant: bit (x*G + y) of env b's W = ceil(G*G/32) words set = wall cell (WO:210-250);
                            *        written by the caller, never by the library; NULL unless ppg_config_gen2.walls */
    uint8_t *row_info;     /* [B,S]  walls variant: 0 = the agent did not go through the movement phase of the last call,
                            *        else 1 + move_blocked_reason (PPG_MOVE_*); NULL unless ppg_config_gen2.walls */
} ppg_buffers;

/* move_blocked_reason of the walls variant (WO:466-488), as row_info - 1 */
#define PPG_MOVE_FREE 0
#define PPG_MOVE_WALL 1
#define PPG_MOVE_OCCUPIED 2
#define PPG_MOVE_CORNER_CUT 3
#define PPG_MOVE_LOS 4

/* ---- second generation ("RQ" = predpreygrass/non_evolutionary/red_queen/predpreygrass_rllib_env.py; the same step
 * as walls_occlusion/predpreygrass_rllib_env.py without walls and line of sight) ------------------------------
 * Two agent types per species ("type_<t>_<species>_<k>"), pools in the order reset() creates them:
 * 0 type_1_predator, 1 type_2_predator, 2 type_1_prey, 3 type_2_prey.
 * row_id  = creation_number << 17 | (type - 1) << 16 | k      (creation_number: 0,1,2,... over all agents of the
 *           episode = insertion order of the reference's agent_positions dict, RQ:584-586)
 * row_key = (type - 1) * 1771561 + ppg_lexkey(k)              (order of list.sort() on the id strings, RQ:270)
 * Observations are what the reference returns (float32 grid, RQ:137-139,352-358): use obs_dtype 1. */
typedef struct ppg_config_gen2 {
    int32_t abi_version;
    int32_t grid_size;            /* RQ:70 */
    int32_t predator_obs_range;   /* RQ:72 */
    int32_t prey_obs_range;       /* RQ:73 */
    int32_t max_steps;            /* RQ:36 */
    int32_t n_possible[4];        /* RQ:56-59, pool order; each <= 65535, sum <= 32767 */
    int32_t n_initial[4];         /* RQ:61-64 */
    int32_t n_grass;              /* RQ:76 */
    int32_t pred_capacity;        /* 64 */
    int32_t prey_capacity;        /* 64, 128 or 256 */
    int32_t grass_capacity;
    int32_t obs_dtype;            /* 0: float64, 1: float32 (the reference's dtype), 2: bfloat16, 3: bfloat16 cells (see ppg_config) */
    int32_t type_1_action_range;  /* RQ:85: odd, 1..7 */
    int32_t type_2_action_range;  /* RQ:86: odd, 1..7 (ignored when no type-2 agent can exist) */
    int32_t reproduction_cooldown_steps; /* RQ:696 */
    /* type-specific values, index = type - 1 (_get_type_specific, RQ:1099-1106) */
    double reward_predator_catch_prey[2];
    double reward_prey_eat_grass[2];
    double reward_predator_step[2];
    double reward_prey_step[2];
    double penalty_prey_caught[2];
    double reproduction_reward_predator[2];
    double reproduction_reward_prey[2];
    double energy_loss_per_step_predator;      /* RQ:50 */
    double energy_loss_per_step_prey;          /* RQ:51 */
    double predator_creation_energy_threshold; /* RQ:52 */
    double prey_creation_energy_threshold;     /* RQ:53 */
    double initial_energy_predator;            /* RQ:66 */
    double initial_energy_prey;                /* RQ:67 */
    double initial_energy_grass;               /* RQ:77 */
    double energy_gain_per_step_grass;         /* RQ:78 */
    double move_energy_cost_factor;            /* RQ:305: cost = distance * factor * energy */
    double max_energy_gain_per_prey;           /* RQ:598 (INFINITY = no cap) */
    double max_energy_gain_per_grass;          /* RQ:665 */
    double max_energy_predator;                /* RQ:605 */
    double max_energy_prey;                    /* RQ:672 */
    double max_energy_grass;                   /* RQ:510 */
    double energy_transfer_efficiency;         /* RQ:599,666 */
    double reproduction_energy_efficiency;     /* RQ:754,840 */
    double reproduction_chance_predator;       /* RQ:700-701 */
    double reproduction_chance_prey;
    double mutation_rate_predator;             /* RQ:81,708 */
    double mutation_rate_prey;                 /* RQ:82,793 */
    /* "WO" = walls_occlusion/predpreygrass_rllib_env.py: the same env plus static walls.  walls != 0 selects it:
     * observation channel 0 shows the walls inside the window (WO:536-541) instead of the out-of-grid mask, a move into a
     * wall cell is refused (WO:469-471), ppg_buffers.wall_bits / row_info are used, the observation tensors have
     * 4 + include_visibility_channel channels. */
    int32_t walls;
    int32_t include_visibility_channel;        /* WO:104: last channel = line-of-sight mask (WO:577-598) */
    int32_t respect_los_for_movement;          /* WO:106: no corner cutting, no moves through walls (WO:475-488) */
    int32_t mask_observation_with_visibility;  /* WO:111: channels 1-3 multiplied by the mask (WO:591-594) */
} ppg_config_gen2;

typedef struct ppg_handle ppg_handle;

int ppg_abi_version(void);

/* Validates cfg and binds the buffers.  Replaces PredPreyGrass.__init__ (BASE:18-127). */
int ppg_create(const ppg_config *cfg, int32_t batch, int32_t device, const ppg_buffers *bufs, ppg_handle **out);
int ppg_destroy(ppg_handle *h);
/* The buffers the handle was created with (borrowed device pointers; the caller keeps owning them): what a binding that did
 * not allocate them itself needs to find observations, row tables and the status words. */
int ppg_get_buffers(const ppg_handle *h, ppg_buffers *out);

/* Second generation: replaces PredPreyGrass.__init__ of the red_queen env (RQ:15-86).  bufs->row_lastrep is required.
 * The handle works with ppg_reset / ppg_observe / ppg_step / ppg_step_ordered / ppg_step_many / ppg_export_grid
 * (reproduction uniforms from Philox, keyed like the random actions) and with ppg_step_uniforms; with ppg_rollout on a cooperative
 * four-wave plan (no walls). */
int ppg_create_gen2(const ppg_config_gen2 *cfg, int32_t batch, int32_t device, const ppg_buffers *bufs, ppg_handle **out);

/* reset() (BASE:129-217) for every env: unique random placement (Philox
 * Fisher-Yates keyed by env_seed -- distributionally, not bitwise, the
 * reference's PCG64 + set-order placement), initial energies, observations.
 * seeds: optional device pointer [B]; copied into env_seed first.  episode is
 * the Philox episode counter to start from (normally 0). */
int ppg_reset(ppg_handle *h, const uint64_t *seeds, uint32_t episode, void *stream);

/* reset() (BASE:129-217) of every env with a GIVEN placement -- the `const ppg_init_state*` form of ppg_reset: what
 * the reference's reset(seed) produces once its cells are known (its PCG64 + set-order placement is captured input, not arithmetic:
 * predpreygrass_amd.placement.reference_placement re-does it on the host).  HOST arrays, cells as (x << 8) | y:
 * predators / prey in id order (agent i of its species gets id i and initial_energy_*), grass patches in patch order
 * (initial_energy_grass each).  grid[type, cell] = energy in id order, so of several agents of one species on a cell the last one
 * owns it (BASE:190-200).  Writes the row tables, the grass table and the env words, then computes the observations (one
 * ppg_observe launch); env_seed is left alone (it keys the device-side random actions).  Base-family handles. */
typedef struct ppg_init_state {
    const uint16_t *pred_xy;   /* [B][n_initial_predators] */
    const uint16_t *prey_xy;   /* [B][n_initial_prey] */
    const uint16_t *grass_xy;  /* [B][n_grass], unique within an env */
    uint32_t episode;          /* Philox episode counter to start from (normally 0) */
    uint32_t reserved_;
} ppg_init_state;
int ppg_reset_from_state(ppg_handle *h, const ppg_init_state *init, void *stream);

/* Observations for all live rows from the state currently in the buffers
 * (reset() tail BASE:215; _get_observation BASE:511; used after the host wrote
 * a captured placement or restored a snapshot, BASE:788-804). */
int ppg_observe(ppg_handle *h, void *stream);

/* step(action_dict) (BASE:219-473) for every env.
 * actions: device int8 [B,S], indexed by the row an agent had in the previous
 * call's output (PPG_ACTION_NONE = not in the dict); may be NULL with
 * PPG_STEP_RANDOM_ACTIONS.  The movement phase applies actions in row order,
 * which is the order of the previous observation dict (RLlib's protocol). */
int ppg_step(ppg_handle *h, const int8_t *actions, uint32_t flags, void *stream);

/* n_steps transitions in ONE launch -- exactly what n_steps calls of ppg_step would do, each step writing its observations / rewards /
 * flags / tables to the same buffers -- for action sources that live on the device: the uniform random policy
 * (PPG_STEP_RANDOM_ACTIONS) or an open-loop action tape `actions` = device int8 [n_steps,B,S].  Bit-identical to n_steps ppg_step calls.
 * On a handle whose wave plan is cooperative (a full GPU of 25x25 grids: the default) it runs the fused form of the cooperative step
 * kernel: a workgroup's envs never interact with any other workgroup's, so the workgroups simply run on from step to step at their
 * own pace -- no launch boundary at which everybody computes and nobody stores -- and ONE launch stream reaches 61 us per 4096-env
 * step where per-step launches need three sub-batches in flight for 66-71 (profiles/EXPERIMENTS.md, round 3).  On other handles (small batches,
 * 64x64 grids) it is the round-1 one-wave-per-env loop, which is SLOWER than per-step launches and kept only as a diagnostic.  Base
 * family without the kickback / drive variants; second-generation handles (ppg_create_gen2, no walls) on a cooperative four-wave plan
 * only (PPG_EINVAL otherwise), reproduction uniforms from the device's Philox streams as in ppg_step. */
int ppg_rollout(ppg_handle *h, int32_t n_steps, const int8_t *actions, uint32_t flags, void *stream);

/* Same, for an action dict whose iteration order differs from the previous observation dict
 * (the order matters in the reference: movement and the decay writes are applied in dict order,
 * BASE:244,259).  act_rank: device uint8 [B,S]; for every row with an action, its position among
 * the acting agents OF ITS TYPE (0..n-1, a permutation).  NULL == row order (ppg_step). */
int ppg_step_ordered(ppg_handle *h, const int8_t *actions, const uint8_t *act_rank, uint32_t flags, void *stream);

/* Second generation only: step(action_dict) (RQ:197-299) with the values `self.rng.random()` returns supplied by the
 * caller.  uniforms: device double [B, uniforms_per_env]; env b consumes its row front to back, one value for every
 * agent past its cooldown (chance gate, RQ:701) plus one more for each of those that passes the gate with enough energy
 * (mutation, RQ:708), in self.agents order.  env_state[PPG_ENV_DRAWS] reports how many were used;
 * PPG_STATUS_UNIFORMS_DRY is raised if the row was too short.  act_rank as in ppg_step_ordered (NULL = row order). */
int ppg_step_uniforms(ppg_handle *h, const int8_t *actions, const uint8_t *act_rank, const double *uniforms,
                      int32_t uniforms_per_env, uint32_t flags, void *stream);

/* ppg_step for n handles (sub-batches of one GPU's envs), handle k on streams[k], in one host call.
 * actions may be NULL (with PPG_STEP_RANDOM_ACTIONS) or an array of n device pointers. */
int ppg_step_many(ppg_handle *const *handles, int32_t n, const int8_t *const *actions, uint32_t flags,
                  void *const *streams);

/* Scheduling hint: this handle's launches overlap with launches of other handles on the same GPU (sub-batches on other
 * streams); envs_in_flight = envs of all of them together (default: the handle's own batch).  Used to choose between the
 * one-wave-per-env step kernel and the four- / eight-waves-per-env ones (wave 0 steps, all waves write the final observations),
 * which are faster while the GPU is not full: eight up to 512 envs in flight, four up to about 3072 or when LDS admits at most
 * 4 envs per CU. */
int ppg_set_envs_in_flight(ppg_handle *h, int32_t envs_in_flight);

/* Scheduling override (tests, A/B tools): which step kernel ppg_step launches for this handle.  waves = wavefronts per workgroup
 * (0 = back to the automatic choice; 1, 2, 4, 8, 16), helper_min_rows = helper wavefronts only stay for envs with at least that
 * many agent rows, coop_envs = envs sharing one workgroup (cooperative kernels: every wave of the workgroup runs one env's
 * transition, then all of them write all the workgroup's observations; 0 = one env per workgroup).  Combinations a variant has
 * no kernel for fall back to the nearest one it has.  Results never depend on the plan -- the GPU tests compare them bit for
 * bit.  The plan is fixed here, at ppg_create and at ppg_set_envs_in_flight: ppg_step itself reads no environment variables. */
int ppg_set_wave_plan(ppg_handle *h, int32_t waves, int32_t helper_min_rows, int32_t coop_envs);
int ppg_get_wave_plan(const ppg_handle *h, int32_t *waves, int32_t *helper_min_rows, int32_t *coop_envs);

/* Walls variant: tell the library that the caller has (re)written wall_bits.  It recomputes, for every cell of every env, the
 * line-of-sight mask over the observation window (one HIP launch; library-owned [B, G*G, words] in HBM) -- from then on
 * observations read a few mask words per agent instead of walking one Bresenham line per window cell (WO:492-525, 577-589).
 * Results are identical with or without it as long as it is called after every change of wall_bits (ppg_import_state does).
 * It also copies the bitmaps to the host and WAITS for the stream: when every env's bitmap equals env 0's (one wall layout for the
 * batch) all envs read env 0's masks from then on (a few KB that stay in L2). */
int ppg_walls_changed(ppg_handle *h, void *stream);

/* Scheduling only, results are unaffected: recompute the order in which the handle's envs are assigned to workgroups --
 * envs with many agent rows (much observation data to write) first, so that the load is spread evenly over the CUs.
 * Stream-ordered like a step (one small launch: a counting sort in LDS, any batch size); populations drift slowly, calling it
 * every few dozen steps is enough (bench.py: every 64).  The order is used by every later launch of the handle. */
int ppg_rebalance(ppg_handle *h, void *stream);

/* grid_world_state (BASE:124): dense float64 [B,4,G,G] rebuilt from the rows. */
int ppg_export_grid(ppg_handle *h, double *grid_out, void *stream);

/* ---- snapshot of one env (get_state_snapshot / restore_state_snapshot, BASE:768-804; consumers
 * evaluate_ppo_from_checkpoint_debug.py:164-182) ------------------------------------------------------------
 * A snapshot is a versioned POD image in HOST memory: struct ppg_state_header followed by env `env`'s slice of every
 * state tensor of ppg_buffers in the order of ppg_state_field[] below (row tables, env words, Philox key + episode,
 * grass table, and for the variants row_parent / row_lastrep / row_info / wall_bits).  Observations, rewards of the last
 * call included, are NOT part of it: ppg_observe() recomputes the observations of the live rows after an import.
 * Both calls are synchronous: they enqueue the copies on `stream` and wait for them. */
#define PPG_STATE_MAGIC 0x53475050u /* "PPGS" */
#define PPG_STATE_VERSION 1u
typedef struct ppg_state_header {
    uint32_t magic, version;
    uint32_t bytes;            /* size of the whole image incl. this header */
    uint32_t gen2, walls;      /* which ppg_create* made the handle */
    uint32_t grid_size, pred_capacity, prey_capacity, grass_capacity, n_wall_words;
    uint32_t reserved[6];
} ppg_state_header;

/* bytes an image of one env of this handle takes */
uint64_t ppg_state_bytes(const ppg_handle *h);
/* env `env` -> blob (host memory, *size bytes available; on return *size = bytes written).  blob == NULL: only report the size. */
int ppg_export_state(ppg_handle *h, int32_t env, void *blob, uint64_t *size, void *stream);
/* blob -> env `env`; the image must come from a handle with the same geometry (checked through the header). */
int ppg_import_state(ppg_handle *h, int32_t env, const void *blob, uint64_t size, void *stream);

/* ---- packed observation image (ONE collective per step for the returned observation dict) --------
 * ppg_pack writes what the last call of the n handles (sub-batches of one GPU, envs concatenated in handle order)
 * returned -- env words, per-row ids / rewards / flags and the observations of the rows IN USE, compacted -- into one
 * contiguous device buffer that a single all-gather (RCCL) can move.  Layout, every section 16-byte aligned, in this order:
 *   ppg_pack_header | env_state int32[n_envs][PPG_ENV_WORDS] | row_off uint32[n_envs][2] (exclusive prefix sums of the
 *   predator / prey row counts) | id_pred int32[Np] | id_prey int32[Nq] | reward_pred double[Np] | reward_prey double[Nq] |
 *   flags_pred uint8[Np] | flags_prey uint8[Nq] | obs_pred elem[Np][blk_pred] | obs_prey elem[Nq][blk_prey]
 * (Np / Nq = rows in use over all envs, env-major, row order = dict order).  If the image does not fit in `capacity`
 * bytes the header has overflow = 1 and bytes_used = the size it needs; the row sections are then not written. */
#define PPG_PACK_MAGIC 0x4B475050u /* "PPGK" */
#define PPG_PACK_VERSION 1u
#define PPG_PACK_F32 0x1u /* float64 observations travel as float32 (observations that already are float32 are copied) */
#define PPG_PACK_NO_OBS 0x2u /* no observation sections (header blk_pred = blk_prey = 0): env words, row offsets, ids, rewards and
                              * flags only, 13 bytes per agent row + 88 per env -- what a learner or logger needs from a rollout whose
                              * policy runs next to the env (ppg_policy_act): ~2 MB per 4096-env shard and step instead of ~370 MB */
#define PPG_PACK_MAX_HANDLES 8
typedef struct ppg_pack_header {
    uint32_t magic, version;
    uint32_t n_envs, n_pred_rows, n_prey_rows;
    uint32_t obs_elem_bytes;       /* 4 or 8 */
    uint32_t blk_pred, blk_prey;   /* elements per observation: channels * R * R */
    uint64_t bytes_used, capacity;
    uint32_t overflow, env_words;
    uint32_t reserved[2];
} ppg_pack_header;                 /* 64 bytes */

/* size of an image with the given totals (host helper; h gives the geometry) */
uint64_t ppg_pack_bytes(const ppg_handle *h, int32_t n_envs, int64_t n_pred_rows, int64_t n_prey_rows, uint32_t flags);
/* asynchronous on `stream`, which the caller has ordered behind the handles' last step (two small launches) */
int ppg_pack(ppg_handle *const *handles, int32_t n, void *out, uint64_t capacity, uint32_t flags, void *stream);

/* ---- host view of the last call (what PredPreyGrass.step() hands back, BASE:219,456-473) -------------------------
 * The reference's step() returns Python dicts of one env; the dict classes of this library rebuild them from the row
 * tables and the observation rows IN USE.  ppg_fetch brings exactly that to the host in ONE image -- one gather launch
 * into a library-owned staging buffer, one asynchronous copy into `host` (pinned memory keeps it asynchronous), one
 * stream synchronisation -- for the envs [env0, env0 + n_envs) of a handle:
 *   ppg_fetch_header | n_envs x record | observation sections, env by env
 * record (record_bytes each) = the env's slice of every state tensor in the order of the state image (ppg_export_state):
 *   row_xy u16[S] | row_energy f64[S] | row_id i32[S] | row_key i32[S] | row_cumrew f64[S] | row_flags u8[S] |
 *   row_reward f64[S] | row_parent i32[S] | env_state i32[PPG_ENV_WORDS] | env_seed u64 | grass_xy u16[NG] |
 *   grass_energy f64[NG] | second generation: row_lastrep i32[S] | walls: row_info u8[S], wall_bits u32[n_wall_words]
 * each slice padded to 8 bytes, the record to 16.  Observation section of an env = its predator blocks in use
 * (env word PPG_ENV_N_PRED_ROWS of them, blk_pred_bytes each), padded to 16 bytes, then its prey blocks likewise, in the
 * handle's observation dtype.  If the image does not fit `capacity`, header.overflow = 1, bytes_used = the size it needs
 * and only header + records are valid.  Synchronous: the image is complete when the call returns. */
#define PPG_FETCH_MAGIC 0x46475050u /* "PPGF" */
#define PPG_FETCH_VERSION 1u
typedef struct ppg_fetch_header {
    uint32_t magic, version;
    uint32_t env0, n_envs;
    uint32_t record_bytes;
    uint32_t blk_pred_bytes, blk_prey_bytes;
    uint32_t overflow;
    uint64_t bytes_used, capacity;
    uint32_t reserved[4];
} ppg_fetch_header;                /* 64 bytes */

/* size of an image of n_envs envs with the given row totals (upper bound: every run of blocks padded to 16 bytes) */
uint64_t ppg_fetch_bytes(const ppg_handle *h, int32_t n_envs, int64_t n_pred_rows, int64_t n_prey_rows);
int ppg_fetch(ppg_handle *h, int32_t env0, int32_t n_envs, void *host, uint64_t capacity, void *stream);

/* ---- policy inference next to the env (base_environment/tune_ppo_base_environment.py:106-141) ----------
 * The reference trains two PPO policies (predator_policy / prey_policy) with RLlib's DefaultPPOTorchRLModule and
 * model_config {conv_filters [[16,[3,3],1],[32,[3,3],1],[64,[3,3],1]], fcnet_hiddens [256,256], fcnet_activation relu}.
 * WHAT RLLIB BUILDS FROM THAT (pinned by the checkpoint the reference tree holds, .../shared_prey/experiments/PPO_v_APPO/.../
 * checkpoint_000099/learner_group/learner/rl_module/type_1_predator/module_state.pkl, ray 2.52.1; tests/golden/rllib_checkpoint/):
 * a 3-D Box is read channels-LAST -- the (C,R,R) observation is an image of C rows x R columns with R channels -- by a CNN encoder
 * of one ZeroPad2d + Conv2d(3x3, stride 1) + ReLU per conv_filters entry (keys encoder.actor_encoder.net.0.cnn.{1,4,7,...}),
 * whose output is permuted back to channels-last and flattened ([row][column][channel]); `fcnet_hiddens` is IGNORED for image
 * observations and the policy head is ONE Linear(flat -> n_actions) (pi.net.mlp.0) unless `head_fcnet_hiddens` is set.
 * ppg_policy_create_spec takes that network (and its variations: 1-6 convolutions, 0-2 hidden head layers, either image
 * reading, either flatten order); ppg_policy_act evaluates it for EVERY row in use of the handles' last call, reading the
 * observation rows in place (obs_pred / obs_prey of ppg_buffers) and writing the chosen action into actions[b][slot] -- the
 * observations never leave the GPU and what a consumer has to move per agent is one byte.  Arithmetic: bf16 operands, fp32
 * accumulation, on the matrix cores (v_mfma_f32_32x32x16_bf16 / 16x16x32); activations are rounded to bf16 between layers.
 * RLlib itself is not importable here: parity is pinned against a float32 PyTorch module with RLlib's parameter names and
 * shapes (predpreygrass_amd.policy.PolicyNet, loaded strictly from the real checkpoint) within the tolerance the tests state. */
typedef struct ppg_policy_weights {   /* HOST pointers, float32, PyTorch layouts */
    const float *conv_w[3];  /* Conv2d.weight [cout][cin][3][3]: cin = C / 16 / 32, cout = 16 / 32 / 64 */
    const float *conv_b[3];  /* Conv2d.bias [cout] */
    const float *fc_w[3];    /* Linear.weight [out][in]: [256][64*P] (input flattened channel-major), [256][256], [n_actions][256] */
    const float *fc_b[3];    /* Linear.bias [out] */
} ppg_policy_weights;

/* How the network reads the (4,R,R) observation as an image with C channels and P positions:
 *   PPG_POLICY_LAYOUT_CHW  channel-first: an R x R image with C = 4 channels (P = R*R) -- conv1 weight [16][4][3][3];
 *   PPG_POLICY_LAYOUT_HWC  channels-last: a 4 x R image with C = R channels (P = 4*R) -- conv1 weight [16][R][3][3].  This is how
 *                          RLlib's CNN encoder reads a 3-D Box (its observation spaces are [H, W, C]), i.e. what a module trained by
 *                          tune_ppo_base_environment.py:106-141 on Box(0, 100, (4,R,R)) holds.
 * The conv1 weight shape of a checkpoint tells the two apart (predpreygrass_amd.policy.load_rllib_state_dict). */
#define PPG_POLICY_LAYOUT_CHW 0
#define PPG_POLICY_LAYOUT_HWC 1

typedef struct ppg_policy ppg_policy;

#define PPG_POLICY_ARGMAX 0x0u  /* action = argmax of the logits (first maximum) */
#define PPG_POLICY_SAMPLE 0x1u  /* action ~ softmax(logits): Gumbel-max with Philox4x32-10 keyed by (seed, env, row) */
#define PPG_POLICY_SEED_ON_DEVICE 0x2u /* `seed` is the address of a uint64 in DEVICE memory that the kernel reads when it runs: a step
                                        * captured into a hipGraph (policy + ppg_step + an increment of that word) can then be replayed
                                        * with a fresh key every time.  Pipeline networks (the reference's) with both policies given. */

/* How the convolution output [channel][position] is flattened into the first Linear layer's input:
 *   PPG_POLICY_FLATTEN_NCHW  torch.flatten of a channel-first tensor: feature = channel * P + position
 *   PPG_POLICY_FLATTEN_NHWC  RLlib: TorchCNN permutes its output back to channels-last before nn.Flatten: feature = position * C + channel */
#define PPG_POLICY_FLATTEN_NCHW 0
#define PPG_POLICY_FLATTEN_NHWC 1
#define PPG_POLICY_MAX_CONV 6
#define PPG_POLICY_MAX_FC 3

typedef struct ppg_policy_spec {   /* weights: HOST pointers, float32, PyTorch layouts */
    int32_t obs_channels;   /* C of the (C,R,R) observation rows: 4; 5 with the walls variant's visibility channel; <= 8 */
    int32_t obs_range;      /* R <= 15 */
    int32_t n_actions;      /* <= 32 */
    int32_t layout;         /* PPG_POLICY_LAYOUT_* */
    int32_t flatten;        /* PPG_POLICY_FLATTEN_* */
    int32_t n_conv;         /* 1..PPG_POLICY_MAX_CONV convolutions 3x3, stride 1, "same" zero padding, ReLU */
    int32_t conv_out[PPG_POLICY_MAX_CONV];   /* output channels per layer: <= 16, <= 32, then <= 64 (multiples of 8) */
    int32_t n_fc;           /* 1..PPG_POLICY_MAX_FC Linear layers behind the flatten; the last one gives the logits, the others ReLU */
    int32_t fc_out[PPG_POLICY_MAX_FC];       /* hidden widths (<= 256), then n_actions */
    const float *conv_w[PPG_POLICY_MAX_CONV];  /* Conv2d.weight [cout][cin][3][3] */
    const float *conv_b[PPG_POLICY_MAX_CONV];  /* Conv2d.bias [cout] */
    const float *fc_w[PPG_POLICY_MAX_FC];      /* Linear.weight [out][in] */
    const float *fc_b[PPG_POLICY_MAX_FC];      /* Linear.bias [out] */
} ppg_policy_spec;

/* The general form.  n_fc == 1 (what RLlib builds: no hidden head layer) runs entirely in LDS -- one persistent launch per species
 * whose workgroups take the convolutions and the head of a few samples at a time; n_fc >= 2 needs n_conv == 3 with 16/32/64 channels
 * (the hidden layers are zero-padded to 256 features).  PPG_EINVAL with a message names anything else. */
int ppg_policy_create_spec(int32_t device, const ppg_policy_spec *spec, ppg_policy **out);
/* obs_range: the R of the species' (4,R,R) observations; n_actions <= 32.  The weights are repacked into MFMA fragment
 * order (bf16) on the device; the host arrays may be freed afterwards.  These two are ppg_policy_create_spec with 4 channels, three
 * convolutions 16/32/64, two hidden layers of 256 and the NCHW flatten (rounds 2-3's network; kept for its callers). */
int ppg_policy_create(int32_t device, int32_t obs_range, int32_t n_actions, const ppg_policy_weights *w, ppg_policy **out);
int ppg_policy_create_layout(int32_t device, int32_t obs_range, int32_t n_actions, int32_t layout, const ppg_policy_weights *w,
                             ppg_policy **out);
int ppg_policy_destroy(ppg_policy *p);
/* One forward pass of `pred` over the predator rows in use and of `prey` over the prey rows in use of all n handles (either
 * may be NULL: that species keeps whatever actions[] holds).  actions[k]: device int8 [B_k, S] of handle k.
 * logits_pred / logits_prey: optional device float [rows in use over all envs][n_actions], env-major in row order (the
 * order of ppg_pack); NULL = not wanted.  seed: Philox key of PPG_POLICY_SAMPLE (use a fresh value per step).
 * Asynchronous on `stream`, which the caller has ordered behind the handles' last step. */
int ppg_policy_act(ppg_policy *pred, ppg_policy *prey, ppg_handle *const *handles, int32_t n, int8_t *const *actions,
                   uint32_t flags, uint64_t seed, float *logits_pred, float *logits_prey, void *stream);
/* multiply-accumulates one observation costs in this network (for FLOP accounting: 2 flops each) */
uint64_t ppg_policy_macs_per_observation(const ppg_policy *p);
const char *ppg_policy_last_error(const ppg_policy *p);
/* Which kernels ppg_policy_create_spec would pick for this network and how they lay out a CU's 160 KB of LDS -- computed without a
 * device and without the weights (the pointers of `spec` are not read); for tests and for sizing.  Fills up to n of:
 *   out[0] kernel family: 0 FC chain (hidden head layers), 1 one-role direct-head kernels, 2 the same for more than three
 *          convolutions, 3 the two-role pipeline (three convolutions, <= 16 actions)
 *   out[1] samples per sub-group   out[2] dynamic LDS bytes per workgroup   out[3] threads per workgroup
 *   family 3 only: out[4] samples a workgroup's table holds, out[5] bytes of a sample's region, out[6] 8-byte row chunks a thread
 *   fetches per sub-group (0: one load per channel), out[7] samples covered per chunk load, out[8..10] byte offsets of the
 *   partial-sum area, the row area and the images, out[11] bytes of the row area
 * Returns PPG_EINVAL for a spec ppg_policy_create_spec would reject on its shape.  (Diagnostic: with the environment variable
 * PPG_POLICY_PIPE=0 set when a policy is created, a network of family 3 gets the kernels of family 1 -- same logits, bit for bit.) */
int ppg_policy_describe(const ppg_policy_spec *spec, int32_t *out, int32_t n);
/* The weight fragments ppg_policy_create_spec uploads for one layer, in MFMA operand order, as bfloat16 bit patterns -- computed
 * without a device, so that the repacking (which lane holds which weight of which k-step) can be checked on the CPU against a plain
 * convolution before a kernel ever runs (tests/test_policy.py).  what: 0 .. n_conv - 1 = that convolution for
 * v_mfma_f32_32x32x16_bf16, [row tile][k-step][lane][8]; PPG_POLICY_PACK_CONV1X = the first convolution as the two-role pipeline
 * takes it (v_mfma_f32_16x16x32_bf16, [kernel row][lane][8]; up to nine input channels); PPG_POLICY_PACK_HEAD = the single Linear head
 * of a network without hidden head layers (v_mfma_f32_16x16x32_bf16, [action tile][k-step][lane][8]).  *n_words = words the layer
 * takes; out == NULL or capacity too small: only the size is reported (PPG_OK). */
#define PPG_POLICY_PACK_CONV1X 100
#define PPG_POLICY_PACK_HEAD 200
#define PPG_POLICY_PACK_SLOTS 300 /* not weights: the two-role pipeline's slot table, one word per slot of a sub-group = sample << 8 |
                                   * position (0xFFFF: empty) -- which position a lane of an MFMA tile computes; ordered so that a tile's
                                   * cells fall into all LDS bank groups (csrc/ppg_policy.h: ppg_slot_table) */
int ppg_policy_pack(const ppg_policy_spec *spec, int32_t what, uint16_t *out, uint64_t capacity, uint64_t *n_words);

/* A device buffer for the caller-owned observation tensors whose physical pages are picked at random from a stretch of device
 * memory `spread` times its size (HIP virtual memory management: spread x as many 2 MB chunks are created, a random subset is
 * mapped in random order, the rest is given back at once).  Optional -- any device pointer works as obs_pred / obs_prey -- but where
 * the pages of those two tensors lie decides how fast HBM takes the step's scattered 1 KB pieces: 62-64 us per 4096-env step with
 * spread 32-64 against 62-91 us for what hipMalloc happens to return and 105-121 us for physically contiguous memory (profiles/EXPERIMENTS.md,
 * round 3).  Costs: spread 32 at 1.8 GB takes about 2.5 s and 58 GB of transient device memory; if the device cannot hold the
 * pool the spread shrinks, and the transient pool never takes more than half of the memory that is free when it is built.  Not in the
 * CPU test build.  ppg_free_spread gives the physical memory back and RETIRES the virtual range
 * (a re-used range was seen to go through stale translations).  Returns PPG_OK or an error code (ppg_spread_last_error()). */
int ppg_alloc_spread(int32_t device, uint64_t bytes, int32_t spread, uint64_t seed, void **out);
int ppg_free_spread(void *ptr);
/* what this process holds / has given up: bytes mapped now, and the virtual ranges (count, bytes) ppg_free_spread has retired.  Any
 * pointer may be NULL. */
int ppg_spread_stats(uint64_t *live_bytes, uint64_t *retired_ranges, uint64_t *retired_bytes);
const char *ppg_spread_last_error(void);   /* of the calling thread */

/* Sort key of the decimal string of `id` (digits d: sum (d+1)*11^(5-pos)); host helper. */
uint32_t ppg_lexkey(uint32_t id);

/* Name of the kernel ppg_step launches for this handle right now (it depends on the variant, the row capacity and -- through
 * ppg_set_envs_in_flight -- on how full the GPU is: one, two, four or eight wavefronts per env).  Diagnostic: profiles name it. */
const char *ppg_step_kernel_name(ppg_handle *h);

/* Dynamic LDS bytes per wavefront the step kernel uses for this handle (diagnostic). */
int32_t ppg_lds_bytes(const ppg_handle *h);

const char *ppg_last_error(const ppg_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* PPG_H */
