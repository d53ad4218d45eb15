// ppg_kernel.h -- the PredPreyGrass transition as hand-written HIP for gfx950 (MI355X).
//
// ONE 64-LANE WAVEFRONT STEPS ONE ENVIRONMENT.  Lane l holds agent row l of each row
// register (register 0: predator rows 0..63; register 1+q: prey rows 64q..64q+63), so the
// whole agent table lives in VGPRs and the per-agent flags live in 64-bit SGPR masks.
// Order-dependent phases of the reference (movement in action order, engagement in
// self.agents order, spawning) run as wave-uniform scalar loops over mask bits that read
// a row with v_readlane and test cell occupancy with one v_cmp + ballot; order-free
// phases (decay, grass regrowth, observation extraction, reward assembly) are lane-parallel.
// The multi-wave step kernels (NW = 4 / 8) add helper wavefronts to the workgroup that do nothing
// but share the final observation writing once wave 0 has finished the transition.
//
// The reference's dense float64 grid (4,G,G) is never materialised.  Every non-zero write
// the reference makes to grid[1|2] stores the writer's *current* energy at the writer's
// own cell, and every energy change is followed by such a write (BASE:247-250,269-273,
// 325,368,405-406), so   grid[type][cell] == energy[owner(cell)]  or 0.   The kernel
// therefore carries one OWNS bit per row ("my cell's grid value is mine") and reproduces
// the reference's ghost-cell behaviour exactly through that bit.  LDS only
// holds an acceleration structure for observation extraction: u16 cell->value-index maps
// per channel plus a float64 value table.
//
// "BASE:n" = line n of predpreygrass/non_evolutionary/base_environment/predpreygrass_rllib_env.py
// in the reference.  Correctness is checked bit-for-bit against oracle/ppg_oracle.c.
#pragma once

#include <stdint.h>

#include "../../include/ppg.h"

#ifndef PPG_WAVE_EMU
#include "wave.h"
#endif

// Diagnostic build only (-DPPG_PROFILE_PHASES, tools/phase_profile.py): per-env shader-clock stamps
// at phase boundaries, written to a buffer that nothing else reads.  Never defined in the product.
#ifdef PPG_PROFILE_PHASES
#define PPG_STAMP(i) do { if (C.prof) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); if (ln == 0) C.prof[(size_t)b * 16 + (i)] = t_; } } while (0)
#else
#define PPG_STAMP(i) do { } while (0)
#endif

namespace ppg {

enum { MODE_STEP = 0, MODE_RESET = 1, MODE_OBSERVE = 2, MODE_EXPORT_GRID = 3, MODE_STEP_ORDERED = 4, MODE_ROLLOUT = 5,
       MODE_STEP_KICK = 6, MODE_STEP_ORDERED_KICK = 7, N_MODES = 8,
       MODE_VIS = 8 };   // walls variant only: (re)compute the per-cell line-of-sight masks (ppg_walls_changed)

// event bits of a row during one call (bits 8-15: kickback counters).  Second generation: EV_REPRO = a child was
// actually created (EV_PARENT alone = reward without a free id, RQ:715-725); EV_TURN = the prey had its own engagement
// turn before a predator of a later class caught it (RQ:225-233 runs in self.agents order, types interleaved).
enum { EV_STARVED = 1, EV_CAUGHT = 2, EV_ATE = 4, EV_PARENT = 8, EV_BORN = 16, EV_TRUNC = 32, EV_REPRO = 64, EV_TURN = 128 };

// walls variant: move_blocked_reason (WO:466-488) as kept in bits 0-2 of Env::keep / in row_info: 0 = the agent did not go
// through the movement phase, otherwise 1 + code
enum { MV_NONE = 1, MV_WALL = 2, MV_OCCUPIED = 3, MV_CORNER_CUT = 4, MV_LOS = 5 };

constexpr uint32_t TAG_ACT = 0x41435431u;  // Philox key domains (see oracle/ppg_oracle.c)
constexpr uint32_t TAG_RST = 0x52535431u;
constexpr uint32_t TAG_SPW = 0x53505731u;
constexpr uint32_t TAG_REP = 0x52455031u;  // second generation: reproduction uniforms
constexpr uint32_t KEY_TYPE2 = 1771561u;   // 11^6: row_key offset of type-2 agents ("type_2_..." sorts after "type_1_...")


struct KParams {
    // config
    int32_t G, Rp, Rq, max_steps;
    int32_t npos_pred, npos_prey, n_init_pred, n_init_prey, n_grass;
    int32_t cap_pred, cap_prey, cap_grass, S;
    int32_t obs_f32;   // observation dtype: 0 float64, 1 float32, 2 bfloat16 (ppg_config.obs_dtype)
    uint32_t g_magic;  // ceil(2^32 / G): cell / G == mulhi(cell, g_magic) for cell < G*G
    double r_catch, r_eat, r_pstep, r_qstep, r_caught, r_repro_p, r_repro_q;
    double loss_p, loss_q, thr_p, thr_q, e0_p, e0_q, e0_g, gain_g;
    double season_hi, season_lo;  // seasonal regrowth multipliers
    int32_t season_len;           // <= 0: no seasonal cycle
    int32_t reward_mode;          // 0 base rewards, 1 dense energy delta, 2 dense + reproduction bonus
    double kick_p, kick_q;        // kickback rewards (grandparent bonus)
    int32_t kickback;             // 0 = base env
    int32_t gen2;                 // 1 = second generation (ppg_config_gen2); everything below up to the LDS layout is its config
    int32_t npos2[4], ninit2[4];  // pool order: type_1_predator, type_2_predator, type_1_prey, type_2_prey
    int32_t ar[2];                // action range per type (RQ:85-86)
    uint32_t ar_inv[2];           // ceil(65536 / range): a / range == (a * inv) >> 16 for a < 49
    int32_t cooldown;             // RQ:696
    int32_t uniforms_per_env;
    double r2_catch[2], r2_eat[2], r2_pstep[2], r2_qstep[2], r2_caught[2], r2_repro_p[2], r2_repro_q[2];  // by type
    double move_factor, cap_gain_prey, cap_gain_grass, max_e_pred, max_e_prey, eff_transfer, eff_repro;
    double chance_p, chance_q, mut_p, mut_q;
    double cap_g;                 // grass regrowth cap: initial_energy_grass (BASE:254) / max_energy_grass (RQ:510)
    // walls variant of the second generation (walls_occlusion/predpreygrass_rllib_env.py, "WO")
    int32_t walls;                // 1: observation channel 0 = walls, wall-blocked moves, per-agent move infos
    int32_t vis_channel;          // include_visibility_channel (WO:104): a fifth observation channel
    int32_t los_move;             // respect_los_for_movement (WO:106)
    int32_t mask_obs;             // mask_observation_with_visibility (WO:111)
    int32_t n_wall_words;         // 32-bit words of the per-env wall bitmap: ceil(G*G / 32)
    int32_t off_wall;             // LDS offset of the wall bitmap
    // per-cell line-of-sight masks, precomputed from the (static) walls by ppg_walls_changed: bit (dx + vis_neg) * vis_w + (dy + vis_neg)
    // of cell (x, y)'s vis_words words = "(x + dx, y + dy) is in the grid and no wall lies strictly between" (WO:492-525, 577-589)
    int32_t vis_neg, vis_w, vis_words;
    int32_t ch0_map;              // cooperative kernels: 1 = four cell maps per env, channel 0's among them (its halo points at the constant 1.0);
                                  // 0 = THREE maps, channel 0 computed from the window position (kernels ppgcm_*, Env's CH0MAP = false) --
                                  // for grids whose LDS footprint decides how many workgroups a CU holds (64x64: 24.1 -> 18.1 KB per env)
    uint32_t rp_magic, rq_magic;  // ceil(2^32 / Rp), ceil(2^32 / Rq): cell / R == mulhi(cell, magic) for cell < R*R
    uint32_t np_magic, nq_magic;  // ceil(2^32 / Rp^2), ceil(2^32 / Rq^2): element / R^2 for element < 8 R^2
    uint32_t *vis_masks;          // library-owned [B, G*G, vis_words]; NULL = not computed: observations walk the lines themselves
    // drive-conditioned variant of the base family (drive_conditioned_environment/predpreygrass_rllib_env.py, "DRV")
    int32_t n_drive[2];           // extra constant-filled observation channels per species (DRV:70-75), <= 4
    int32_t drive_kind[2][4];     // 0 hunger_pressure, 1 reproductive_readiness, 2 prey_opportunity, 3 predator_danger_pressure,
                                  // 4 grass_opportunity (DRV:587-608)
    int32_t off_win;              // LDS offset of the window staging area (one float64 per window cell)
    int32_t pad3_;
    double hunger_safe[2], norm_prey_opp, norm_pred_danger, norm_grass_opp;   // DRV:76-86
    // LDS layout (bytes from the start of dynamic LDS)
    int32_t map_n;    // u16 entries per channel map (>= G*G, multiple of 8)
    int32_t off_map;  // 4 maps: [0] always zero (channel 0), [1] predators, [2] prey, [3] grass.  Cooperative kernels with ch0_map 0:
                      // THREE maps (predators, prey, grass) -- channel 0 is a function of the window position (Env::chmap)
    int32_t off_val;  // float64 value table: [0]=0, 1+row predators, 1+cap_pred+row prey, then grass
    int32_t off_scr;  // 8-byte scratch per row (permutation / reset random words)
    int32_t off_lut;  // observation element descriptors (see Env::obs_row), predators then prey
    int32_t nch_p, nch_q;  // 128-element chunks per (4,R,R) block: ceil(4*R*R/128)
    int32_t lds_bytes;
    // buffers (caller-owned)
    uint16_t *row_xy;
    double *row_e;
    int32_t *row_id;
    uint32_t *row_key;
    double *row_cum;
    uint8_t *row_flags;
    double *row_reward;
    int32_t *env_state;
    uint64_t *env_seed;
    uint16_t *grass_xy;
    double *grass_e;
    void *obs_pred;
    void *obs_prey;
    int32_t *row_parent;
    int32_t *row_lastrep;     // second generation: agent_last_reproduction
    uint32_t *wall_bits;      // walls variant: [B, n_wall_words], bit (x*G+y) set = wall
    uint8_t *row_info;        // walls variant: [B,S] 0 = no move info, else 1 + move_blocked_reason code (WO:466-488)
    const uint32_t *obs_lut;  // library-owned, (nch_p + nch_q) * 128 words
    // per-launch
    const int8_t *actions;
    const uint8_t *act_rank;  // optional [B,S]: position of each row in its type's action sequence
    const uint64_t *seeds;
    const double *uniforms;   // second generation, ppg_step_uniforms: [B, uniforms_per_env]
    double *grid_out;
    unsigned long long *prof;  // diagnostic build only
    const int32_t *env_order;  // optional permutation of 0..batch-1 (ppg_rebalance): workgroup i steps env env_order[i]
    uint32_t flags;
    uint32_t reset_episode;
    int32_t n_steps;   // transitions per launch (ppg_step: 1; ppg_rollout: n)
    int32_t helper_min_rows;   // multi-wave kernels: helper wavefronts only stay for envs with at least this many agent rows at the
                               // start of the call (0 = always); for lighter envs they exit at once and wave 0 writes all rows
    int32_t mode;
    int32_t batch;
    // cooperative step kernels (ppgc_*, Env's COOP): coop_e envs share one workgroup of NW >= coop_e wavefronts.  The cell maps are
    // padded by `pad` cells on every side (Gp = G + 2 pad), so an observation window never leaves its map.
    int32_t pad, Gp;
    int32_t coop_e;            // envs per workgroup
    int32_t lds_env_bytes;     // LDS region of one env (map / val / scr offsets above are relative to it)
    int32_t off_lut2;          // from the start of dynamic LDS: the workgroup's copy of obs_lut2
    int32_t off_ctl;           // from the start of dynamic LDS: control words (Env::CTL_*)
    int32_t blk_p, blk_q;      // elements per observation block: channels x Rp^2, channels x Rq^2
    uint32_t bp_magic, bq_magic;  // ceil(2^32 / blk): element / blk == mulhi(element, magic)
    const uint32_t *coop_tab;  // library-owned: blk_p + blk_q observation descriptors.  Element (channel, i, j) of a species' (4,R,R)
                               // block: bits 0-15 the signed map offset relative to the observer's padded cell, map index * map_n
                               // + (i - off) * Gp + (j - off); bits 16-31 the value-table section of the channel.  ch0_map 1: then
                               // map_n / 4 words, the padded channel-0 map of an empty grid (halo cells = Env::ONE_IDX).  ch0_map 0:
                               // channel 0 ("outside the grid", BASE:520-523) has no map: bits 16-31 = 0xFFFF, bits 4-7 (i - off) + 8,
                               // bits 0-3 (j - off) + 8 -- the element is 1.0 iff (x + i - off, y + j - off) lies outside the grid
};

// ---------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------

PPG_DEVICE uint64_t bit64(int k) { return 1ull << k; }
PPG_DEVICE uint64_t lowmask(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }

PPG_DEVICE double readlane_f64(double v, int k) {
    long long b = __double_as_longlong(v);
    uint32_t lo = wv::readlane((uint32_t)b, k), hi = wv::readlane((uint32_t)((uint64_t)b >> 32), k);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
PPG_DEVICE double writelane_f64(double v, int k, double s) {
    long long b = __double_as_longlong(v), sb = __double_as_longlong(s);
    uint32_t lo = wv::writelane((uint32_t)b, k, (uint32_t)sb);
    uint32_t hi = wv::writelane((uint32_t)((uint64_t)b >> 32), k, (uint32_t)((uint64_t)sb >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
PPG_DEVICE double first_f64(double v) {
    long long b = __double_as_longlong(v);
    uint32_t lo = wv::first((uint32_t)b), hi = wv::first((uint32_t)((uint64_t)b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// Order of Python's list.sort() on "prey_<id>" strings (BASE:468): decimal digits compared
// left to right, a shorter string that is a prefix sorts first.  digit d -> d+1, absent -> 0,
// base 11, six positions.
PPG_DEVICE uint32_t lexkey(uint32_t id) {
    uint32_t L = 1 + (id >= 10u) + (id >= 100u) + (id >= 1000u) + (id >= 10000u) + (id >= 100000u);
    uint32_t pw = L == 1 ? 161051u : L == 2 ? 14641u : L == 3 ? 1331u : L == 4 ? 121u : L == 5 ? 11u : 1u;
    uint32_t key = 0;
    for (uint32_t q = 0; q < 6; ++q) {
        if (q < L) {
            uint32_t nx = id / 10u;
            key += (id - nx * 10u + 1u) * pw;
            pw *= 11u;
            id = nx;
        }
    }
    return key;
}

// math.sqrt(dx*dx + dy*dy) for displacements of at most 3 cells per axis (RQ:310): correctly rounded constants
PPG_DEVICE double move_distance(int d2) {
    return d2 == 0 ? 0.0 : d2 == 1 ? 1.0 : d2 == 2 ? 1.4142135623730951 : d2 == 4 ? 2.0 : d2 == 5 ? 2.23606797749979
         : d2 == 8 ? 2.8284271247461903 : d2 == 9 ? 3.0 : d2 == 10 ? 3.1622776601683795 : d2 == 13 ? 3.605551275463989
         : 4.242640687119285;
}

// float -> bfloat16 bits, round to nearest even (finite values: energies and 0 / 1).  Exactly what v_cvt_pk_bf16_f32 and the
// policy kernels' staging (`(__bf16)(float)x`) produce, so bf16 observation rows give bit-identical logits.
PPG_DEVICE uint32_t bf16_bits(float f) {
    uint32_t u;
#ifdef PPG_WAVE_EMU
    memcpy(&u, &f, 4);
#else
    u = __float_as_uint(f);
#endif
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// elements o, o + 1 of an observation buffer (o even) as one vector store: float64 (the reference's dtype), float32, or bfloat16
// (obs_dtype 2: the compact rows the policy kernels stage from -- SURVEY 8(f) N4)
PPG_DEVICE void store_obs_pair(void *base, int obs_dtype, size_t o, double v0, double v1) {
    if (obs_dtype == 1) {
        float2 f; f.x = (float)v0; f.y = (float)v1;
        *(float2 *)((float *)base + o) = f;
    } else if (obs_dtype == 2) {
        *(uint32_t *)((uint16_t *)base + o) = bf16_bits((float)v0) | (bf16_bits((float)v1) << 16);
    } else {
        double2 g; g.x = v0; g.y = v1;
        *(double2 *)((double *)base + o) = g;
    }
}

PPG_DEVICE void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = wv::mulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = wv::mulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------------------------
// one environment, one wavefront
// ---------------------------------------------------------------------------------

// ORDERED: compile the explicit-action-order path (ppg_step_ordered).  It indexes row registers with
// run-time values, which costs registers and scratch, so it lives in its own kernel variant.
// FASTOBS: observation descriptors of this lane live in registers (needs <= 2 predator and <= 3 prey
// chunks, i.e. Rp <= 7 and Rq <= 9); otherwise they are read from an LDS copy.
// FUSED: the multi-step rollout loop (ppg_rollout) is compiled in; ppg_step's kernel has a single step body.
// KP: the type the parameters are read through: `const KParams` (by-value kernel argument) or the same struct in
// the constant address space, read in place from the kernarg segment (fused rollout).
// KICK: the kickback-reward variant (grandparent bonus) is compiled in.  Measured: merely carrying that code costs the
// base path 11 % (register / SGPR pressure), so it has its own kernel variants.
// GEN2: the second-generation step (two agent types per species, move cost, energy caps, stochastic reproduction:
// red_queen/predpreygrass_rllib_env.py, "RQ").  Its own kernel variants; the base kernels compile none of it.
// WALLS (with GEN2): the walls_occlusion variant -- static walls in a per-env bitmap, observation channel 0 = walls,
// optional line-of-sight mask / fifth channel, wall- and LOS-blocked moves, per-agent move infos.
// DRIVE (base family): the drive-conditioned variant -- extra observation channels filled with per-agent scalars.
// NW: wavefronts per environment.  1 everywhere except the multi-wave step kernels (ppgw_*): there wave 0 runs the
// whole transition and all NW waves of the workgroup write the final observations (the phase that dominates a wave's
// run time) -- for launches that cannot fill the GPU with one wave per env (small batches; large grids whose LDS
// footprint allows only a few envs per CU).
// MAP8 (= NQ <= 2, chosen by env_main): the cell maps hold 8-bit indices LOCAL to their channel (predator row + 1, prey row + 1,
// grass patch + 1; 0 = empty) instead of 16-bit indices into the whole value table -- half the LDS, which is what decides how many
// envs of a 64x64 grid fit on a CU (BASELINE config 4).  Needs <= 128 prey rows and <= 255 grass patches (ppg_create checks).
// COOP: the cooperative step kernels (ppgc_*).  A workgroup of NW wavefronts steps P.coop_e <= NW envs: wave k < coop_e runs the
// transition of env k in its own LDS region and publishes the rows to observe; after ONE workgroup barrier all NW waves write the
// final observations of all the workgroup's envs, cut into whole 1 KB pieces that ignore row boundaries (an env's live rows are a
// run of elements; piece p of the workgroup goes to wave p mod NW).  Every store instruction writes a full 1 KB, the rows of
// the workgroup's envs are balanced over its waves at piece grain, every wave runs a transition (no idle helper waves on a full
// GPU) and the waves of a workgroup write neighbouring addresses at the same time.  Cell maps are padded (KParams::pad), so no
// window element needs a bounds check: channel 0's halo points at a constant 1.0 (BASE:522-523), the other halos at 0.0.  Where
// LDS decides the occupancy (round 6: 64x64 grids 24.1 -> 18.1 KB per env = four instead of three workgroups per CU) channel 0 has
// NO map and is computed from the window position (KParams::ch0_map 0; the extra arithmetic per element costs the float32 rows of
// the second generation 11 %, so the small grids keep their fourth map).
template <int NQ, bool ORDERED, bool FASTOBS, bool FUSED, bool KICK, bool GEN2, bool WALLS, bool DRIVE, int NW, class KP, class KC, bool COOP = false, bool CH0MAP = true>
struct Env {
    static constexpr int T = 1 + NQ;  // row registers: 0 = predators, 1.. = prey
    static constexpr bool MAP8 = NQ <= 2;
    // Helper wavefronts that leave light envs to wave 0 (KParams::helper_min_rows) need BOTH observation paths in one kernel.  The
    // walls and drive variants' observation code is large: with two copies of it their kernels ran 10-15 % slower, so their helpers
    // always stay.
    static constexpr bool ADAPTIVE_HELPERS = NW > 1 && !WALLS && !DRIVE && !COOP;
    // cumulative_rewards rides through the step in registers (fetched with the rows, permuted with them).  The other kernels read
    // it back from HBM at the row's start-of-step slot when the rewards are assembled, to save two registers per row register --
    // but that is a second dependent memory round trip per call, and under a saturated memory system a round trip is 5-10 k cycles.
    // Round 5: the multi-wave kernels of the base family too (BASELINE configs 2 and 4: ppgw16_step / ppgwp_step) -- their phase profiles
    // (profiles/r05) show the "store" phase, i.e. this read-back, at 16-18 % of a wavefront's transition.
#ifndef PPG_CARRY_CUM_MULTIWAVE
#define PPG_CARRY_CUM_MULTIWAVE 1
#endif
    // (round 6, measured and not kept: the walls variant's four-wave kernel too -- 94.3 against 92.3 us per 4096-env step at its
    // 64-register cap, six more row registers spill; 102 us at 80 registers: profiles/r06/g_*)
    static constexpr bool CARRY_CUM = COOP || (PPG_CARRY_CUM_MULTIWAVE && NW > 1 && !GEN2 && !WALLS && !DRIVE && !KICK && NQ <= 2);
    template <bool B8, class Dummy = void> struct MapElem { typedef uint16_t type; };
    template <class Dummy> struct MapElem<true, Dummy> { typedef uint8_t type; };
    typedef typename MapElem<MAP8>::type map_t;

    KP &P;   // hot parameters: by-value kernel argument, resident in SGPRs
    KC &C;   // cold parameters (rewards, thresholds, table pointers ...): the same struct read in place from the
             // kernarg segment at their single use sites, so they do not occupy SGPRs for the whole kernel
    const int b;
    const int ln;
    int wave_idx = 0;  // index of this wavefront in its workgroup (multi-wave kernels; 0 otherwise)
    bool helpers = true;  // multi-wave kernels: the helper wavefronts of this env take part (see KParams::helper_min_rows)

    map_t *map;      // LDS
    double *val;     // LDS
    uint64_t *scr;   // LDS
    uint32_t *lut;   // LDS
    uint32_t *wallw; // LDS: wall bitmap (WALLS)
    // COOP: the workgroup's shared LDS area (set by env_main): observation descriptors and control words
    uint32_t *lut2 = nullptr;
    uint32_t *ctl = nullptr;
    // control words: per env slot k of the workgroup CTL_SLOT + 4k: live predator rows, live prey rows, env index (-1 = no env);
    // per wavefront w CTL_MID + w: the (row, cell) entry of a mid-step observation
    enum { CTL_SLOT = 0, CTL_MID = 64, CTL_WORDS = 80 };

    // per-lane row fields
    uint32_t xy[T];
    int32_t id[T];
    uint32_t key[T];
    double e[T];
    int32_t act[T];
    uint32_t ev[T];
    uint32_t keep[T];  // row flags that survive a truncation call (ATE)
    uint32_t rank[T];  // explicit action order (only when C.act_rank is given)
    uint32_t gxyr[2];  // grass_xy of patches ln and ln+64 (static within an episode)
    uint32_t lutr[10]; // FASTOBS: this lane's descriptors, predator chunks 0-1 then prey chunks 0-2, two words each
    int32_t lr[T];     // GEN2: agent_last_reproduction of the row
    double cum[T];     // CARRY_CUM: cumulative_rewards of the row

    // wave-uniform state
    uint64_t rows[T], alive[T], owns[T];
    int n_rows[2], next_id[2], n_alive[2];
    int step, fb_count, calls;
    int obs_count[2];
    uint32_t envflags, status, episode;
    uint64_t seed;
    bool cooc[2];  // some cell may hold two live agents of this type
    uint64_t t2m[T];   // GEN2: rows holding a type-2 agent
    int next_id2[2];   // GEN2: _next_idx of the type-2 pools (next_id: type 1)
    int draws;         // GEN2: uniforms consumed by this call

    PPG_MEMBER Env(KP &p, KC &c, int b_, unsigned char *lds, int lane)
        : P(p), C(c), b(b_), ln(lane),
          map((map_t *)(lds + p.off_map)), val((double *)(lds + p.off_val)),
          scr((uint64_t *)(lds + p.off_scr)), lut((uint32_t *)(lds + p.off_lut)), wallw((uint32_t *)(lds + c.off_wall)) {}

    // ---- index helpers -------------------------------------------------------------
    static PPG_MEMBER int type_of(int r) { return r ? 1 : 0; }
    static PPG_MEMBER int row_of(int r, int k) { return r ? (r - 1) * 64 + k : k; }  // row within its type
    PPG_MEMBER int slot_of(int r, int k) const { return r ? P.cap_pred + (r - 1) * 64 + k : k; }  // row in [0,S)
    // Index of an entity's energy in the LDS value table.  16-bit maps: [0] = 0.0, then all row slots, then the grass patches.
    // MAP8: one section per channel, each led by a zero entry so that "map entry + section base" needs no test for an empty cell:
    // [0] = 0.0 | predators 1..64 || [129] = 0.0 | prey 130..257 || [258] = 0.0 | grass 259..513  (section base = 129 * (channel - 1)).
    // The cooperative kernels pack the sections (LDS decides how many envs a CU holds): [0] = 0.0 | predators 1..64 | [65] = 1.0 (the
    // "outside the grid" value of channel 0's halo, ch0_map 1) || [66] = 0.0 | prey 67..194 || [195] = 0.0 | grass 196...
    static constexpr int SEC_Q = COOP ? 66 : 129, SEC_G = SEC_Q + 129;
    PPG_MEMBER int validx(int r, int k) const { return MAP8 ? (r ? SEC_Q + 1 + row_of(r, k) : 1 + k) : 1 + slot_of(r, k); }
    PPG_MEMBER int validx_row(int type, int row) const { return MAP8 ? (type ? SEC_Q + 1 + row : 1 + row) : 1 + (type ? P.cap_pred + row : row); }
    PPG_MEMBER int grass_validx(int p) const { return MAP8 ? SEC_G + 1 + p : 1 + P.S + p; }
    PPG_MEMBER int cell_of(uint32_t s_xy) const {
        if (COOP) return ((int)(s_xy >> 8) + P.pad) * P.Gp + (int)(s_xy & 255u) + P.pad;   // padded maps
        return (int)(s_xy >> 8) * P.G + (int)(s_xy & 255u);
    }
    // map index of cell c = x * G + y (0 <= c < G*G)
    PPG_MEMBER int cell_index(int c) const {
        if (!COOP) return c;
        const int x = (int)wv::mulhi((uint32_t)c, C.g_magic);
        return (x + P.pad) * P.Gp + (c - x * P.G) + P.pad;
    }
    // the cell map of channel ch (1 predators, 2 prey, 3 grass; 0 = the all-zero map of channel 0, which the cooperative kernels do not have)
    // (CH0MAP = false, cooperative kernels only: three maps -- KParams::ch0_map 0; a kernel of its own, ppgcm_*: both forms in one
    // kernel made every inlined copy of coop_pieces carry two loops, +25 % code, and cost the small grids 1-2 %)
    static_assert(CH0MAP || COOP, "only the cooperative kernels come without a channel-0 map");
    static constexpr bool THREE = !CH0MAP;
    PPG_MEMBER map_t *chmap(int ch) const { return map + (THREE ? ch - 1 : ch) * P.map_n; }
    static constexpr int N_MAPS = THREE ? 3 : 4;
    // what a map entry of channel ch means as an index into the value table, and back (MAP8: channel-local 8-bit indices)
    PPG_MEMBER int map_base(int ch) const { return MAP8 ? (ch == 2 ? SEC_Q : ch == 3 ? SEC_G : 0) : 0; }
    PPG_MEMBER map_t to_map(int ch, int vidx) const { return (map_t)(vidx - map_base(ch)); }
    PPG_MEMBER int from_map(int ch, uint32_t m) const { return (int)m + map_base(ch); }   // (an empty cell lands on the section's zero entry)

    // ---- walls (WO) ----------------------------------------------------------------------
    PPG_MEMBER bool wall_at(int x, int y) const {
        const int c = x * P.G + y;
        return (wallw[c >> 5] >> (c & 31)) & 1u;
    }
    // _line_of_sight_clear (WO:492-525) == the bresenham walk of _get_observation (WO:550-589): no wall strictly between
    // the two cells.  The reference's float error term dx/2.0 is carried doubled, as an integer.
    PPG_MEMBER bool los_clear(int x0, int y0, int x1, int y1) const {
        const int dx = x1 > x0 ? x1 - x0 : x0 - x1, dy = y1 > y0 ? y1 - y0 : y0 - y1;
        const int sx = x1 > x0 ? 1 : -1, sy = y1 > y0 ? 1 : -1;
        int x = x0, y = y0;
        bool clear = true;
        if (dx >= dy) {
            int err2 = dx;
            while (x != x1) {
                if (!(x == x0 && y == y0) && !(x == x1 && y == y1) && wall_at(x, y)) clear = false;
                err2 -= 2 * dy;
                if (err2 < 0) { y += sy; err2 += 2 * dx; }
                x += sx;
            }
        } else {
            int err2 = dy;
            while (y != y1) {
                if (!(x == x0 && y == y0) && !(x == x1 && y == y1) && wall_at(x, y)) clear = false;
                err2 -= 2 * dx;
                if (err2 < 0) { x += sx; err2 += 2 * dy; }
                y += sy;
            }
        }
        return clear;
    }
    // What the walls do to a move from (x,y) to the clipped target (tx,ty): MV_WALL, MV_CORNER_CUT, MV_LOS or MV_NONE
    // (WO:469-488; the occupancy test sits between the wall test and the line-of-sight tests and is applied by the caller).
    PPG_MEMBER uint32_t wall_verdict(int x, int y, int tx, int ty) const {
        if (wall_at(tx, ty)) return MV_WALL;
        if (!C.los_move || (tx == x && ty == y)) return MV_NONE;
        const int ddx = tx - x, ddy = ty - y;
        if ((ddx == 1 || ddx == -1) && (ddy == 1 || ddy == -1))
            return (wall_at(x + ddx, y) || wall_at(x, y + ddy)) ? (uint32_t)MV_CORNER_CUT : (uint32_t)MV_NONE;
        return los_clear(x, y, tx, ty) ? (uint32_t)MV_NONE : (uint32_t)MV_LOS;
    }
    // The same for moves of at most two cells per axis (action ranges 3 and 5, all the reference configures), without a loop:
    // a one-cell diagonal looks at the two cells beside the corner (WO:474-480); a two-cell move has exactly ONE cell strictly
    // between its ends -- bresenham's minor axis steps with the first major step only when both displacements are 2 -- and
    // the three bits are read side by side.
    PPG_MEMBER uint32_t wall_verdict_near(int x, int y, int tx, int ty) const {
        const int ddx = tx - x, ddy = ty - y;
        const int adx = ddx < 0 ? -ddx : ddx, ady = ddy < 0 ? -ddy : ddy;
        const int sx = (ddx > 0) - (ddx < 0), sy = (ddy > 0) - (ddy < 0);
        const bool diag1 = adx == 1 && ady == 1, far = adx == 2 || ady == 2;
        int ax = tx, ay = ty, bx = tx, by = ty;
        if (diag1) { ax = tx; ay = y; bx = x; by = ty; }
        if (far) {
            ax = adx >= ady ? x + sx : x;
            ay = adx >= ady ? (ady == 2 ? y + sy : y) : y + sy;
            bx = ax; by = ay;
        }
        const bool wt = wall_at(tx, ty), wa = wall_at(ax, ay), wb = wall_at(bx, by);
        if (wt) return MV_WALL;
        if (!C.los_move || (adx | ady) == 0) return MV_NONE;
        if (diag1) return (wa || wb) ? (uint32_t)MV_CORNER_CUT : (uint32_t)MV_NONE;
        if (far) return wa ? (uint32_t)MV_LOS : (uint32_t)MV_NONE;
        return MV_NONE;
    }
    PPG_MEMBER void set_move_info(int r, uint32_t code) { keep[r] = (keep[r] & ~7u) | code; }

    // alive rows of `type` standing on s_xy, per register
    PPG_MEMBER void match(int type, uint32_t s_xy, uint64_t (&m)[T]) const {
#pragma unroll
        for (int r = 0; r < T; ++r) m[r] = (type_of(r) == type) ? (wv::ballot(xy[r] == s_xy) & alive[r]) : 0ull;
    }
    PPG_MEMBER bool any_agent_at(uint32_t s_xy) const {
        uint64_t acc = 0;
#pragma unroll
        for (int r = 0; r < T; ++r) acc |= wv::ballot(xy[r] == s_xy) & alive[r];
        return acc != 0;
    }

    // "grid[ch][cell] = 0" (BASE:268,272,293,297,335): whoever owned the cell no longer does.
    PPG_MEMBER void grid_zero(int type, uint32_t s_xy, bool touch_lds) {
#pragma unroll
        for (int r = 0; r < T; ++r)
            if (type_of(r) == type) owns[r] &= ~wv::ballot(xy[r] == s_xy);
        if (touch_lds && ln == 0) chmap(1 + type)[cell_of(s_xy)] = 0;
    }
    // "grid[ch][cell] = energy of row (r,k)" (BASE:247,250,269,273,325,368,405,406)
    PPG_MEMBER void grid_set(int r, int k, uint32_t s_xy, double s_e, bool touch_lds) {
        const int type = type_of(r);
#pragma unroll
        for (int q = 0; q < T; ++q)
            if (type_of(q) == type) owns[q] &= ~wv::ballot(xy[q] == s_xy);
#pragma unroll
        for (int q = 0; q < T; ++q) owns[q] |= (q == r) ? bit64(k) : 0ull;
        if (touch_lds && ln == 0) {
            chmap(1 + type)[cell_of(s_xy)] = to_map(1 + type, validx(r, k));
            val[validx(r, k)] = s_e;
        }
    }

    // ---- load ----------------------------------------------------------------------
    // Every global load that does not depend on another load is issued first, back to back, so the
    // wave pays ONE memory round trip: env words, seed, the first two row registers (speculatively:
    // rows beyond n_rows are valid memory holding stale data and are masked out), the first 128 grass
    // patches and the observation descriptor table.
    struct Pre {
        uint32_t w_env;
        uint64_t sd;
        uint32_t xy[T], key[T], fl[T];
        int32_t id[T], a[T];
        double e[T];
        double cum[T];
        uint32_t gxy[2];
        double ge[2];
        uint2 lutd[5];
    };

    PPG_MEMBER void prefetch(Pre &p, bool want_rows, bool want_actions) {
        const int32_t *es = C.env_state + (size_t)b * PPG_ENV_WORDS;
        p.w_env = ln < PPG_ENV_WORDS ? (uint32_t)es[ln] : 0u;
        p.sd = C.env_seed[b];
#pragma unroll
        for (int r = 0; r < T; ++r) {
            p.xy[r] = 0xFFFFu; p.key[r] = 0; p.fl[r] = 0; p.id[r] = 0; p.a[r] = -1; p.e[r] = 0.0; p.cum[r] = 0.0;
            if (r < 2 && want_rows) {
                const size_t s = (size_t)b * P.S + slot_of(r, ln);
                p.xy[r] = C.row_xy[s];
                p.e[r] = C.row_e[s];
                if (CARRY_CUM) p.cum[r] = C.row_cum[s];
                p.id[r] = C.row_id[s];
                p.key[r] = C.row_key[s];
                p.fl[r] = C.row_flags[s];
                if (want_actions) p.a[r] = C.actions[s];
            }
        }
        const size_t gb = (size_t)b * C.cap_grass;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pp = ln + 64 * q;
            p.gxy[q] = 0; p.ge[q] = 0.0;
            if (want_rows && pp < C.n_grass) { p.gxy[q] = C.grass_xy[gb + pp]; p.ge[q] = C.grass_e[gb + pp]; }
        }
        if (FASTOBS) {  // this lane's observation descriptors (row-independent), kept in registers
            const uint2 *L2 = (const uint2 *)C.obs_lut;
#pragma unroll
            for (int c = 0; c < 2; ++c) { p.lutd[c].x = 0; p.lutd[c].y = 0; if (c < P.nch_p) p.lutd[c] = L2[c * 64 + ln]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) { p.lutd[2 + c].x = 0; p.lutd[2 + c].y = 0; if (c < P.nch_q) p.lutd[2 + c] = L2[(P.nch_p + c) * 64 + ln]; }
        }
    }

    PPG_MEMBER void load_env_words(const Pre &p) {
        const uint32_t w = p.w_env;
        n_rows[0] = (int)wv::readlane(w, PPG_ENV_N_PRED_ROWS);
        n_rows[1] = (int)wv::readlane(w, PPG_ENV_N_PREY_ROWS);
        next_id[0] = (int)wv::readlane(w, PPG_ENV_NEXT_PRED_ID);
        next_id[1] = (int)wv::readlane(w, PPG_ENV_NEXT_PREY_ID);
        step = (int)wv::readlane(w, PPG_ENV_STEP);
        envflags = wv::readlane(w, PPG_ENV_FLAGS);
        status = wv::readlane(w, PPG_ENV_STATUS);
        episode = wv::readlane(w, PPG_ENV_EPISODE);
        fb_count = (int)wv::readlane(w, PPG_ENV_FALLBACK_SPAWNS);
        calls = (int)wv::readlane(w, PPG_ENV_CALLS);
        obs_count[0] = (int)wv::readlane(w, PPG_ENV_OBS_PRED);
        obs_count[1] = (int)wv::readlane(w, PPG_ENV_OBS_PREY);
        next_id2[0] = next_id2[1] = 0;
        draws = 0;
        if (GEN2) {
            next_id2[0] = (int)wv::readlane(w, PPG_ENV_NEXT_PRED_ID_T2);
            next_id2[1] = (int)wv::readlane(w, PPG_ENV_NEXT_PREY_ID_T2);
        }
        seed = ((uint64_t)wv::first((uint32_t)(p.sd >> 32)) << 32) | wv::first((uint32_t)p.sd);
    }

    PPG_MEMBER void load_rows(const Pre &p) {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            const bool valid = i < n_rows[type_of(r)];
            uint32_t fl = 0;
            xy[r] = 0xFFFFu; id[r] = 0; key[r] = 0; e[r] = 0.0; act[r] = -1; ev[r] = 0; cum[r] = 0.0;
            if (valid) {
                if (r < 2) {
                    xy[r] = p.xy[r]; e[r] = p.e[r]; id[r] = p.id[r]; key[r] = p.key[r];
                    fl = p.fl[r]; act[r] = p.a[r];
                    if (CARRY_CUM) cum[r] = p.cum[r];
                } else {  // rows 64.. of the prey table: rarely in use, loaded on demand
                    const size_t s = (size_t)b * P.S + slot_of(r, ln);
                    xy[r] = C.row_xy[s];
                    e[r] = C.row_e[s];
                    if (CARRY_CUM) cum[r] = C.row_cum[s];
                    id[r] = C.row_id[s];
                    key[r] = C.row_key[s];
                    fl = C.row_flags[s];
                    if (C.actions) act[r] = C.actions[s];
                }
            }
            keep[r] = (fl & (PPG_ROW_ATE | (GEN2 ? PPG_ROW_GRID_E0 : 0u))) | ((uint32_t)slot_of(r, ln) << 8);  // bits 8..: where this row's start-of-step energy lives
            lr[r] = 0;
            t2m[r] = GEN2 ? (wv::ballot(valid && ((id[r] >> 16) & 1)) ) : 0ull;
            rows[r] = wv::ballot(valid);
            alive[r] = rows[r] & ~wv::ballot(valid && (fl & PPG_ROW_DIED));
            owns[r] = wv::ballot(valid && (fl & PPG_ROW_OWNS)) & alive[r];
        }
        n_alive[0] = n_alive[1] = 0;
#pragma unroll
        for (int r = 0; r < T; ++r) n_alive[type_of(r)] += wv::popc(alive[r]);
    }

    // maps -> all zero, observation descriptors -> LDS
    PPG_MEMBER void init_lds(const Pre &p) {
        if (!COOP) init_maps();   // (COOP: coop_tab_store)
        if (COOP) {
            if (CH0MAP && ln == 0) val[ONE_IDX] = 1.0;
        } else if (FASTOBS) {
#pragma unroll
            for (int c = 0; c < 5; ++c) { lutr[2 * c] = p.lutd[c].x; lutr[2 * c + 1] = p.lutd[c].y; }
        } else {
            for (int i = ln; i < (P.nch_p + P.nch_q) * 128; i += 64) lut[i] = C.obs_lut[i];
        }
        if (ln == 0) val[0] = 0.0;
        if (MAP8 && ln < 2) val[ln ? SEC_G : SEC_Q] = 0.0;   // the zero entries leading the prey and grass sections
        gxyr[0] = p.gxy[0];
        gxyr[1] = p.gxy[1];
        if (WALLS)
            for (int i = ln; i < C.n_wall_words; i += 64) wallw[i] = C.wall_bits[(size_t)b * C.n_wall_words + i];
    }

    // all cell maps empty.  COOP with a channel-0 map: plus that map's halo -> the constant 1.0 of the value table ("outside the grid",
    // BASE:520-523); the halos of channels 1-3 stay 0 -> the zero entry of their section.
    static constexpr int ONE_IDX = 65;   // a free entry of the predator section (rows use 1..64)
    PPG_MEMBER void zero_maps(int first_word) {   // words first_word.. of the map area
        uint32_t *m32 = (uint32_t *)map;
        const int n32 = N_MAPS * P.map_n * (int)sizeof(map_t) / 4;
        // (16-byte stores where the range allows: 64x64 grids zero 14.7 KB per step)
        const int lo16 = (first_word + 3) >> 2, n128 = n32 >> 2;
        uint4 *m128 = (uint4 *)map;
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (int i = first_word + ln; i < 4 * lo16 && i < n32; i += 64) m32[i] = 0u;
        for (int i = lo16 + ln; i < n128; i += 64) m128[i] = zero;
        for (int i = (4 * n128 > first_word ? 4 * n128 : first_word) + ln; i < n32; i += 64) m32[i] = 0u;
    }
    PPG_MEMBER void init_maps() {
        if (COOP && CH0MAP) {   // (channel 0 from the template behind the descriptors in C.coop_tab; a step has it prefetched: TabPre)
            const uint32_t *tmpl = C.coop_tab + C.blk_p + C.blk_q;
            const int n0 = P.map_n / 4;
            uint32_t *m32 = (uint32_t *)map;
            for (int i = ln; i < n0; i += 64) m32[i] = tmpl[i];
            zero_maps(n0);
        } else {
            zero_maps(0);
        }
    }
    // COOP: the workgroup's descriptor table and (ch0_map) this env's channel-0 map come from C.coop_tab.  Their loads are issued in
    // front of everything else and held in registers (up to LUT_REGS / TMPL_REGS words per lane, enough for 7x7 / 9x9 windows on a
    // 25x25 grid; larger geometries finish with plain copy loops), so the tables cost no memory round trip of their own.
    static constexpr int LUT_REGS = 9, TMPL_REGS = 5;
    struct TabPre { uint32_t l[LUT_REGS], m[TMPL_REGS]; };
    PPG_MEMBER void coop_tab_issue(TabPre &t) const {
        const int nl = C.blk_p + C.blk_q, nm = CH0MAP ? P.map_n / 4 : 0;
#pragma unroll
        for (int u = 0; u < LUT_REGS; ++u) { t.l[u] = 0; if (u * 64 + ln < nl) t.l[u] = C.coop_tab[u * 64 + ln]; }
#pragma unroll
        for (int u = 0; u < TMPL_REGS; ++u) { t.m[u] = 0; if (u * 64 + ln < nm) t.m[u] = C.coop_tab[nl + u * 64 + ln]; }
    }
    PPG_MEMBER void coop_tab_store(const TabPre &t) {
        const int nl = C.blk_p + C.blk_q, nm = CH0MAP ? P.map_n / 4 : 0;
        uint32_t *m32 = (uint32_t *)map;
        zero_maps(nm);   // channels 1-3: empty
#pragma unroll
        for (int u = 0; u < LUT_REGS; ++u) if (u * 64 + ln < nl) lut2[u * 64 + ln] = t.l[u];
#pragma unroll
        for (int u = 0; u < TMPL_REGS; ++u) if (u * 64 + ln < nm) m32[u * 64 + ln] = t.m[u];
        for (int i = LUT_REGS * 64 + ln; i < nl; i += 64) lut2[i] = C.coop_tab[i];
        for (int i = TMPL_REGS * 64 + ln; i < nm; i += 64) m32[i] = C.coop_tab[nl + i];
    }

    // grass table -> LDS (value table + channel-3 map).  regrow: BASE:252-256.
    PPG_MEMBER void load_grass(bool regrow, const Pre &p) {
        const size_t gb = (size_t)b * C.cap_grass;
        // seasonal variant: square wave on current_step (base_environment_seasonal/...:224-234,268)
        double gain = C.gain_g;
        if (C.season_len > 0) gain = C.gain_g * (((step / C.season_len) & 1) ? C.season_lo : C.season_hi);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pp = ln + 64 * q;
            if (pp < C.n_grass) {
                double g = p.ge[q];
                if (regrow) {
                    double v = g + gain;
                    g = (C.cap_g < v) ? C.cap_g : v;  // Python min(v, cap)
                }
                val[grass_validx(pp)] = g;
                chmap(3)[cell_of(gxyr[q])] = to_map(3, grass_validx(pp));
            }
        }
        for (int pp = 128 + ln; pp < C.n_grass; pp += 64) {
            double g = C.grass_e[gb + pp];
            if (regrow) {
                double v = g + gain;
                g = (C.cap_g < v) ? C.cap_g : v;
            }
            val[grass_validx(pp)] = g;
            chmap(3)[cell_of(C.grass_xy[gb + pp])] = to_map(3, grass_validx(pp));
        }
    }

    // ---- actions -------------------------------------------------------------------
    // size of this lane's action space: 9 (BASE:108), or range^2 of the agent's type (RQ:974-985)
    PPG_MEMBER int n_actions(int r) const {
        if (!GEN2) return 9;
        const int a = ((id[r] >> 16) & 1) ? C.ar[1] : C.ar[0];
        return a * a;
    }
    // action -> (dx, dy): BASE:96-106 (a//3-1, a%3-1); RQ:141-146 with the range of the agent's type
    PPG_MEMBER void move_vector(int a, bool type2, int &dx, int &dy) const {
        if (!GEN2) {
            const int ax = (a * 11) >> 5;  // a / 3 for 0..8
            dx = ax - 1; dy = a - 3 * ax - 1;
        } else {
            const int side = type2 ? C.ar[1] : C.ar[0];
            const uint32_t inv = type2 ? C.ar_inv[1] : C.ar_inv[0];
            const int ax = (int)(((uint32_t)a * inv) >> 16), delta = (side - 1) >> 1;
            dx = ax - delta; dy = a - ax * side - delta;
        }
    }
    // the value grid[type, pos] shows for this lane's row: its energy, or the birth value (RQ:760)
    PPG_MEMBER double shown(int r) const {
        if (GEN2 && (keep[r] & PPG_ROW_GRID_E0)) return r ? C.e0_q : C.e0_p;
        return e[r];
    }
    PPG_MEMBER bool shown_positive(int r) const {
        if (GEN2) return (float)shown(r) > 0.0f;  // the reference's grid is float32 (RQ:138,339)
        return e[r] > 0.0;
    }

    PPG_MEMBER void load_actions(uint64_t (&acted)[T]) {
        bool bad = false;
        if (C.flags & PPG_STEP_RANDOM_ACTIONS) {
            uint32_t w[4];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if ((r & 3) == 0)
                    philox4x32_10((uint32_t)step, (uint32_t)ln + 64u * (uint32_t)(r >> 2), 0u, episode,
                                  (uint32_t)seed, (uint32_t)(seed >> 32) ^ TAG_ACT, w);
                act[r] = (int32_t)wv::mulhi(w[r & 3], (uint32_t)n_actions(r));
            }
        } else {
#pragma unroll
            for (int r = 0; r < T; ++r) {
                int a = ((alive[r] >> ln) & 1ull) ? act[r] : -1;  // fetched with the rows
                if (a < -1 || a >= n_actions(r)) { bad = true; a = -1; }
                act[r] = a;
            }
        }
        if (wv::ballot(bad)) status |= PPG_STATUS_BAD_ACTION;
#pragma unroll
        for (int r = 0; r < T; ++r) acted[r] = alive[r] & wv::ballot(act[r] >= 0);
#pragma unroll
        for (int r = 0; r < T; ++r) {
            rank[r] = 0;
            if (ORDERED && C.act_rank && ((acted[r] >> ln) & 1ull)) rank[r] = C.act_rank[(size_t)b * P.S + slot_of(r, ln)];
        }
    }

    // Explicit action order (a dict whose order differs from the previous observation dict): rows of
    // `type` that act, as (register, lane) pairs in action order, through the LDS scratch.
    PPG_MEMBER int publish_order(int type, const uint64_t (&acted)[T]) {
        uint16_t *ord = (uint16_t *)scr + (type ? 64 : 0);
        int n = 0;
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r) {
            if (type_of(r) != type) continue;
            if ((acted[r] >> ln) & 1ull) ord[rank[r]] = (uint16_t)row_of(r, ln);
            n += wv::popc(acted[r]);
        }
        wv::sync();
        return n;
    }
    PPG_MEMBER void ordered_row(int type, int i, int &r, int &k) const {
        const uint16_t *ord = (const uint16_t *)scr + (type ? 64 : 0);
        const int row = (int)wv::first((uint32_t)ord[i]);
        r = type ? 1 + (row >> 6) : 0;
        k = row & 63;
    }
    // xy of row (r,k) where r may be a run-time (wave-uniform) register index.  The lane is read from every
    // register first and the scalars are selected afterwards: selecting between the member arrays themselves makes
    // the compiler select between their ADDRESSES, which pins the whole Env object (and the parameters) in scratch.
    // With a compile-time r the unused reads fold away.
    PPG_MEMBER uint32_t xy_at(int r, int k) const {
        uint32_t v = wv::readlane(xy[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane(xy[q], k); v = (q == r) ? vq : v; }
        return v;
    }
    PPG_MEMBER int act_at(int r, int k) const {
        uint32_t v = wv::readlane((uint32_t)act[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane((uint32_t)act[q], k); v = (q == r) ? vq : v; }
        return (int)v;
    }
    PPG_MEMBER uint32_t id_at(int r, int k) const {
        uint32_t v = wv::readlane((uint32_t)id[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane((uint32_t)id[q], k); v = (q == r) ? vq : v; }
        return v;
    }
    PPG_MEMBER double e_at(int r, int k) const {
        double v = readlane_f64(e[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const double vq = readlane_f64(e[q], k); v = (q == r) ? vq : v; }
        return v;
    }

    // ---- step 1: decay (BASE:244-250) ----------------------------------------------
    PPG_MEMBER void decay(const uint64_t (&acted)[T]) {
        // Same-type co-occupancy check on the (still all-zero) channel maps used as claim boards.
        bool mism[T];
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) chmap(1 + type_of(r))[cell_of(xy[r])] = to_map(1 + type_of(r), validx(r, ln));
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r)
            mism[r] = ((alive[r] >> ln) & 1ull) && chmap(1 + type_of(r))[cell_of(xy[r])] != to_map(1 + type_of(r), validx(r, ln));
        uint64_t mm[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < T; ++r) mm[type_of(r)] |= wv::ballot(mism[r]);
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) chmap(1 + type_of(r))[cell_of(xy[r])] = 0;
        cooc[0] = mm[0] != 0;
        cooc[1] = mm[1] != 0;

#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((acted[r] >> ln) & 1ull) {
                e[r] -= (r ? C.loss_q : C.loss_p);
                if (GEN2) keep[r] &= ~(uint32_t)PPG_ROW_GRID_E0;  // RQ:489: the grid now shows the real energy
            }

        // grid[type, pos] = energy, in action order
#pragma unroll
        for (int type = 0; type < 2; ++type) {
            if (!cooc[type]) {
#pragma unroll
                for (int r = 0; r < T; ++r)
                    if (type_of(r) == type) owns[r] |= acted[r];  // one live agent per cell: each acting agent owns its cell
            } else if (ORDERED && C.act_rank) {
                const int n = publish_order(type, acted);
                for (int i = 0; i < n; ++i) {
                    int r, k;
                    ordered_row(type, i, r, k);
                    grid_set(r, k, xy_at(r, k), 0.0, false);
                }
            } else {
#pragma unroll
                for (int r = 0; r < T; ++r) {
                    if (type_of(r) != type) continue;
                    uint64_t m = acted[r];
                    while (m) {
                        const int k = wv::ctz(m);
                        m &= m - 1;
                        grid_set(r, k, wv::readlane(xy[r], k), 0.0, false);
                    }
                }
            }
        }
    }

    // ---- step 2: movement in action order (BASE:259-276, _get_move BASE:495-509) ------
    // What an acting row wants, computed for all rows at once before anybody moves (nothing another agent does changes it:
    // _get_move reads the agent's own position and action only, BASE:495-505): bits 0-15 the clipped target cell, bits 16-20 the
    // squared displacement (second generation: the move's energy cost, RQ:301-313), bits 24-26 what the walls say (WO:466-488).
    // A target the walls refuse as a WALL cell is the agent's own cell (WO:469-471).
    PPG_MEMBER uint32_t move_wish(int r, bool acts) const {
        const int G1 = P.G - 1;
        int dx = 0, dy = 0;
        if (act[r] >= 0) move_vector(act[r], GEN2 && ((id[r] >> 16) & 1), dx, dy);
        const int x = (int)(xy[r] >> 8), y = (int)(xy[r] & 255u);
        int tx = x + dx, ty = y + dy;
        tx = tx < 0 ? 0 : (tx > G1 ? G1 : tx);  // np.clip, BASE:505
        ty = ty < 0 ? 0 : (ty > G1 ? G1 : ty);
        uint32_t verdict = MV_NONE;
        if (WALLS && acts) {
            verdict = (C.ar[0] <= 5 && C.ar[1] <= 5) ? wall_verdict_near(x, y, tx, ty) : wall_verdict(x, y, tx, ty);
            if (verdict == MV_WALL) { tx = x; ty = y; }
        }
        const int ddx = tx - x, ddy = ty - y;
        return ((uint32_t)tx << 8) | (uint32_t)ty | (((uint32_t)(ddx * ddx + ddy * ddy) & 31u) << 16) | (verdict << 24);
    }
    PPG_MEMBER uint32_t wish_at(const uint32_t (&wish)[T], int r, int k) const {
        uint32_t v = wv::readlane(wish[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane(wish[q], k); v = (q == r) ? vq : v; }
        return v;
    }

    // One agent at its turn: row (r,k); r may be a run-time register index (explicit-order path) -- with a
    // compile-time r every (q == r) below folds away.  moved[]: rows that changed cell (their move cost is charged by the caller,
    // lane-parallel); sp[]: rows whose energy after that cost still shows as positive on the grid.
    PPG_MEMBER void move_agent(int r, int k, uint64_t (&pos)[T], const uint32_t (&wish)[T], uint64_t (&moved)[T], const uint64_t (&sp)[T], bool costly) {
        const int type = type_of(r);
        const uint32_t s_xy = xy_at(r, k);
        const uint32_t w = wish_at(wish, r, k);
        const uint32_t t_xy = w & 0xFFFFu;
        const uint32_t verdict = w >> 24;
        uint64_t mt[T], mo[T];
        match(type, t_xy, mt);
        uint64_t occ = 0;  // grid[type, target] > 0 (BASE:506): an owner with positive energy sits there
#pragma unroll
        for (int q = 0; q < T; ++q) occ |= mt[q] & owns[q] & pos[q];
        if (WALLS) {  // WO:466-488: wall, then occupied, then corner cutting / line of sight
            const uint32_t reason = verdict == MV_WALL ? (uint32_t)MV_WALL : (occ ? (uint32_t)MV_OCCUPIED : verdict);
#pragma unroll
            for (int q = 0; q < T; ++q)
                if (q == r && ln == k) set_move_info(q, reason);
            if (verdict == MV_CORNER_CUT || verdict == MV_LOS) occ = 1;  // refused like an occupied target: stay
        }
        if (t_xy == s_xy) {
#pragma unroll
            for (int q = 0; q < T; ++q) mo[q] = mt[q];
        } else if (!cooc[type]) {   // no cell holds two live agents of this type: the agent is alone on its cell
#pragma unroll
            for (int q = 0; q < T; ++q) mo[q] = (q == r) ? bit64(k) : 0ull;
        } else {
            match(type, s_xy, mo);
        }
        // grid[old] = 0 (BASE:268/272)
#pragma unroll
        for (int q = 0; q < T; ++q) owns[q] &= ~mo[q];
        uint64_t others = 0;
        if (occ) {  // stay (BASE:506-507): grid[old] = energy
#pragma unroll
            for (int q = 0; q < T; ++q) others |= mo[q] & ~((q == r) ? bit64(k) : 0ull);
        } else {    // move: grid[new] = energy (BASE:269/273)
#pragma unroll
            for (int q = 0; q < T; ++q) {
                xy[q] = (q == r) ? wv::writelane(xy[q], k, t_xy) : xy[q];
                owns[q] &= ~mt[q];
                others |= mt[q] & ~((q == r) ? bit64(k) : 0ull);
            }
            if (GEN2 && costly && t_xy != s_xy) {
                // _get_movement_energy_cost (RQ:301-313) is paid before the grid write (RQ:526,538): the grid shows the energy after it
#pragma unroll
                for (int q = 0; q < T; ++q)
                    if (q == r) { moved[q] |= bit64(k); pos[q] = (pos[q] & ~bit64(k)) | (sp[q] & bit64(k)); }
            }
        }
#pragma unroll
        for (int q = 0; q < T; ++q) owns[q] |= (q == r) ? bit64(k) : 0ull;
        if (others) cooc[type] = true;
    }

    // one more agent touches the cell (its own, or the target of its move): counts on the -- at this point all-zero -- channel maps
    PPG_MEMBER void touch(int ch, uint32_t s_xy) { wv::lds_count(chmap(ch) + cell_of(s_xy)); }

    PPG_MEMBER void move(const uint64_t (&acted)[T]) {
        uint64_t pos[T];  // grid value > 0 requires the owner's energy > 0 (BASE:506)
        uint32_t wish[T];
        uint64_t moved[T], sp[T];
        const bool costly = GEN2 && C.move_factor != 0.0;
        // distance * factor per squared displacement (RQ:310-312), once per wavefront in the LDS scratch instead of a chain of selects
        // per row register.  BEHIND the explicit-order path's row lists (publish_order: predators at bytes 0..127, prey at
        // 128..128 + 2 * cap_prey <= 640): bytes 640..895 of a scratch that is at least 1024 bytes in every layout (ppg_host.h).
        double *cost = (double *)scr + 80;
        if (costly) {
            if (ln < 32) cost[ln] = ln < 19 ? move_distance(ln) * C.move_factor : 0.0;   // (rows not in use index anything below 32)
            wv::sync();
        }
#pragma unroll
        for (int r = 0; r < T; ++r) {
            pos[r] = wv::ballot(shown_positive(r)) & alive[r];
            wish[r] = move_wish(r, (acted[r] >> ln) & 1ull);
            moved[r] = 0; sp[r] = 0;
            if (costly) sp[r] = wv::ballot((float)(e[r] - cost[(wish[r] >> 16) & 31u] * e[r]) > 0.0f);
        }
        if (ORDERED && C.act_rank) {
#pragma unroll
            for (int type = 0; type < 2; ++type) {
                const int n = publish_order(type, acted);
                for (int i = 0; i < n; ++i) {
                    int r, k;
                    ordered_row(type, i, r, k);
                    move_agent(r, k, pos, wish, moved, sp, costly);
                }
            }
        } else {
            // An agent whose old cell and target cell are touched by no other agent of its type commutes
            // with all others: its move cannot be blocked (an empty target cell holds 0, see the header)
            // and nobody reads or writes its cells.  Those agents move lane-parallel; only agents that
            // share a cell with someone (contested target, target occupied, someone entering my cell)
            // go through the ordered loop.  Who touches a cell is COUNTED on the (all-zero) channel maps,
            // both species at once (they never meet on a channel): every live agent counts on its own cell,
            // every agent that wants to leave on its target; alone = both counts are 1.
#ifdef PPG_PROFILE_MOVE   // (diagnostic build: cycles of the lane-parallel part / of the ordered loop, agents in the ordered loop)
            const unsigned long long mv_t0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
            bool mover[T];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                mover[r] = ((acted[r] >> ln) & 1ull) && (wish[r] & 0xFFFFu) != xy[r];
                if ((alive[r] >> ln) & 1ull) touch(1 + type_of(r), xy[r]);
                if (mover[r]) touch(1 + type_of(r), wish[r] & 0xFFFFu);
            }
            wv::sync();
            uint64_t todo[T];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                map_t *A = chmap(1 + type_of(r));
                bool c = false;
                if ((alive[r] >> ln) & 1ull) c = A[cell_of(xy[r])] != 1 || (mover[r] && A[cell_of(wish[r] & 0xFFFFu)] != 1);
                // a move the walls refuse still has to see whether its target is occupied at its turn (the reported
                // reason depends on it, WO:472-488): ordered loop
                if (WALLS && ((wish[r] >> 24) == MV_CORNER_CUT || (wish[r] >> 24) == MV_LOS)) c = true;
                todo[r] = (cooc[type_of(r)] ? ~0ull : wv::ballot(c)) & acted[r];
            }
#pragma unroll
            for (int r = 0; r < T; ++r) {   // (in program order behind the reads above: one wavefront's LDS accesses do not overtake each other)
                map_t *A = chmap(1 + type_of(r));
                if ((alive[r] >> ln) & 1ull) {
                    A[cell_of(xy[r])] = 0;
                    if (mover[r]) A[cell_of(wish[r] & 0xFFFFu)] = 0;
                }
            }
            wv::sync();
#pragma unroll
            for (int r = 0; r < T; ++r) {
                const uint64_t simple = acted[r] & ~todo[r];
                if (WALLS && ((simple >> ln) & 1ull))  // nobody else touches its cells: the target is free, or its own cell
                    set_move_info(r, (wish[r] >> 24) == MV_WALL ? (uint32_t)MV_WALL
                                     : (!mover[r] && shown_positive(r)) ? (uint32_t)MV_OCCUPIED : (uint32_t)MV_NONE);
                if ((simple >> ln) & 1ull) xy[r] = wish[r] & 0xFFFFu;   // BASE:263
                if (costly) moved[r] = simple & wv::ballot(mover[r]);
                owns[r] |= simple;                            // grid[new] = energy, BASE:269/273
            }
#ifdef PPG_PROFILE_MOVE
            const unsigned long long mv_t1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            unsigned long long mv_n = 0, mv_all = 0;
#pragma unroll
            for (int r = 0; r < T; ++r) { mv_n += (unsigned long long)wv::popc(todo[r]); mv_all += (unsigned long long)wv::popc(acted[r]); }
#endif
#pragma unroll
            for (int r = 0; r < T; ++r) {
                uint64_t m = todo[r];
                while (m) {
                    const int k = wv::ctz(m);
                    m &= m - 1;
                    move_agent(r, k, pos, wish, moved, sp, costly);
                }
            }
#ifdef PPG_PROFILE_MOVE
            const unsigned long long mv_t2 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (C.prof && ln == 0) {
                C.prof[(size_t)b * 16 + 13] = mv_t1 - mv_t0; C.prof[(size_t)b * 16 + 14] = mv_t2 - mv_t1; C.prof[(size_t)b * 16 + 15] = (mv_all << 16) | mv_n;
            }
#endif
        }
        if (costly) {   // RQ:301-313,526: distance * factor * energy, for every agent that changed cell
#pragma unroll
            for (int r = 0; r < T; ++r)
                if ((moved[r] >> ln) & 1ull) e[r] = e[r] - cost[(wish[r] >> 16) & 31u] * e[r];
        }
    }

    // ---- drop last call's dead rows and bring the rows into self.agents order ----------
    // (BASE:222-225 removal; BASE:468 sort).  Rows are [sorted prefix..., appended rows...];
    // appended rows are inserted by counting smaller keys with ballots.
    PPG_MEMBER void compact_and_sort(bool do_sort) {
#pragma unroll
        for (int type = 0; type < 2; ++type) {
            // sorted-prefix length m over this type's rows
            int m_sorted = n_rows[type];
            bool has_dead = false;
#pragma unroll
            for (int r = 0; r < T; ++r)
                if (type_of(r) == type) has_dead = has_dead || ((rows[r] & ~alive[r]) != 0);
            if (do_sort) {
#pragma unroll
                for (int r = T - 1; r >= 0; --r) {
                    if (type_of(r) != type) continue;
                    uint32_t prev = wv::shfl_up1(key[r]);
                    if (r >= 2) {
                        uint32_t carry = wv::readlane(key[r - 1], 63);
                        if (ln == 0) prev = carry;
                    }
                    const bool first_row = (row_of(r, ln) == 0);
                    uint64_t brk = wv::ballot(!first_row && key[r] < prev) & rows[r];
                    if (brk) m_sorted = row_of(r, wv::ctz(brk));
                }
            }
            if (!has_dead && m_sorted >= n_rows[type]) continue;  // nothing to do

            uint32_t rk[T];
            uint64_t sorted_alive[T], unsorted_alive[T];
            int before = 0;
#pragma unroll
            for (int r = 0; r < T; ++r) {
                rk[r] = 0; sorted_alive[r] = 0; unsorted_alive[r] = 0;
                if (type_of(r) != type) continue;
                const int lo = row_of(r, 0);
                uint64_t in_prefix = lowmask(m_sorted - lo);
                sorted_alive[r] = alive[r] & in_prefix;
                unsorted_alive[r] = alive[r] & ~in_prefix;
                rk[r] = (uint32_t)before + wv::prefix(sorted_alive[r]);
                before += wv::popc(sorted_alive[r]);
            }
#pragma unroll
            for (int ru = 0; ru < T; ++ru) {
                if (type_of(ru) != type) continue;
                uint64_t mu = unsorted_alive[ru];
                while (mu) {
                    const int ku = wv::ctz(mu);
                    mu &= mu - 1;
                    const uint32_t s_key = wv::readlane(key[ru], ku);
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < T; ++r) {
                        if (type_of(r) != type) continue;
                        cnt += wv::popc(wv::ballot(key[r] < s_key) & alive[r]);
                        if (((sorted_alive[r] >> ln) & 1ull) && key[r] > s_key) rk[r] += 1;
                    }
#pragma unroll
                    for (int r = 0; r < T; ++r)
                        if (r == ru) rk[r] = wv::writelane(rk[r], ku, (uint32_t)cnt);
                }
            }
            // scatter through LDS, one 8-byte field at a time
            const int sbase = 0;   // (one species at a time)
            int n_new = 0;
#pragma unroll
            for (int r = 0; r < T; ++r)
                if (type_of(r) == type) n_new += wv::popc(alive[r]);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (f == 1 && !CARRY_CUM) continue;  // (the cumulative reward: only the cooperative kernels carry it in registers)
#pragma unroll
                for (int r = 0; r < T; ++r) {
                    if (type_of(r) != type) continue;
                    if ((alive[r] >> ln) & 1ull) {
                        uint64_t v;
                        if (f == 0) v = (uint64_t)__double_as_longlong(e[r]);
                        else if (f == 1) v = (uint64_t)__double_as_longlong(cum[r]);
                        else if (f == 2) v = ((uint64_t)key[r] << 32) | (uint32_t)id[r];
                        else v = (uint64_t)xy[r] | ((uint64_t)((owns[r] >> ln) & 1ull) << 16) | ((uint64_t)keep[r] << 20);
                        scr[sbase + rk[r]] = v;
                    }
                }
                wv::sync();
#pragma unroll
                for (int r = 0; r < T; ++r) {
                    if (type_of(r) != type) continue;
                    const int i = row_of(r, ln);
                    if (i < n_new) {
                        uint64_t v = scr[sbase + i];
                        if (f == 0) e[r] = __longlong_as_double((long long)v);
                        else if (f == 1) cum[r] = __longlong_as_double((long long)v);
                        else if (f == 2) { key[r] = (uint32_t)(v >> 32); id[r] = (int32_t)(uint32_t)v; }
                        else { xy[r] = (uint32_t)(v & 0xFFFFu); ev[r] = (uint32_t)((v >> 16) & 1u); keep[r] = (uint32_t)(v >> 20) & 0x3FFFFu; }
                    } else if (f == 3) {
                        xy[r] = 0xFFFFu; ev[r] = 0; keep[r] = 0;
                    }
                }
                wv::sync();
            }
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if (type_of(r) != type) continue;
                rows[r] = lowmask(n_new - row_of(r, 0));
                alive[r] = rows[r];
                owns[r] = wv::ballot(ev[r] & 1u) & rows[r];
                ev[r] = 0;
            }
            n_rows[type] = n_new;
        }
    }

    // GEN2, after the rows have their final order: type masks, and agent_last_reproduction of every surviving row,
    // read from HBM at the row's start-of-step slot like the cumulative rewards
    PPG_MEMBER void after_compact() {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            t2m[r] = wv::ballot((id[r] >> 16) & 1) & rows[r];
            lr[r] = ((alive[r] >> ln) & 1ull) ? C.row_lastrep[(size_t)b * P.S + (keep[r] >> 8)] : 0;
        }
    }

    // ---- LDS acceleration structure for observations --------------------------------
    PPG_MEMBER void build_maps() {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            if ((alive[r] >> ln) & 1ull) {
                val[validx(r, ln)] = shown(r);
                if ((owns[r] >> ln) & 1ull) chmap(1 + type_of(r))[cell_of(xy[r])] = to_map(1 + type_of(r), validx(r, ln));
            }
        }
        wv::sync();
    }

    // _get_observation (BASE:511-526) + _obs_clip (BASE:528-539) for the agent of `type` in
    // per-type row j standing on s_xy; coalesced 16-byte stores of the (4,R,R) block.
    //
    // Lane l of chunk ch produces elements e = 128*ch + 2l and e+1 of the block (C order: channel,
    // i, j).  Everything that depends only on (R, G, e) is precomputed on the host into one LDS word
    // per element:  bits 0-15  moff = c*map_n + (i-off)*G + (j-off)   (signed; map index relative to
    //               the observer's cell),  bits 16-19 (i-off)+8,  bits 20-23 (j-off)+8,  bits 24-25 c,
    //               bit 26 element exists (e < 4*R*R),  bit 27 inside the (2*off+1)^2 window.
    // A row whose window lies inside the grid takes the branch-uniform fast path: value =
    // val[map[moff + cell]], no bounds checks (channel 0 reads the all-zero map 0).
    // FASTOBS version: descriptors in registers, all map reads issued together, then all value reads,
    // then the stores -- two LDS latencies per row.
    template <int TYPE>
    PPG_MEMBER void obs_row_fast(int j, uint32_t s_xy) {
        constexpr int NCH = TYPE ? 3 : 2;
        constexpr int BASE = TYPE ? 4 : 0;
        wv::sync();
        const int R = TYPE ? P.Rq : P.Rp;
        const int blk = 4 * R * R;
        const int off = (R - 1) / 2;
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int s_cell = x * P.G + y;
        const bool interior = (R & 1) && x >= off && y >= off && x + off < P.G && y + off < P.G;
        const size_t obase = ((size_t)b * (TYPE ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)blk;
        uint32_t idx[NCH][2];
        bool one[NCH][2];
        double v[NCH][2];
        if (interior) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = lutr[BASE + 2 * c + h];
                    idx[c][h] = (uint32_t)map[(int)(int16_t)(w & 0xFFFFu) + s_cell] + (MAP8 ? (w >> 30) * 129u : 0u);  // bits 30-31: section
                    one[c][h] = false;
                }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = lutr[BASE + 2 * c + h];
                    const int gx = x + (int)((w >> 16) & 15u) - 8, gy = y + (int)((w >> 20) & 15u) - 8;
                    const bool inb = (w & 0x8000000u) && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                    idx[c][h] = (uint32_t)map[inb ? (int)(int16_t)(w & 0xFFFFu) + s_cell : 0] + ((MAP8 && inb) ? (w >> 30) * 129u : 0u);
                    one[c][h] = !inb && (w & 0x3000000u) == 0u;  // channel 0 outside the grid (BASE:522-523)
                }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[c][h] = val[idx[c][h]];
#ifdef PPG_EXP_NO_OBS_READS  // ablation build only (tools/exp_variants.py): stores without LDS lookups
#pragma unroll
        for (int c = 0; c < NCH; ++c) { v[c][0] = 0.0; v[c][1] = 0.0; }
#endif
#ifdef PPG_EXP_NO_OBS_STORES  // ablation build only: no observation stores at all
        if (P.batch > 0) { wv::sync(); return; }
#endif
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (lutr[BASE + 2 * c] & 0x4000000u) {
                const double v0 = one[c][0] ? 1.0 : v[c][0], v1 = one[c][1] ? 1.0 : v[c][1];
                const size_t o = obase + (size_t)c * 128 + 2 * (size_t)ln;
                store_obs_pair(TYPE ? P.obs_prey : P.obs_pred, P.obs_f32, o, v0, v1);
            }
        }
        wv::sync();
    }

    // ---- drive channels (DRV:551-616) ------------------------------------------------------------
    // np.sum over the n staged float64 values win[lo .. lo+n): numpy's pairwise summation (plain loop below 8 elements,
    // eight interleaved accumulators up to 128, two halves above) -- the order of the additions is part of the result.
    PPG_MEMBER double np_sum_block(const double *win, int lo, int n) const {
        if (n < 8) {
            double res = 0.0;
            for (int i = 0; i < n; ++i) res += first_f64(win[lo + i]);
            return res;
        }
        const int n8 = n - (n & 7);
        double acc = 0.0;
        if (ln < 8) {
            acc = win[lo + ln];
            for (int i = 8 + ln; i < n8; i += 8) acc += win[lo + i];
        }
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = readlane_f64(acc, q);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = n8; i < n; ++i) res += first_f64(win[lo + i]);
        return res;
    }
    // np.sum(observation[ch]) for the agent of `type` standing on s_xy (DRV:601-608)
    PPG_MEMBER double window_sum(int type, int ch, uint32_t s_xy) {
        // (not `type ? P.Rq : P.Rp`: with a run-time type hipcc selects the fields' ADDRESSES and spills both to scratch)
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0), n = R * R;
#ifdef PPG_EXP_DRIVE_NO_SUM  // ablation build only: what the window sums cost altogether
        if (P.batch > 0) return 0.0;
#endif
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int s_cell = x * P.G + y;
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;   // one staging area per wave of a multi-wave workgroup
        double *win = (double *)((unsigned char *)map + C.off_win - P.off_map) + wave_idx * 4 * rmax * rmax;
        const uint32_t *L = lut + (type ? P.nch_p * 128 : 0);
        const bool strided = (((4 + (type ? C.n_drive[1] : C.n_drive[0])) * n) & 1) != 0;   // see obs_row / ppg_build_lut
        wv::sync();
        for (int i = ln; i < n; i += 64) {
            const int el = ch * n + i, w7 = el & 127;
            const uint32_t w = L[strided ? (el & ~127) + (w7 & 63) * 2 + (w7 >> 6) : el];
            const int gx = x + (int)((w >> 16) & 15u) - 8, gy = y + (int)((w >> 20) & 15u) - 8;
            const bool inb = (w & 0x8000000u) && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
            win[i] = val[from_map(ch, map[inb ? (int)(int16_t)(w & 0xFFFFu) + s_cell : 0])];
        }
        wv::sync();
        double res;
#ifdef PPG_EXP_DRIVE_NO_REDUCE  // ablation build only: staging without the ordered reduction
        if (P.batch > 0) { res = first_f64(win[0]); wv::sync(); return res; }
#endif
        if (n <= 128) {
            res = np_sum_block(win, 0, n);
        } else {
            int n2 = n / 2;
            n2 -= n2 & 7;
            const double a = np_sum_block(win, 0, n2);
            res = a + np_sum_block(win, n2, n - n2);
        }
        wv::sync();
        return res;
    }
    // _safe_clip01 (DRV:612-615)
    static PPG_MEMBER double safe_clip01(double v) {
        if (!(v - v == 0.0)) return 0.0;
        return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    }
    // the drive features of one agent (DRV:577-610); s_e = its energy at this moment
    PPG_MEMBER void drive_features(int type, double s_e, uint32_t s_xy, double (&dv)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dv[k] = 0.0;
            if (k >= (type ? C.n_drive[1] : C.n_drive[0])) continue;
            const int kind = type ? C.drive_kind[1][k] : C.drive_kind[0][k];
            double v;
            if (kind == 0) v = 1.0 - s_e / (type ? C.hunger_safe[1] : C.hunger_safe[0]);
            else if (kind == 1) v = s_e / (type ? C.thr_q : C.thr_p);
            else if (kind == 2) v = window_sum(type, 2, s_xy) / C.norm_prey_opp;
            else if (kind == 3) v = window_sum(type, 1, s_xy) / C.norm_pred_danger;
            else v = window_sum(type, 3, s_xy) / C.norm_grass_opp;
            dv[k] = safe_clip01(v);
        }
    }

    // _get_observation of the walls env (WO:527-601), one window CELL per lane (64 cells per pass): in-grid test, wall bit,
    // line-of-sight bit and the three channel lookups are done once per cell and feed all 4 / 5 channels -- the per-element
    // formulation below does that work once per channel.  Channel 0 = walls inside the window (0 outside the grid); channels
    // 1-3 optionally multiplied, in float32 like the reference, by the mask; optional last channel = the mask itself, which is
    // computed for every cell of the R x R array that maps into the grid (also the last row / column of an even R, which the
    // window copy WO:543 leaves untouched).  Consecutive lanes write consecutive elements of a channel plane.
    PPG_MEMBER void obs_row_walls(int type, int j, uint32_t s_xy) {
        wv::sync();  // LDS writes of the sequential phases -> visible
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);   // (arithmetic, not a select of fields: see window_sum)
        const uint32_t rmagic = C.rp_magic + (type ? C.rq_magic - C.rp_magic : 0u);
        const int n = R * R, off = (R - 1) / 2, Wc = 2 * off + 1;
        const int nchan = C.vis_channel ? 5 : 4;
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;   // (areas are strided by the larger window: waves work on both species)
        float *visb = (float *)((unsigned char *)map + C.off_win - P.off_map) + wave_idx * rmax * rmax;
        const uint32_t *visw = (const uint32_t *)visb;
        const bool want_vis = C.mask_obs || C.vis_channel;
        const bool have_masks = C.vis_masks != nullptr;
        if (want_vis && have_masks) {
            // walls are static: the mask of this agent's cell was computed when they were set (ppg_walls_changed) -- a few words
            // instead of one Bresenham walk per window cell
            if (ln < C.vis_words) ((uint32_t *)visb)[ln] = C.vis_masks[((size_t)b * P.G * P.G + x * P.G + y) * C.vis_words + ln];
            wv::sync();
        } else if (want_vis) {
            for (int i = ln; i < n; i += 64) {
                const int ci = (int)wv::mulhi((uint32_t)i, rmagic), cj = i - ci * R;
                const int gx = x - off + ci, gy = y - off + cj;
                const bool in_grid = (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                visb[i] = (in_grid && los_clear(x, y, gx, gy)) ? 1.0f : 0.0f;
            }
            wv::sync();
        }
        const size_t obase = ((size_t)b * (type ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)(nchan * n);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int cell = c0 + ln;
            const bool valid = cell < n;
            const int ci = (int)wv::mulhi((uint32_t)cell, rmagic), cj = cell - ci * R;
            const int gx = x - off + ci, gy = y - off + cj;
            const bool in_grid = valid && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
            const bool inb = in_grid && ci < Wc && cj < Wc;
            const int a = inb ? gx * P.G + gy : 0;
            double v[5];
            v[0] = (inb && ((wallw[a >> 5] >> (a & 31)) & 1u)) ? 1.0 : 0.0;
            float vis = 0.0f;
            if (want_vis && in_grid) {
                if (have_masks) {
                    const int bi = (ci - off + C.vis_neg) * C.vis_w + (cj - off + C.vis_neg);
                    vis = ((visw[bi >> 5] >> (bi & 31)) & 1u) ? 1.0f : 0.0f;
                } else {
                    vis = visb[cell];
                }
            }
#pragma unroll
            for (int ch = 1; ch < 4; ++ch) {
                double t = val[from_map(ch, chmap(ch)[a])];
                if (!inb) t = 0.0;
                if (C.mask_obs) t = (double)((float)t * (inb ? vis : 0.0f));
                v[ch] = t;
            }
            v[4] = (double)vis;
            if (valid) {
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) {
                    if (ch >= nchan) continue;
                    const size_t o = obase + (size_t)ch * n + cell;
                    if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[o] = (float)v[ch];
                    else ((double *)(type ? P.obs_prey : P.obs_pred))[o] = v[ch];
                }
            }
        }
        wv::sync();  // reads done before the caller touches the maps again
    }

    // _get_observation of the drive-conditioned env (DRV:551-616), one window CELL per lane: the three world channels of a cell
    // are looked up once, stored, and staged in LDS for the window sums -- np.sum(observation[c]) in numpy's order: eight
    // interleaved accumulators r_q = a[q] + a[q+8] + ... (lanes 8g .. 8g+7 of lane group g = channel g+1 run them side by side for
    // all three channels), combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by three xor-shuffles (IEEE addition commutes, so
    // every lane of a group ends with the same bits), then the tail elements one by one.  Needs 8 <= R*R <= 128 (numpy switches
    // to a plain loop below and to recursive halves above); other sizes take the per-element path.
    // Tried and slower (per 1365-env launch, against this version = 1.00): staging all four planes and writing the block in
    // element order with 16-byte stores 1.11 (the second pass over LDS costs more than the wider stores save); two adjacent cells
    // per lane with aligned pair stores 1.07 (shuffles for the odd planes, 41 of 64 lanes busy); element order through the
    // descriptor table with the staging folded into the same pass 1.11 (one lookup per ELEMENT instead of per cell, also for
    // plane 0).  The per-cell lookups are what this path is bound by, not the width of its stores.
    PPG_MEMBER void obs_row_drive(int type, int j, uint32_t s_xy, double s_e) {
        wv::sync();
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);
        const uint32_t rmagic = C.rp_magic + (type ? C.rq_magic - C.rp_magic : 0u);
        const int n = R * R, off = (R - 1) / 2, Wc = 2 * off + 1;
        const int nd = type ? C.n_drive[1] : C.n_drive[0];
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq, rmax2 = rmax * rmax;
        double *win = (double *)((unsigned char *)map + C.off_win - P.off_map) + wave_idx * 4 * rmax2;
        const size_t obase = ((size_t)b * (type ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)((4 + nd) * n);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int cell = c0 + ln;
            const bool valid = cell < n;
            const int ci = (int)wv::mulhi((uint32_t)cell, rmagic), cj = cell - ci * R;
            const int gx = x - off + ci, gy = y - off + cj;
            const bool inb = valid && ci < Wc && cj < Wc && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
            const int a = inb ? gx * P.G + gy : 0;
            double v[4];
            v[0] = inb ? 0.0 : 1.0;                      // DRV:561-562: 1 everywhere except the in-grid part of the window
#pragma unroll
            for (int ch = 1; ch < 4; ++ch) {
                const double t = val[from_map(ch, chmap(ch)[a])];
                v[ch] = inb ? t : 0.0;
            }
            if (valid) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const size_t o = obase + (size_t)ch * n + cell;
                    if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[o] = (float)v[ch];
                    else ((double *)(type ? P.obs_prey : P.obs_pred))[o] = v[ch];
                    if (ch) win[(ch - 1) * rmax2 + cell] = v[ch];
                }
            }
        }
        wv::sync();
        // the three window sums, side by side
        const int grp = ln >> 3, q = ln & 7, n8 = n & ~7;
        const double *wc = win + (grp < 3 ? grp : 0) * rmax2;
        double acc = wc[q];
        for (int i = 8 + q; i < n8; i += 8) acc += wc[i];
        acc = acc + wv::shfl_xor_f64(acc, 1);
        acc = acc + wv::shfl_xor_f64(acc, 2);
        acc = acc + wv::shfl_xor_f64(acc, 4);
        for (int i = n8; i < n; ++i) acc += wc[i];
        const double sum1 = readlane_f64(acc, 0), sum2 = readlane_f64(acc, 8), sum3 = readlane_f64(acc, 16);
        double dv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= nd) continue;
            const int kind = type ? C.drive_kind[1][k] : C.drive_kind[0][k];
            double t;
            if (kind == 0) t = 1.0 - s_e / (type ? C.hunger_safe[1] : C.hunger_safe[0]);
            else if (kind == 1) t = s_e / (type ? C.thr_q : C.thr_p);
            else if (kind == 2) t = sum2 / C.norm_prey_opp;
            else if (kind == 3) t = sum1 / C.norm_pred_danger;
            else t = sum3 / C.norm_grass_opp;
            dv[k] = safe_clip01(t);
        }
        // the drive planes: one scalar per plane (DRV:566-569).  The nd planes are ONE contiguous run of nd * n elements: written as
        // element pairs (16-byte stores for float64), behind one leading single element when the run starts on an odd element
        {
            const size_t start = obase + (size_t)4 * n;
            const int len = nd * n, sh = (int)(start & 1);
            auto plane_value = [&](int e) { return e < n ? dv[0] : e < 2 * n ? dv[1] : e < 3 * n ? dv[2] : dv[3]; };
            if (sh && ln == 0) {
                if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[start] = (float)dv[0];
                else ((double *)(type ? P.obs_prey : P.obs_pred))[start] = dv[0];
            }
            for (int e = sh + 2 * ln; e < len; e += 128) {
                const double v0 = plane_value(e), v1 = plane_value(e + 1);
                if (e + 1 < len) {
                    if (P.obs_f32) {
                        float2 f; f.x = (float)v0; f.y = (float)v1;
                        *(float2 *)((float *)(type ? P.obs_prey : P.obs_pred) + start + e) = f;
                    } else {
                        double2 g; g.x = v0; g.y = v1;
                        *(double2 *)((double *)(type ? P.obs_prey : P.obs_pred) + start + e) = g;
                    }
                } else if (P.obs_f32) {
                    ((float *)(type ? P.obs_prey : P.obs_pred))[start + e] = (float)v0;
                } else {
                    ((double *)(type ? P.obs_prey : P.obs_pred))[start + e] = v0;
                }
            }
        }
        wv::sync();
    }

    // ---- COOP: observations as whole 1 KB pieces of an env's run of live rows ---------------------------------
    // The live rows of `type` of the env whose LDS region is `region` are listed in `list` (n_live words: row << 16 | the agent's
    // padded cell -- ch0_map 0: | x << 8 | y); concatenated they are a run of n_live * blk elements.  Piece p is elements 128 p ..
    // 128 p + 127 of the run: lane l produces elements 128 p + 2l and + 1 (blk is even: a pair never straddles two rows) --
    // BASE:511-526 per element: value = val[map[cell + offset of the element] + section of its channel]; the padded maps make the
    // window clipping of _obs_clip (BASE:528-539) implicit.  ch0_map 0: channel 0 is 1.0 iff the element's cell lies outside the grid
    // (BASE:520-523), from the agent's position and the element's window offsets alone.  This wavefront writes pieces first,
    // first + stride, ...
    PPG_MEMBER void coop_pieces(int type, const unsigned char *region, const uint32_t *list, int n_live, int eb, int first, int stride) {
        const map_t *m = (const map_t *)(region + P.off_map);
        const double *vt = (const double *)(region + P.off_val);
        const int blk = C.blk_p + (type ? C.blk_q - C.blk_p : 0);   // (arithmetic, not a select of fields: see window_sum)
        const uint32_t magic = C.bp_magic + (type ? C.bq_magic - C.bp_magic : 0u);
        const uint32_t *L = lut2 + (type ? C.blk_p : 0);
        const int total = n_live * blk;
        const size_t obase = (size_t)eb * (size_t)(type ? P.cap_prey : P.cap_pred) * (size_t)blk;
        // pieces in flight per wavefront: the three dependent LDS lookups of one hide behind the other's.  (Round 6, measured and not
        // kept: four in flight, and the first lookup of the next group issued beside the map reads of the group in hand -- 53.8 / 55.4
        // against 53.0 us per 4096-env step on 64x64 grids, 63.5 / 67.4 against 62.3 on the headline: the write phase is bound by how
        // fast the memory system takes the stores, not by this chain.  profiles/r06/b_*)
        constexpr int U = 2;
        if (CH0MAP) {   // four maps: every element is a map lookup
            const uint32_t safe_cell = (uint32_t)(P.pad * P.Gp + P.pad);   // lanes behind the end of the run look at cell (0,0): inside the maps
            for (int p0 = first; p0 * 128 < total; p0 += U * stride) {
                uint32_t o[U], i0[U], i1[U];
                bool on[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s0 = (p0 + u * stride) * 128 + 2 * ln;
                    on[u] = s0 < total;
                    const uint32_t sc = on[u] ? (uint32_t)s0 : 0u;
                    const uint32_t i = wv::mulhi(sc, magic), w = sc - wv::mul24(i, (uint32_t)blk);   // (rows x block elements < 2^24)
                    const uint32_t ent = on[u] ? list[i] : safe_cell;
                    const uint2 d = *(const uint2 *)(L + w);
                    const int pc = (int)(ent & 0xFFFFu);
                    i0[u] = (uint32_t)m[pc + (int)(int16_t)(d.x & 0xFFFFu)] + (d.x >> 16);
                    i1[u] = (uint32_t)m[pc + (int)(int16_t)(d.y & 0xFFFFu)] + (d.y >> 16);
                    o[u] = wv::mul24(ent >> 16, (uint32_t)blk) + w;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double v0 = vt[i0[u]], v1 = vt[i1[u]];
                    if (!on[u]) continue;
                    store_obs_pair(type ? P.obs_prey : P.obs_pred, P.obs_f32, obase + o[u], v0, v1);
                }
            }
            return;
        }
        const uint32_t G = (uint32_t)P.G;
        for (int p0 = first; p0 * 128 < total; p0 += U * stride) {
            uint32_t o[U], i0[U], i1[U];
            bool on[U], out0[U], out1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s0 = (p0 + u * stride) * 128 + 2 * ln;
                on[u] = s0 < total;
                const uint32_t sc = on[u] ? (uint32_t)s0 : 0u;
                const uint32_t i = wv::mulhi(sc, magic), w = sc - wv::mul24(i, (uint32_t)blk);
                const uint32_t ent = on[u] ? list[i] : 0u;   // (lanes behind the end of the run look at cell (0,0): inside the maps)
                const uint2 d = *(const uint2 *)(L + w);
                const uint32_t ax = (ent >> 8) & 255u, ay = ent & 255u;
                const int pc = (int)(wv::mul24(ax + (uint32_t)P.pad, (uint32_t)P.Gp) + ay + (uint32_t)P.pad);
                const bool z0 = (d.x >> 16) == 0xFFFFu, z1 = (d.y >> 16) == 0xFFFFu;   // channel 0: no map
                // (a channel-0 descriptor's low bits as a map offset stay inside the three maps: no lane reads outside LDS)
                const uint32_t m0 = (uint32_t)m[pc + (int)(int16_t)(d.x & 0xFFFFu)], m1 = (uint32_t)m[pc + (int)(int16_t)(d.y & 0xFFFFu)];
                i0[u] = z0 ? 0u : m0 + (d.x >> 16);
                i1[u] = z1 ? 0u : m1 + (d.y >> 16);
                // outside the grid: unsigned compares (a coordinate below 0 wraps far above G); computed for every lane, no branches
                const uint32_t tx0 = ax + ((d.x >> 4) & 15u) - 8u, ty0 = ay + (d.x & 15u) - 8u;
                const uint32_t tx1 = ax + ((d.y >> 4) & 15u) - 8u, ty1 = ay + (d.y & 15u) - 8u;
                out0[u] = z0 & ((tx0 > ty0 ? tx0 : ty0) >= G);
                out1[u] = z1 & ((tx1 > ty1 ? tx1 : ty1) >= G);
                o[u] = wv::mul24(ent >> 16, (uint32_t)blk) + w;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double t0 = vt[i0[u]], t1 = vt[i1[u]];   // (channel 0 inside the grid: entry 0 = 0.0)
                const double v0 = out0[u] ? 1.0 : t0, v1 = out1[u] ? 1.0 : t1;
                if (!on[u]) continue;
                store_obs_pair(type ? P.obs_prey : P.obs_pred, P.obs_f32, obase + o[u], v0, v1);
            }
        }
    }
    // a mid-step observation (an agent that starves or is caught, BASE:287,327): its block alone, at this point of the sequence
    PPG_MEMBER void obs_row_coop(int type, int j, uint32_t s_xy) {
        wv::sync();   // LDS writes of the sequential phases -> visible
        uint32_t *mid = ctl + CTL_MID + wave_idx;
        if (ln == 0) mid[0] = ((uint32_t)j << 16) | (CH0MAP ? (uint32_t)cell_of(s_xy) : s_xy);
        wv::sync();
        coop_pieces(type, (const unsigned char *)map - P.off_map, mid, 1, b, 0, 1);
        wv::sync();   // reads done before the caller touches the maps again
    }
    // the rows to observe at the end of the call, per species, in the env's scratch: [0] predators, [64] prey
    PPG_MEMBER void coop_publish() {
        uint32_t *lst = (uint32_t *)scr;
        int n[2] = {0, 0};
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int type = type_of(r);
            if ((alive[r] >> ln) & 1ull)
                lst[(type ? 64 : 0) + n[type] + (int)wv::prefix(alive[r])] = ((uint32_t)row_of(r, ln) << 16) | (CH0MAP ? (uint32_t)cell_of(xy[r]) : xy[r]);
            n[type] += wv::popc(alive[r]);
        }
        if (ln == 0) {
            uint32_t *slot = ctl + CTL_SLOT + 4 * wave_idx;
            slot[0] = (uint32_t)n[0]; slot[1] = (uint32_t)n[1]; slot[2] = (uint32_t)b;
        }
        wv::sync();
    }
    // after the workgroup barrier: all the workgroup's envs, piece p of the workgroup to wavefront p mod NW
    PPG_MEMBER void coop_write_all(const unsigned char *wg_lds) {
        int at = 0;   // pieces handed out so far, mod NW
        for (int k = 0; k < C.coop_e; ++k) {
            const uint32_t *slot = ctl + CTL_SLOT + 4 * k;
            const int eb = (int)wv::first(slot[2]);
            if (eb < 0) continue;
            const unsigned char *region = wg_lds + (size_t)k * C.lds_env_bytes;
            const uint32_t *lst = (const uint32_t *)(region + P.off_scr);
#pragma unroll
            for (int type = 0; type < 2; ++type) {
                const int n_live = (int)wv::first(slot[type]);
                const int blk = C.blk_p + (type ? C.blk_q - C.blk_p : 0);
                const int pieces = (n_live * blk + 127) >> 7;
                int first = wave_idx - at;
                if (first < 0) first += NW;
                coop_pieces(type, region, lst + (type ? 64 : 0), n_live, eb, first, NW);
                at = (at + pieces) % NW;
            }
        }
    }

    PPG_MEMBER void obs_row(int type, int j, uint32_t s_xy, double s_e = 0.0) {
        if (COOP) { obs_row_coop(type, j, s_xy); return; }
        if (FASTOBS) {
            if (type) obs_row_fast<1>(j, s_xy);
            else obs_row_fast<0>(j, s_xy);
            return;
        }
        if (WALLS) { obs_row_walls(type, j, s_xy); return; }
        if (DRIVE) {
            const int Rn = P.Rp + (type ? P.Rq - P.Rp : 0);
            if (Rn * Rn >= 8 && Rn * Rn <= 128) { obs_row_drive(type, j, s_xy, s_e); return; }
        }
        wv::sync();  // LDS writes of the sequential phases -> visible
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);   // (arithmetic, not a select of fields: see window_sum)
        double dv[4] = {0.0, 0.0, 0.0, 0.0};
        if (DRIVE) drive_features(type, s_e, s_xy, dv);
        const int blk = C.blk_p + (type ? C.blk_q - C.blk_p : 0);   // channels x R x R
        const int off = (R - 1) / 2;
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int s_cell = x * P.G + y;
        const bool interior = !WALLS && !DRIVE && (R & 1) && x >= off && y >= off && x + off < P.G && y + off < P.G;
        const uint2 *L = (const uint2 *)(lut + (type ? P.nch_p * 128 : 0));
        const int nch = type ? P.nch_q : P.nch_p;
        // walls variant: the line-of-sight mask of this agent, one value per cell of the R x R array (WO:577-589), staged in
        // LDS once and used by up to four channels (every wave of a multi-wave workgroup has its own staging area)
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;   // (areas are strided by the larger window: waves work on both species)
        float *visb = (float *)((unsigned char *)map + C.off_win - P.off_map) + wave_idx * rmax * rmax;
        const bool want_vis = WALLS && (C.mask_obs || C.vis_channel);
        const bool have_masks = WALLS && C.vis_masks != nullptr;
        const uint32_t *visw = (const uint32_t *)visb;
        if (want_vis && have_masks) {
            // walls are static: the mask of this agent's cell was computed when they were set (ppg_walls_changed) -- a few words
            // instead of one Bresenham walk per window cell
            if (ln < C.vis_words) ((uint32_t *)visb)[ln] = C.vis_masks[((size_t)b * P.G * P.G + s_cell) * C.vis_words + ln];
            wv::sync();
        } else if (want_vis) {
            for (int i = ln; i < R * R; i += 64) {
                const int ci = i / R, cj = i - ci * R;
                const int gx = x - off + ci, gy = y - off + cj;
                const bool in_grid = (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                visb[i] = (in_grid && los_clear(x, y, gx, gy)) ? 1.0f : 0.0f;
            }
            wv::sync();
        }
        const size_t obase = ((size_t)b * (type ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)blk;
        for (int ch = 0; ch < nch; ++ch) {
            const uint2 d = L[ch * 64 + ln];
            double v[2];
            if (interior) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = h ? d.y : d.x;
                    const int a = (int)(int16_t)(w & 0xFFFFu) + s_cell;
                    v[h] = val[from_map((int)((w >> 24) & 3u), map[a])];  // a non-existent element has moff 0: reads the observer's own cell in the all-zero map, unused
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = h ? d.y : d.x;
                    const int gx = x + (int)((w >> 16) & 15u) - 8, gy = y + (int)((w >> 20) & 15u) - 8;
                    const bool inb = (w & 0x8000000u) && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                    const int a = (int)(int16_t)(w & 0xFFFFu) + s_cell;
                    double t = val[from_map((int)((w >> 24) & 3u), map[inb ? a : 0])];   // map[0] (channel 0, cell 0) is always 0 -> val[0] = 0.0
                                                                                          // (a drive element carries its plane index there: its map entry is 0 anyway)
                    if (DRIVE && (w & 0x20000000u)) {       // a drive channel: the whole (R,R) plane holds one scalar (DRV:566-569)
                        const uint32_t k = (w >> 24) & 3u;
                        t = k == 0 ? dv[0] : k == 1 ? dv[1] : k == 2 ? dv[2] : dv[3];
                    } else if (!WALLS) {
                        if (!inb && (w & 0x3000000u) == 0u) t = 1.0;  // channel 0: 1 outside the grid (BASE:522-523)
                    } else {
                        // _get_observation of the walls env (WO:527-601): channel 0 = walls inside the window (0 outside the
                        // grid); channels 1-3 optionally multiplied -- in float32, like the reference -- by the line-of-
                        // sight mask; optional last channel = the mask itself
                        const bool vis_elem = (w & 0x10000000u) != 0u;
                        // the mask is computed for every cell of the R x R array that maps into the grid (WO:577-589), also
                        // for the last row / column of an even R, which the window copy (WO:543) leaves untouched
                        const bool in_grid = (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                        const bool need_vis = vis_elem ? in_grid : (inb && C.mask_obs && (w & 0x3000000u) != 0u);
                        const int vdx = (int)((w >> 16) & 15u) - 8, vdy = (int)((w >> 20) & 15u) - 8;
                        float vis = 0.0f;
                        if (need_vis && have_masks) {
                            const int bi = (vdx + C.vis_neg) * C.vis_w + (vdy + C.vis_neg);
                            vis = ((visw[bi >> 5] >> (bi & 31)) & 1u) ? 1.0f : 0.0f;
                        } else if (need_vis) {
                            vis = visb[(vdx + off) * R + (vdy + off)];
                        }
                        if (vis_elem) t = (double)vis;
                        else if ((w & 0x3000000u) == 0u) t = (inb && wall_at(gx, gy)) ? 1.0 : 0.0;
                        else if (C.mask_obs) t = (double)((float)t * vis);
                    }
                    v[h] = t;
                }
            }
            if ((WALLS || DRIVE) && (blk & 1)) {
                // an odd number of channels x an odd window: blocks start at odd element offsets, so element pairs cannot be
                // stored as aligned vectors.  For these geometries the host lays the descriptors out "strided": this lane's
                // two elements are ch*128 + ln and ch*128 + 64 + ln, i.e. each store instruction writes 64 consecutive
                // elements (ppg_build_lut)
                const size_t o = obase + (size_t)ch * 128 + (size_t)ln;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!((h ? d.y : d.x) & 0x4000000u)) continue;
                    if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[o + 64 * h] = (float)v[h];
                    else ((double *)(type ? P.obs_prey : P.obs_pred))[o + 64 * h] = v[h];
                }
            } else if (d.x & 0x4000000u) {
                const size_t o = obase + (size_t)ch * 128 + 2 * (size_t)ln;
                store_obs_pair(type ? P.obs_prey : P.obs_pred, P.obs_f32, o, v[0], v[1]);
            }
        }
        wv::sync();  // reads done before the caller touches the maps again
    }

    // multi-wave variants: rows of the published list, every stride-th one starting at `w`
    PPG_MEMBER void obs_shared(int w, int stride = NW) {
        const uint32_t *lst = (const uint32_t *)scr;
        const uint32_t head = wv::first(lst[0]);   // rows in the list | predators among them (they come first) << 16
        const int n = (int)(head & 0xFFFFu);
        if (FASTOBS) {
            // predators come first in the list: two loops with a compile-time species each (one loop with a run-time species keeps
            // both species' unrolled observation code and all ten descriptor registers live together: +20 registers)
            const int n_pred = (int)(head >> 16);
            int i = w;
            for (; i < n_pred; i += stride) {
                const uint32_t en = wv::first(lst[1 + i]);
                obs_row_fast<0>((int)((en >> 16) & 0x7FFFu), en & 0xFFFFu);
            }
            for (; i < n; i += stride) {
                const uint32_t en = wv::first(lst[1 + i]);
                obs_row_fast<1>((int)((en >> 16) & 0x7FFFu), en & 0xFFFFu);
            }
            return;
        }
        for (int i = w; i < n; i += stride) {
            const uint32_t en = wv::first(lst[1 + i]);
            const int ty = (int)(en >> 31), row = (int)((en >> 16) & 0x7FFFu);
            // (drive variant: the agent's energy is its entry of the LDS value table -- row energies are kept current there)
            const double s_e = DRIVE ? first_f64(val[validx_row(ty, row)]) : 0.0;
            obs_row(ty, row, en & 0xFFFFu, s_e);
        }
    }
    // a helper wave of a multi-wave workgroup: wait until wave 0 has finished the transition, then write its share
    PPG_MEMBER void run_helper(int w) {
        if (FASTOBS) {
            const uint2 *L2 = (const uint2 *)C.obs_lut;
#pragma unroll
            for (int c = 0; c < 2; ++c) { uint2 d; d.x = 0; d.y = 0; if (c < P.nch_p) d = L2[c * 64 + ln]; lutr[2 * c] = d.x; lutr[2 * c + 1] = d.y; }
#pragma unroll
            for (int c = 0; c < 3; ++c) { uint2 d; d.x = 0; d.y = 0; if (c < P.nch_q) d = L2[(P.nch_p + c) * 64 + ln]; lutr[4 + 2 * c] = d.x; lutr[5 + 2 * c] = d.y; }
        }
        wv::wg_barrier();
        obs_shared(w);
    }

    // Multi-wave kernels: the shared writing of the published rows.  It comes AFTER rewards_and_store: the row registers are dead by
    // then, which is what keeps these kernels inside 128 registers (with the stores behind the observation loops they spilled).
    static constexpr bool DEFER_OBS = NW > 1 && !COOP;
    PPG_MEMBER void obs_finish() {
        if (!DEFER_OBS) return;
        if (!ADAPTIVE_HELPERS || helpers) { wv::wg_barrier(); obs_shared(0); }
        else { wv::sync(); obs_shared(0, 1); }
    }

    // write_now = false (step paths of the multi-wave kernels): publish only, obs_finish() follows the table stores
    PPG_MEMBER void obs_all_alive(bool write_now = true) {
        if (COOP) { coop_publish(); return; }   // written by the whole workgroup after its barrier (env_main)
        if (NW > 1) {  // publish (type, row, cell) of every live row, then all waves of the workgroup share the rows
            // (an env whose helper waves have left -- ADAPTIVE_HELPERS -- goes through the same list with stride 1: a second, register-
            // indexed copy of the observation code in one kernel is what pushed the multi-wave kernels over 128 registers)
            uint32_t *lst = (uint32_t *)scr;
            int n = 0;
            wv::sync();
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if ((alive[r] >> ln) & 1ull)
                    lst[1 + n + (int)wv::prefix(alive[r])] = ((uint32_t)type_of(r) << 31) | ((uint32_t)row_of(r, ln) << 16) | xy[r];
                n += wv::popc(alive[r]);
            }
            if (ln == 0) lst[0] = (uint32_t)n | ((uint32_t)wv::popc(alive[0]) << 16);
            if (write_now) obs_finish();
            return;
        }
#pragma unroll
        for (int r = 0; r < T; ++r) {
            uint64_t m = alive[r];
            while (m) {
                const int k = wv::ctz(m);
                m &= m - 1;
                obs_row(type_of(r), row_of(r, k), wv::readlane(xy[r], k), DRIVE ? readlane_f64(e[r], k) : 0.0);
            }
        }
    }

    // ---- step 3: engagement in self.agents order (BASE:279-380) ----------------------
    PPG_MEMBER void starve(int r, int k, uint32_t s_xy) {  // BASE:284-301
        const int type = type_of(r);
        obs_row(type, row_of(r, k), s_xy, DRIVE ? e_at(r, k) : 0.0);
        if (ln == k) ev[r] |= EV_STARVED;
        n_alive[type] -= 1;
        grid_zero(type, s_xy, true);
#pragma unroll
        for (int q = 0; q < T; ++q) alive[q] &= ~((q == r) ? bit64(k) : 0ull);
    }

    PPG_MEMBER void engage_predators(uint64_t sel = ~0ull) {
        uint64_t m = alive[0] & sel;
        while (m) {
            const int k = wv::ctz(m);
            m &= m - 1;
            const double s_e = readlane_f64(e[0], k);
            const uint32_t s_xy = wv::readlane(xy[0], k);
            if (s_e <= 0.0) { starve(0, k, s_xy); continue; }
            uint64_t pm[T];
            match(1, s_xy, pm);
            int total = 0;
#pragma unroll
            for (int q = 1; q < T; ++q) total += wv::popc(pm[q]);
            if (total == 0) continue;  // reward_predator_step, BASE:341
            // first prey in agent_positions order == lowest id (ids are handed out in insertion order)
            // (GEN2: the creation number sits in the top bits of row_id, so the same comparison picks the first-inserted prey)
            int cr = 0, ck = 0;
            uint32_t best = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 1; q < T; ++q) {
                uint64_t mq = pm[q];
                while (mq) {
                    const int kk = wv::ctz(mq);
                    mq &= mq - 1;
                    const uint32_t cid = wv::readlane((uint32_t)id[q], kk);
                    if (cid < best) { best = cid; cr = q; ck = kk; }
                }
            }
            double pe = 0.0;
#pragma unroll
            for (int q = 1; q < T; ++q)
                if (q == cr) pe = readlane_f64(e[q], ck);
            double ne = s_e + pe;                           // BASE:324 (E1: pe may be <= 0)
            if (GEN2) {  // RQ:598-606: capped gain times the transfer efficiency, then the predator's energy cap
                const double raw = (C.cap_gain_prey < pe) ? C.cap_gain_prey : pe;
                ne = s_e + raw * C.eff_transfer;
                ne = (C.max_e_pred < ne) ? C.max_e_pred : ne;
                if (ln == k) keep[0] &= ~(uint32_t)PPG_ROW_GRID_E0;
            }
            e[0] = writelane_f64(e[0], k, ne);
            if (ln == k) ev[0] |= EV_ATE;                   // BASE:319
            grid_set(0, k, s_xy, ne, true);                 // BASE:325
            obs_row(1, row_of(cr, ck), s_xy, pe);           // BASE:327 (before the prey is erased)
            n_alive[1] -= 1;
#pragma unroll
            for (int q = 1; q < T; ++q) {
                alive[q] &= ~((q == cr) ? bit64(ck) : 0ull);
                ev[q] |= (q == cr && ln == ck) ? (uint32_t)EV_CAUGHT : 0u;
            }
            grid_zero(1, s_xy, true);                       // BASE:335
        }
    }

    // GEN2: the gain of a prey eating grass energy g (RQ:665-673)
    PPG_MEMBER double prey_after_eating(double s_e, double g) const {
        if (!GEN2) return s_e + g;                          // BASE:367
        const double raw = (C.cap_gain_grass < g) ? C.cap_gain_grass : g;
        const double ne = s_e + raw * C.eff_transfer;
        return (C.max_e_prey < ne) ? C.max_e_prey : ne;
    }

    // sel: 0 = every live prey; 1 / 2 = only type-1 / type-2 prey (GEN2 runs the engagement class by class)
    PPG_MEMBER void engage_prey(int sel = 0) {
        uint32_t pidx[T];
        uint64_t ong[T], stv[T];
        uint64_t anystv = 0;
#pragma unroll
        for (int r = 1; r < T; ++r) {
            const uint64_t mine = alive[r] & (sel == 0 ? ~0ull : (sel == 2 ? t2m[r] : ~t2m[r]));
            const uint32_t gm = ((mine >> ln) & 1ull) ? (uint32_t)chmap(3)[cell_of(xy[r])] : 0u;
            pidx[r] = gm ? (uint32_t)from_map(3, gm) : 0u;   // 0 = not standing on a patch
            ong[r] = wv::ballot(pidx[r] != 0u) & mine;
            stv[r] = wv::ballot(e[r] <= 0.0) & mine;
            anystv |= stv[r];
            if (GEN2 && ((mine >> ln) & 1ull)) ev[r] |= EV_TURN;
        }
        if (!anystv && !cooc[1]) {
            // no mid-step observation needed and one prey per cell: all eaters at once (BASE:359-372)
#pragma unroll
            for (int r = 1; r < T; ++r) {
                if ((ong[r] >> ln) & 1ull) {
                    e[r] = prey_after_eating(e[r], val[pidx[r]]);
                    if (GEN2) keep[r] &= ~(uint32_t)PPG_ROW_GRID_E0;
                    val[pidx[r]] = 0.0;
                    val[validx(r, ln)] = e[r];
                    chmap(2)[cell_of(xy[r])] = to_map(2, validx(r, ln));
                    ev[r] |= EV_ATE;
                }
                owns[r] |= ong[r];
            }
            return;
        }
#pragma unroll
        for (int r = 1; r < T; ++r) {
            uint64_t m = ong[r] | stv[r];
            while (m) {
                const int k = wv::ctz(m);
                m &= m - 1;
                const double s_e = readlane_f64(e[r], k);
                const uint32_t s_xy = wv::readlane(xy[r], k);
                if (s_e <= 0.0) { starve(r, k, s_xy); continue; }
                const uint32_t p = wv::readlane(pidx[r], k);
                wv::sync();
                const double g = first_f64(val[p]);
                const double ne = prey_after_eating(s_e, g);  // BASE:367
                e[r] = writelane_f64(e[r], k, ne);
                if (ln == k) { ev[r] |= EV_ATE; if (GEN2) keep[r] &= ~(uint32_t)PPG_ROW_GRID_E0; }  // BASE:362
                grid_set(r, k, s_xy, ne, true);             // BASE:368
                if (ln == 0) val[p] = 0.0;                  // BASE:371-372
            }
        }
    }

    // ---- step 5: reproduction (BASE:389-448, _find_available_spawn_position BASE:738-766) ----
    PPG_MEMBER bool fallback_spawn(int type, int cid, uint32_t &child_xy) {
        // BASE:759-764.  The reference draws from the unseeded global np.random; the build's
        // contract (oracle/ppg_oracle.c:find_spawn) is the k-th free cell in x-major order.
        // the occupancy board: the map of channel 0 (all-zero inside the grid); the cooperative kernels without such a map borrow bit 7
        // of the predator map's entries (8-bit maps, predator entries are <= 65) for the length of this function
        static_assert(!COOP || MAP8, "the cooperative kernels run on 8-bit maps");
        constexpr bool borrow = THREE;
        map_t *occ = borrow ? chmap(1) : chmap(0);
        const uint32_t OCC = borrow ? 0x80u : 1u;
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) {   // (two agents on one cell write the same byte value)
                map_t *at = occ + cell_of(xy[r]);
                *at = (map_t)(borrow ? ((uint32_t)*at | OCC) : OCC);
            }
        wv::sync();
        const int n = P.G * P.G;
        int nfree = 0;
        for (int base = 0; base < n; base += 64) {
            const int c = base + ln;
            nfree += wv::popc(wv::ballot(c < n && ((uint32_t)occ[cell_index(c < n ? c : 0)] & OCC) == 0u));
        }
        bool ok = false;
        if (nfree > 0) {
            uint32_t w[4];
            philox4x32_10((uint32_t)step, (uint32_t)cid, (uint32_t)type, episode, (uint32_t)seed,
                          (uint32_t)(seed >> 32) ^ TAG_SPW, w);
            int kth = (int)wv::mulhi(wv::first(w[0]), (uint32_t)nfree);
            for (int base = 0; base < n; base += 64) {
                const int c = base + ln;
                uint64_t fm = wv::ballot(c < n && ((uint32_t)occ[cell_index(c < n ? c : 0)] & OCC) == 0u);
                const int cnt = wv::popc(fm);
                if (kth < cnt) {
                    for (int s = 0; s < kth; ++s) fm &= fm - 1;
                    const int cellidx = base + wv::ctz(fm);
                    const uint32_t cx = wv::mulhi((uint32_t)cellidx, C.g_magic);
                    child_xy = (cx << 8) | ((uint32_t)cellidx - cx * (uint32_t)P.G);
                    ok = true;
                    break;
                }
                kth -= cnt;
            }
        }
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) {
                map_t *at = occ + cell_of(xy[r]);
                *at = (map_t)(borrow ? ((uint32_t)*at & ~OCC) : 0u);
            }
        wv::sync();
        return ok;
    }

    PPG_MEMBER void reproduce() {
        uint64_t cand[T];
#pragma unroll
        for (int r = 0; r < T; ++r) cand[r] = alive[r] & wv::ballot(e[r] >= (r ? C.thr_q : C.thr_p));
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int type = type_of(r);
            const int npos = type ? C.npos_prey : C.npos_pred;
            const int cap = type ? P.cap_prey : P.cap_pred;
            const double e0 = type ? C.e0_q : C.e0_p;
            uint64_t m = cand[r];
            while (m) {
                const int k = wv::ctz(m);
                m &= m - 1;
                if (next_id[type] >= npos) continue;  // id pool exhausted: no child, no reward (E6)
                if (n_rows[type] >= cap) {
                    status |= type ? PPG_STATUS_PREY_OVERFLOW : PPG_STATUS_PRED_OVERFLOW;
                    continue;
                }
                const uint32_t s_xy = wv::readlane(xy[r], k);
                const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
                uint32_t child_xy = 0;
                bool found = false;
#pragma unroll
                for (int d = 0; d < 4; ++d) {  // (x-1,y),(x+1,y),(x,y-1),(x,y+1), BASE:749
                    const int cx = x + (d == 0 ? -1 : d == 1 ? 1 : 0), cy = y + (d == 2 ? -1 : d == 3 ? 1 : 0);
                    if (found || cx < 0 || cx >= P.G || cy < 0 || cy >= P.G) continue;
                    const uint32_t c_xy = ((uint32_t)cx << 8) | (uint32_t)cy;
                    if (!any_agent_at(c_xy)) { child_xy = c_xy; found = true; }
                }
                const int cid = next_id[type];
                if (!found) {
                    status |= PPG_STATUS_FALLBACK_SPAWN;
                    fb_count += 1;
                    if (!fallback_spawn(type, cid, child_xy)) { status |= PPG_STATUS_FAILED_SPAWN; continue; }
                }
                next_id[type] += 1;                        // BASE:397/426
                const int j = n_rows[type]++;              // appended to self.agents, BASE:398/427
                const int cr = type ? 1 + (j >> 6) : 0, ck = j & 63;
                const uint32_t ckey = lexkey((uint32_t)cid);
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    if (q != cr) continue;
                    xy[q] = wv::writelane(xy[q], ck, child_xy);
                    id[q] = (int32_t)wv::writelane((uint32_t)id[q], ck, (uint32_t)cid);
                    key[q] = wv::writelane(key[q], ck, ckey);
                    e[q] = writelane_f64(e[q], ck, e0);          // BASE:403
                    if (ln == ck) ev[q] = EV_BORN;
                }
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    rows[q] |= (q == cr) ? bit64(ck) : 0ull;
                    alive[q] |= (q == cr) ? bit64(ck) : 0ull;
                }
                n_alive[type] += 1;
                grid_set(cr, ck, child_xy, e0, true);        // BASE:405
                const double ne = readlane_f64(e[r], k) - e0;  // BASE:404
                e[r] = writelane_f64(e[r], k, ne);
                if (ln == k) ev[r] |= EV_PARENT;               // reward overwrite, BASE:409/438 (E4)
                grid_set(r, k, s_xy, ne, true);                // BASE:406
                if (KICK) {
                    // kickback variant: agent_parent[child] = parent (KICK:434); the parent's own parent, if still
                    // alive, gets a bonus (KICK:443-447).  Whether that bonus lands before or after the grandparent's
                    // own reproduction in this loop decides if its reward survives (BASE:409 overwrites), so the two
                    // cases are counted separately (ev bits 8-11 / 12-15) and replayed in rewards_and_store.
                    const int my_id = (int)wv::readlane((uint32_t)id[r], k);
                    if (ln == 0) ((int32_t *)scr)[slot_of(cr, ck)] = my_id;
                    const uint32_t k_keep = wv::readlane(keep[r], k);
                    const int gp = (int)wv::first((uint32_t)C.row_parent[(size_t)b * P.S + (k_keep >> 8)]);
                    if (gp >= 0) {
#pragma unroll
                        for (int q = 0; q < T; ++q) {
                            if (type_of(q) != type) continue;
                            const uint64_t gm = wv::ballot(id[q] == gp) & alive[q] & ~wv::ballot(ev[q] & EV_BORN);
                            if (gm) {
                                const int gk = wv::ctz(gm);
                                const uint32_t gev = wv::readlane(ev[q], gk);
                                const int sh = (gev & EV_PARENT) ? 12 : 8;
                                if (((gev >> sh) & 15u) == 15u) status |= PPG_STATUS_KICK_OVERFLOW;
                                else if (ln == gk) ev[q] += 1u << sh;
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- second generation: reproduction with cooldown, chance gate and mutation (RQ:695-866) ----------
    // self.rng.random() number d of this call: the caller's stream (ppg_step_uniforms) or Philox keyed by (step, d)
    PPG_MEMBER double uniform(int d) {
        if (C.uniforms) {
            if (d >= C.uniforms_per_env) { status |= PPG_STATUS_UNIFORMS_DRY; return 0.0; }
            return first_f64(C.uniforms[(size_t)b * C.uniforms_per_env + d]);
        }
        uint32_t w[4];
        philox4x32_10((uint32_t)step, (uint32_t)d, 0u, episode, (uint32_t)seed, (uint32_t)(seed >> 32) ^ TAG_REP, w);
        return first_f64(((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) * (1.0 / 9007199254740992.0));
    }

    // The parent in row (r,k) of species SP passed the gates with enough energy (RQ:704-778 / 789-866).
    template <int SP>
    PPG_MEMBER void spawn2(int r, int k, bool mutated) {
        const uint32_t pid = id_at(r, k);
        const int pty = (int)((pid >> 16) & 1u), nty = mutated ? (pty ^ 1) : pty;   // RQ:705-712
        const int cur = nty ? next_id2[SP] : next_id[SP];
        if (cur >= C.npos2[SP * 2 + nty]) {  // RQ:715-725: no id left in that pool -- the reward is granted anyway
#pragma unroll
            for (int q = 0; q < T; ++q) ev[q] |= (q == r && ln == k) ? (uint32_t)EV_PARENT : 0u;
            return;
        }
        const int cap = SP ? P.cap_prey : P.cap_pred;
        if (n_rows[SP] >= cap) { status |= SP ? PPG_STATUS_PREY_OVERFLOW : PPG_STATUS_PRED_OVERFLOW; return; }
        const uint32_t s_xy = xy_at(r, k);
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        uint32_t child_xy = 0;
        bool found = false;
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // (x-1,y),(x+1,y),(x,y-1),(x,y+1), RQ:384-394
            const int cx = x + (d == 0 ? -1 : d == 1 ? 1 : 0), cy = y + (d == 2 ? -1 : d == 3 ? 1 : 0);
            if (found || cx < 0 || cx >= P.G || cy < 0 || cy >= P.G) continue;
            const uint32_t c_xy = ((uint32_t)cx << 8) | (uint32_t)cy;
            if (!any_agent_at(c_xy)) { child_xy = c_xy; found = true; }
        }
        if (!found) {
            status |= PPG_STATUS_FALLBACK_SPAWN;
            fb_count += 1;
            if (!fallback_spawn(SP, cur, child_xy)) { status |= PPG_STATUS_FAILED_SPAWN; return; }
        }
        const int seq = next_id[0] + next_id2[0] + next_id[1] + next_id2[1];  // agents created so far this episode
        if (nty) next_id2[SP] += 1; else next_id[SP] += 1;    // RQ:728
        const int j = n_rows[SP]++;                           // appended to self.agents, RQ:729
        const int cr = SP ? 1 + (j >> 6) : 0, ck = j & 63;
        const uint32_t cidw = ((uint32_t)seq << 17) | ((uint32_t)nty << 16) | (uint32_t)cur;
        const uint32_t ckey = (nty ? KEY_TYPE2 : 0u) + lexkey((uint32_t)cur);
        const double e0 = SP ? C.e0_q : C.e0_p;
#pragma unroll
        for (int q = 0; q < T; ++q) {
            if (type_of(q) != SP || q != cr) continue;
            xy[q] = wv::writelane(xy[q], ck, child_xy);
            id[q] = (int32_t)wv::writelane((uint32_t)id[q], ck, cidw);
            key[q] = wv::writelane(key[q], ck, ckey);
            e[q] = writelane_f64(e[q], ck, e0 * C.eff_repro);   // RQ:754-756
            if (ln == ck) { ev[q] = EV_BORN; keep[q] = PPG_ROW_GRID_E0; }
        }
#pragma unroll
        for (int q = 0; q < T; ++q) {
            rows[q] |= (q == cr) ? bit64(ck) : 0ull;
            alive[q] |= (q == cr) ? bit64(ck) : 0ull;
            t2m[q] |= (q == cr && nty) ? bit64(ck) : 0ull;
        }
        n_alive[SP] += 1;                                     // RQ:763
        grid_set(cr, ck, child_xy, e0, true);                 // RQ:760: the grid shows the full initial energy
        const double ne = e_at(r, k) - e0;                    // RQ:757
#pragma unroll
        for (int q = 0; q < T; ++q) {
            if (type_of(q) != SP) continue;
            e[q] = (q == r) ? writelane_f64(e[q], k, ne) : e[q];
            if (q == r && ln == k) { ev[q] |= EV_PARENT | EV_REPRO; keep[q] &= ~(uint32_t)PPG_ROW_GRID_E0; }  // RQ:737,767
        }
        grid_set(r, k, s_xy, ne, true);                       // RQ:761
    }

    // row_order: self.agents is still in creation order (the call right after reset): predators then prey.  Otherwise
    // it is sorted: type_1_predator*, type_1_prey*, type_2_predator*, type_2_prey* (RQ:270).
    PPG_MEMBER void reproduce2(bool row_order) {
        uint64_t elig[T], cand[T];
#pragma unroll
        for (int r = 0; r < T; ++r) {
            elig[r] = alive[r] & wv::ballot(step - lr[r] >= C.cooldown);                 // RQ:697
            cand[r] = elig[r] & wv::ballot(e[r] >= (r ? C.thr_q : C.thr_p));             // RQ:704/789
        }
        // Every eligible agent draws once (chance gate); only those with enough energy matter afterwards.  Publish
        // the candidates in self.agents order, each with the number of eligible agents in front of it.
        uint32_t *lst = (uint32_t *)scr;
        int n_cand = 0, base = 0;
        wv::sync();
#pragma unroll
        for (int sgi = 0; sgi < 4; ++sgi) {
            if (row_order && sgi >= 2) continue;
            const int species = sgi & 1, ty = sgi >> 1;
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if (type_of(r) != species) continue;
                const uint64_t segm = row_order ? ~0ull : (ty ? t2m[r] : ~t2m[r]);
                const uint64_t el = elig[r] & segm, cm = cand[r] & segm;
                if ((cm >> ln) & 1ull)
                    lst[n_cand + (int)wv::prefix(cm)] = ((uint32_t)(base + (int)wv::prefix(el)) << 16) |
                                                        ((uint32_t)species << 15) | (uint32_t)row_of(r, ln);
                n_cand += wv::popc(cm);
                base += wv::popc(el);
            }
        }
        wv::sync();
        int n_second = 0;
        for (int i = 0; i < n_cand; ++i) {
            const uint32_t w = wv::first(lst[i]);
            const int species = (int)((w >> 15) & 1u), row = (int)(w & 0x7FFFu);
            const int d1 = (int)(w >> 16) + n_second;
            if (uniform(d1) > (species ? C.chance_q : C.chance_p)) continue;             // RQ:701-702
            const double u2 = uniform(d1 + 1);                                          // RQ:708/793
            n_second += 1;
            if (species) spawn2<1>(1 + (row >> 6), row & 63, u2 < C.mut_q);
            else spawn2<0>(0, row & 63, u2 < C.mut_p);
        }
        draws = base + n_second;
    }

    // ---- rewards, cumulative rewards (BASE:288,322-323,328-329,341-344,365-366,375-378,408-411) ----
    // COOP: the table stores come AFTER the shared observation writing (coop_main calls finish_stores()).  Under a saturated store
    // pipe the ~25 store instructions of the tables take 9 k cycles to issue; in front of the workgroup barrier that is 9 k cycles in
    // which the helper waves cannot start writing (interleaved A/B of two builds: 66.4 -> 65.1 us per 4096-env step).
    bool pend = false, pend_grass = false, pend_transition = false, pend_done = false;
    PPG_MEMBER void finish_stores() {
        if (pend) { pend = false; rewards_and_store(pend_grass, pend_transition); }
    }
    PPG_MEMBER void rewards_and_store(bool write_grass, bool transition = true) {
        if (COOP && !pend_done) { pend = true; pend_done = true; pend_grass = write_grass; pend_transition = transition; return; }
        int n_new[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < T; ++r) n_new[type_of(r)] += wv::popc(wv::ballot(ev[r] & EV_BORN) & rows[r]);
        double rew_[T], cum_[T];
        uint32_t fl_[T];
        int32_t par_[T];
        const bool dense = !GEN2 && transition && C.reward_mode != 0;
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            const uint32_t v = ev[r];
            double rew = 0.0, c = 0.0;
            uint32_t fl = 0;
            if (i < n_rows[type_of(r)]) {
                // cumulative_rewards of a surviving agent still sits in HBM at the row's start-of-step slot
                // (keep[] bits 8..): read it here instead of carrying two registers per row through the step
                if (transition && !(v & EV_BORN)) c = CARRY_CUM ? cum[r] : C.row_cum[(size_t)b * P.S + (keep[r] >> 8)];
                if (!transition) {
                    rew = 0.0;  // reset returns observations only; cumulative_rewards = 0 (BASE:150)
                } else if (v & EV_BORN) {
                    c = 0.0;    // BASE:410
                } else if (v & EV_TRUNC) {
                    rew = 0.0;
                } else if (GEN2) {
                    // type-specific rewards (_get_type_specific, RQ:1099-1106); cumulative_rewards is credited where the
                    // reference credits it (RQ:596,618,643,663,691,721,769)
                    const bool t2 = (id[r] >> 16) & 1;
                    if (v & EV_STARVED) {
                        rew = 0.0;                                                       // RQ:554
                    } else if (v & EV_CAUGHT) {
                        if (v & EV_TURN) {  // it had its own turn before a predator of a later class caught it
                            const double x = (v & EV_ATE) ? (t2 ? C.r2_eat[1] : C.r2_eat[0]) : (t2 ? C.r2_qstep[1] : C.r2_qstep[0]);
                            c += x;
                            if (v & EV_ATE) c += x;
                        }
                        rew = t2 ? C.r2_caught[1] : C.r2_caught[0]; c += rew;           // RQ:616-618
                    } else {
                        if (v & EV_ATE) {
                            rew = r ? (t2 ? C.r2_eat[1] : C.r2_eat[0]) : (t2 ? C.r2_catch[1] : C.r2_catch[0]);
                            c += rew; c += rew;                                          // RQ:596+643 / 663+691
                        } else {
                            rew = r ? (t2 ? C.r2_qstep[1] : C.r2_qstep[0]) : (t2 ? C.r2_pstep[1] : C.r2_pstep[0]);
                            c += rew;                                                    // RQ:643 / 691
                        }
                        if (v & EV_PARENT) {                                             // RQ:719-721 / 767-769 overwrite
                            rew = r ? (t2 ? C.r2_repro_q[1] : C.r2_repro_q[0]) : (t2 ? C.r2_repro_p[1] : C.r2_repro_p[0]);
                            c += rew;
                        }
                    }
                } else if (dense) {
                    // dense variants: reward = energy now - energy at the start of the step (still in HBM at the
                    // row's old slot); a caught prey's account goes to zero (0.0 - before)
                    const double before = C.row_e[(size_t)b * P.S + (keep[r] >> 8)];
                    rew = (v & EV_CAUGHT) ? (0.0 - before) : (e[r] - before);
                    if (C.reward_mode == 2 && !(v & (EV_STARVED | EV_CAUGHT)))
                        rew = rew + ((v & EV_PARENT) ? (r ? C.r_repro_q : C.r_repro_p) : 0.0);
                    c += rew;
                } else if (v & EV_STARVED) {
                    rew = 0.0;
                } else if (v & EV_CAUGHT) {
                    rew = C.r_caught; c += rew;
                } else {
                    if (v & EV_ATE) { rew = r ? C.r_eat : C.r_catch; c += rew; c += rew; }
                    else { rew = r ? C.r_qstep : C.r_pstep; c += rew; }
                    if (KICK) {
                        const double kb = r ? C.kick_q : C.kick_p;
                        const uint32_t n_before = (v >> 8) & 15u, n_after = (v >> 12) & 15u;
                        for (uint32_t i = 0; i < n_before; ++i) { rew = rew + kb; c = c + kb; }   // KICK:446-447
                        if (v & EV_PARENT) { rew = r ? C.r_repro_q : C.r_repro_p; c += rew; }       // BASE:409 overwrites
                        for (uint32_t i = 0; i < n_after; ++i) { rew = rew + kb; c = c + kb; }
                    } else if (v & EV_PARENT) {
                        rew = r ? C.r_repro_q : C.r_repro_p; c += rew;
                    }
                }
                if (v & (EV_STARVED | EV_CAUGHT)) fl |= PPG_ROW_DIED;
                if ((owns[r] >> ln) & 1ull) fl |= PPG_ROW_OWNS;
                if (v & EV_BORN) fl |= PPG_ROW_NEWBORN;
                if (v & EV_ATE) fl |= PPG_ROW_ATE;
                if (v & EV_TRUNC) fl |= PPG_ROW_TRUNC | (GEN2 ? 0u : (keep[r] & PPG_ROW_ATE));  // RQ:200 clears agents_just_ate first
                if (GEN2) {
                    fl |= keep[r] & PPG_ROW_GRID_E0;
                    // agent_last_reproduction: -cooldown at registration (RQ:999), current_step at a birth (RQ:737;
                    // `step` has already been advanced when this runs)
                    if (!transition || (v & EV_BORN)) lr[r] = -C.cooldown;
                    else if (v & EV_REPRO) lr[r] = step - 1;
                }
            }
            rew_[r] = rew; cum_[r] = c; fl_[r] = fl;
            par_[r] = -1;
            if (KICK && transition && i < n_rows[type_of(r)] && !(v & (EV_STARVED | EV_CAUGHT))) {
                // agent_parent rides along with its row: newborns got it in reproduce() (LDS), survivors keep theirs
                if (v & EV_BORN) par_[r] = ((const int32_t *)scr)[slot_of(r, ln)];
                else par_[r] = C.row_parent[(size_t)b * P.S + (keep[r] >> 8)];
            }
        }
        // every lane has its start-of-step values before any row is overwritten (CARRY_CUM: nothing was read here unless the dense
        // reward modes looked up the start-of-step energies)
        if (transition && (!CARRY_CUM || dense)) wv::drain_loads();
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            if (i >= n_rows[type_of(r)]) continue;
            const size_t s = (size_t)b * P.S + slot_of(r, ln);
            C.row_xy[s] = (uint16_t)xy[r];
            C.row_e[s] = e[r];
            C.row_id[s] = id[r];
            C.row_key[s] = key[r];
            C.row_cum[s] = cum_[r];
            C.row_flags[s] = (uint8_t)fl_[r];
            C.row_reward[s] = rew_[r];
            if (KICK) C.row_parent[s] = par_[r];
            if (GEN2) C.row_lastrep[s] = lr[r];
            if (WALLS) C.row_info[s] = (uint8_t)((transition && !(ev[r] & (EV_TRUNC | EV_BORN))) ? (keep[r] & 7u) : 0u);
            keep[r] = (keep[r] & ~0xFFu) | (fl_[r] & PPG_ROW_ATE);
        }
        obs_count[0] += n_rows[0];       // every row in use got an observation
        obs_count[1] += n_rows[1];
        if (write_grass) {
            const size_t gb = (size_t)b * C.cap_grass;
            for (int p = ln; p < C.n_grass; p += 64) C.grass_e[gb + p] = val[grass_validx(p)];
        }
        int32_t *es = C.env_state + (size_t)b * PPG_ENV_WORDS;
        if (ln < PPG_ENV_WORDS) {
            int32_t w = 0;
            switch (ln) {
                case PPG_ENV_N_PRED_ROWS: w = n_rows[0]; break;
                case PPG_ENV_N_PREY_ROWS: w = n_rows[1]; break;
                case PPG_ENV_N_PRED_NEW: w = n_new[0]; break;
                case PPG_ENV_N_PREY_NEW: w = n_new[1]; break;
                case PPG_ENV_NEXT_PRED_ID: w = next_id[0]; break;
                case PPG_ENV_NEXT_PREY_ID: w = next_id[1]; break;
                case PPG_ENV_STEP: w = step; break;
                case PPG_ENV_N_PRED_ALIVE: w = n_alive[0]; break;
                case PPG_ENV_N_PREY_ALIVE: w = n_alive[1]; break;
                case PPG_ENV_FLAGS: w = (int32_t)envflags; break;
                case PPG_ENV_STATUS: w = (int32_t)status; break;
                case PPG_ENV_EPISODE: w = (int32_t)episode; break;
                case PPG_ENV_FALLBACK_SPAWNS: w = fb_count; break;
                case PPG_ENV_CALLS: w = calls; break;
                case PPG_ENV_OBS_PRED: w = obs_count[0]; break;
                case PPG_ENV_OBS_PREY: w = obs_count[1]; break;
                case PPG_ENV_NEXT_PRED_ID_T2: w = next_id2[0]; break;
                case PPG_ENV_NEXT_PREY_ID_T2: w = next_id2[1]; break;
                case PPG_ENV_DRAWS: w = draws; break;
                default: w = 0; break;
            }
            es[ln] = w;
        }
    }

    // ---- reset (BASE:129-217) with Philox Fisher-Yates placement ---------------------
    PPG_MEMBER void do_reset(uint32_t new_episode) {
        episode = new_episode;
        const int n = P.G * P.G;
        const int K = C.n_init_pred + C.n_init_prey + C.n_grass;
        // two arrays of 16-bit cell indices over the map area: the G*G cells, and the K placed entities (MAP8: the four 8-bit maps
        // together hold two arrays of map_n >= G*G entries; three maps hold G*G + K entries -- ppg_coop_layout admits only
        // configurations where they do)
        uint16_t *perm = MAP8 ? (uint16_t *)map : (uint16_t *)chmap(1);
        uint16_t *ent = THREE ? (uint16_t *)map + ((n + 7) & ~7) : MAP8 ? (uint16_t *)map + P.map_n : (uint16_t *)chmap(2);
        uint32_t *rnd = (uint32_t *)scr;  // 256 words per round
        wv::sync();
        int n_free = n;
        if (!WALLS) {
            for (int i = ln; i < n; i += 64) perm[i] = (uint16_t)i;
        } else {  // the cells that are not walls, in cell-index order (the walls stay; build contract, see oracle/rq_oracle.c)
            n_free = 0;
            for (int base = 0; base < n; base += 64) {
                const int c = base + ln;
                const bool fr = c < n && !((wallw[c >> 5] >> (c & 31)) & 1u);
                const uint64_t m = wv::ballot(fr);
                if (fr) perm[n_free + (int)wv::prefix(m)] = (uint16_t)c;
                n_free += wv::popc(m);
            }
        }
        // more entities than free cells (walls set after create; ppg_create rejects it for the open grid, BASE:167-168): flagged,
        // and only the entities that fit are placed -- the Fisher-Yates below must never index past the free cells
        const int Kp = K < n_free ? K : n_free;
        if (K > n_free) status |= PPG_STATUS_FAILED_SPAWN;
        for (int i = Kp + ln; i < K; i += 64) ent[i] = perm[0];  // (defined, never meaningful: the status bit is set)
        for (int base = 0; base < Kp; base += 256) {
            uint32_t w[4];
            philox4x32_10((uint32_t)(base >> 2) + (uint32_t)ln, 0u, 0u, episode, (uint32_t)seed,
                          (uint32_t)(seed >> 32) ^ TAG_RST, w);
            wv::sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) rnd[4 * ln + q] = w[q];
            wv::sync();
            const int hi = (Kp - base) < 256 ? (Kp - base) : 256;
            for (int kk = 0; kk < hi; ++kk) {
                const int k = base + kk;
                const uint32_t rr = wv::first(rnd[kk]);
                const int j = k + (int)wv::mulhi(rr, (uint32_t)(n_free - k));
                const uint32_t a = wv::first(perm[k]);
                const uint32_t bb = wv::first(perm[j]);
                if (ln == 0) { perm[j] = (uint16_t)a; perm[k] = (uint16_t)bb; ent[k] = (uint16_t)bb; }
            }
        }
        wv::sync();
        const int P0 = C.n_init_pred, Q0 = C.n_init_prey;
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            const int cnt = r ? Q0 : P0;
            const bool valid = i < cnt;
            xy[r] = 0xFFFFu; id[r] = 0; key[r] = 0; e[r] = 0.0; act[r] = -1; ev[r] = 0; keep[r] = 0;
            if (valid) {
                const uint32_t c = ent[(r ? P0 : 0) + i];
                const uint32_t cx = wv::mulhi(c, C.g_magic);
                xy[r] = (cx << 8) | (c - cx * (uint32_t)P.G);
                id[r] = i;
                key[r] = lexkey((uint32_t)i);
                if (GEN2) {  // RQ:125-133: type 1 first, then type 2; creation number = position in self.agents
                    const int n1 = r ? C.ninit2[2] : C.ninit2[0];
                    const int t2 = i >= n1, idx = t2 ? i - n1 : i, seq = r ? P0 + i : i;
                    id[r] = (int32_t)(((uint32_t)seq << 17) | ((uint32_t)t2 << 16) | (uint32_t)idx);
                    key[r] = (t2 ? KEY_TYPE2 : 0u) + lexkey((uint32_t)idx);
                }
                e[r] = r ? C.e0_q : C.e0_p;
            }
            rows[r] = wv::ballot(valid);
            alive[r] = rows[r];
            owns[r] = rows[r];
        }
        const size_t gb = (size_t)b * C.cap_grass;
        for (int p = ln; p < C.n_grass; p += 64) {
            const uint32_t c = ent[P0 + Q0 + p];
            const uint32_t cx = wv::mulhi(c, C.g_magic);
            const uint32_t gxy = (cx << 8) | (c - cx * (uint32_t)P.G);
            C.grass_xy[gb + p] = (uint16_t)gxy;
            C.grass_e[gb + p] = C.e0_g;
            if (p == ln) gxyr[0] = gxy;
            if (p == ln + 64) gxyr[1] = gxy;
        }
        wv::sync();
        if (COOP) init_maps();   // (the arrays lay across the padded maps and their halos)
        else for (int i = ln; i < n; i += 64) { perm[i] = 0; ent[i] = 0; }
        wv::sync();
        for (int p = ln; p < C.n_grass; p += 64) {
            // re-read what this lane just wrote (same lane, same address)
            val[grass_validx(p)] = C.e0_g;
            chmap(3)[cell_of(C.grass_xy[gb + p])] = to_map(3, grass_validx(p));
        }
        n_rows[0] = P0; n_rows[1] = Q0;
        next_id[0] = P0; next_id[1] = Q0;            // BASE:153-154
        if (GEN2) {                                  // RQ:129
            next_id[0] = C.ninit2[0]; next_id2[0] = C.ninit2[1];
            next_id[1] = C.ninit2[2]; next_id2[1] = C.ninit2[3];
#pragma unroll
            for (int r = 0; r < T; ++r) t2m[r] = wv::ballot((id[r] >> 16) & 1) & rows[r];
        }
        n_alive[0] = P0; n_alive[1] = Q0;            // BASE:210-211
        step = 0;                                    // BASE:134
        fb_count = 0;
        envflags = PPG_ENVF_WAS_RESET | PPG_ENVF_LIST_IS_ROW_ORDER;
        build_maps();
        obs_all_alive(false);                        // BASE:215
        rewards_and_store(false, false);
        obs_finish();
    }

    // ---- the transition ----------------------------------------------------------------
    // One transition: the tables were prefetched from HBM into `pre`.  `it` = index into the action tape.
    PPG_MEMBER void step_body(const Pre &pre, int it) {
        calls += 1;
        if ((C.flags & PPG_STEP_AUTO_RESET) && (envflags & PPG_ENVF_DONE)) {
            wv::sync();
            do_reset(episode + 1u);
            return;
        }
        load_rows(pre);
        if (FUSED && it > 0 && C.actions && !(C.flags & PPG_STEP_RANDOM_ACTIONS)) {  // action tape [n_steps,B,S]
#pragma unroll
            for (int r = 0; r < T; ++r)
                if ((alive[r] >> ln) & 1ull) act[r] = C.actions[((size_t)it * P.batch + b) * P.S + slot_of(r, ln)];
        }
        const bool list_is_row_order = (envflags & PPG_ENVF_LIST_IS_ROW_ORDER) != 0;

        if (step >= C.max_steps) {  // truncation, BASE:228-238: no state change
            wv::sync();
            load_grass(false, pre);
            compact_and_sort(!list_is_row_order);
            if (GEN2) after_compact();
            build_maps();
            obs_all_alive(false);
#pragma unroll
            for (int r = 0; r < T; ++r) ev[r] = ((alive[r] >> ln) & 1ull) ? EV_TRUNC : 0u;
            envflags = (envflags & PPG_ENVF_LIST_IS_ROW_ORDER) | PPG_ENVF_TRUNC_ALL | PPG_ENVF_DONE;
            rewards_and_store(false);  // (agents_just_ate is untouched by a truncation call: keep[] rides along in the flags)
            obs_finish();
            return;
        }

        PPG_STAMP(1);
        uint64_t acted[T];
        load_actions(acted);
#pragma unroll
        for (int r = 0; r < T; ++r) keep[r] &= GEN2 ? ~(uint32_t)(PPG_ROW_ATE | 7u) : ~0xFFu;  // agents_just_ate.clear(), BASE:241
        wv::sync();                                // LDS zeros visible
        PPG_STAMP(2);
        decay(acted);                              // BASE:244-250
        PPG_STAMP(3);
        load_grass(true, pre);                     // BASE:252-256
        PPG_STAMP(4);
        move(acted);                               // BASE:259-276
        PPG_STAMP(5);
        compact_and_sort(!list_is_row_order);      // BASE:222-225 + the sort of BASE:468
        PPG_STAMP(6);
        if (GEN2) after_compact();
        build_maps();
        PPG_STAMP(7);
        if (!GEN2 || list_is_row_order) {
            engage_predators();                    // BASE:302-346 (+ starvation BASE:284-301)
            wv::sync();
            PPG_STAMP(8);
            engage_prey();                         // BASE:347-380
            wv::sync();
        } else {
            // RQ:225-233 walks the sorted self.agents: type_1_predator*, type_1_prey*, type_2_predator*, type_2_prey*
            engage_predators(~t2m[0]);
            wv::sync();
            engage_prey(1);
            wv::sync();
            PPG_STAMP(8);
            engage_predators(t2m[0]);
            wv::sync();
            engage_prey(2);
            wv::sync();
        }
        PPG_STAMP(9);
        if (GEN2) reproduce2(list_is_row_order);   // RQ:248-254
        else reproduce();                          // BASE:389-448
        PPG_STAMP(10);
        obs_all_alive(false);                      // BASE:451-453
        PPG_STAMP(11);
        step += 1;                                 // BASE:471
        envflags = 0;
        if (n_alive[0] <= 0 || n_alive[1] <= 0) envflags |= PPG_ENVF_TERM_ALL | PPG_ENVF_DONE;  // BASE:466
        rewards_and_store(true);
        PPG_STAMP(12);
        obs_finish();
    }

    PPG_MEMBER void run_step(int it = 0) {
        PPG_STAMP(0);
        Pre pre;
        TabPre tab;
        if (COOP) coop_tab_issue(tab);
        prefetch(pre, true, C.actions != nullptr && !(C.flags & PPG_STEP_RANDOM_ACTIONS));
        if (COOP) coop_tab_store(tab);   // (waits for the table words only: the row loads behind them stay in flight)
        init_lds(pre);
        load_env_words(pre);
        step_body(pre, it);
    }

    PPG_MEMBER void run_reset() {
        Pre pre;
        prefetch(pre, false, false);
        load_env_words(pre);
        init_lds(pre);
        if (C.seeds) {
            uint64_t sd = C.seeds[b];
            seed = ((uint64_t)wv::first((uint32_t)(sd >> 32)) << 32) | wv::first((uint32_t)sd);
            if (ln == 0) C.env_seed[b] = seed;
        }
        status = 0;
        calls = 0;
        wv::sync();
        do_reset(C.reset_episode);
    }

    PPG_MEMBER void run_observe() {
        Pre pre;
        prefetch(pre, true, false);
        load_env_words(pre);
        init_lds(pre);
        load_rows(pre);
        wv::sync();
        load_grass(false, pre);
        build_maps();
        obs_all_alive();
    }

    // MODE_VIS: the line-of-sight masks of every cell of this env from its wall bitmap (one word = 32 window offsets per item)
    PPG_MEMBER void run_vis() {
        for (int i = ln; i < C.n_wall_words; i += 64) wallw[i] = C.wall_bits[(size_t)b * C.n_wall_words + i];
        wv::sync();
        const int n = P.G * P.G, nw = C.vis_words, wm = C.vis_w, neg = C.vis_neg;
        for (int it = ln; it < n * nw; it += 64) {
            const int cell = it / nw, w = it - cell * nw;
            const int x = (int)wv::mulhi((uint32_t)cell, C.g_magic), y = cell - x * P.G;
            uint32_t word = 0;
            for (int k = 0; k < 32; ++k) {
                const int bi = 32 * w + k;
                if (bi >= wm * wm) break;
                const int ci = bi / wm, cj = bi - ci * wm;
                const int gx = x - neg + ci, gy = y - neg + cj;
                if ((unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G && los_clear(x, y, gx, gy)) word |= 1u << k;
            }
            C.vis_masks[((size_t)b * n + cell) * nw + w] = word;
        }
    }

    PPG_MEMBER void run_export_grid() {
        Pre pre;
        prefetch(pre, true, false);
        load_env_words(pre);
        init_lds(pre);
        load_rows(pre);
        wv::sync();
        load_grass(false, pre);
        build_maps();
        const int n = P.G * P.G;
        double *out = C.grid_out + (size_t)b * 4 * n;
        for (int i = ln; i < 4 * n; i += 64) {
            const int ch = i / n, c = i - ch * n;
            out[i] = ch ? val[from_map(ch, chmap(ch)[c])] : ((WALLS && ((wallw[c >> 5] >> (c & 31)) & 1u)) ? 1.0 : 0.0);  // WO:271-273
        }
    }
};

template <int NQ, int MODE, bool FASTOBS, bool GEN2 = false, bool WALLS = false, bool DRIVE = false, int NW = 1>
PPG_DEVICE void env_main(const KParams &P, unsigned char *lds) {
    int b = PPG_BLOCK_INDEX();
    if (b >= P.batch) return;
    {   // scheduling only: heavy envs first, so that consecutive workgroups (which land on different CUs) spread the load
        const PPG_CONSTANT_AS KParams *Pk = PPG_KERNARG_PTR(KParams, P);
        if (Pk->env_order) b = (int)wv::first((uint32_t)Pk->env_order[b]);   // (kept scalar: b feeds every address)
    }
    if (MODE == MODE_ROLLOUT) {
        // ppg_rollout: n_steps transitions in one launch.  Every lane always loads and stores the same slots
        // (its rows, its env words, its grass patches), so a fused step reads its state back through memory
        // (L2-hot, same-lane read-after-write); nothing is carried in registers, there is no launch gap, and the
        // observation stores of step t drain while step t+1 computes.  The whole per-env context is rebuilt
        // every iteration from laundered roots so that hipcc cannot hoist loop-invariant values out of the
        // step loop (that hoisting is what made earlier formulations spill hundreds of registers).
        // STATUS (round 1): bit-identical to n single steps and spill-free, but each fused step is still slower
        // than a ppg_step launch (parameters are scalar loads at their use sites here); bench.py does not use it.
        const int n = P.n_steps;
        for (int it = 0; it < n; ++it) {
            const PPG_CONSTANT_AS KParams *Pc = PPG_KERNARG_PTR(KParams, P);
            int bb = b, lane = wv::lane();
            unsigned char *l = lds;
            PPG_LAUNDER_S(Pc);
            PPG_LAUNDER_S(bb);
            PPG_LAUNDER_V(lane);
            PPG_LAUNDER_V(l);
            Env<NQ, false, FASTOBS, true, false, false, false, false, 1, const PPG_CONSTANT_AS KParams, const PPG_CONSTANT_AS KParams> env(*Pc, *Pc, bb, l, lane);
            env.run_step(it);
            wv::sync();
        }
        return;
    }
    const PPG_CONSTANT_AS KParams *Pcold = PPG_KERNARG_PTR(KParams, P);  // KParams is the kernel's only argument
    Env<NQ, MODE == MODE_STEP_ORDERED || MODE == MODE_STEP_ORDERED_KICK, FASTOBS, false,
        MODE == MODE_STEP_KICK || MODE == MODE_STEP_ORDERED_KICK, GEN2, WALLS, DRIVE, NW, const KParams, const PPG_CONSTANT_AS KParams>
        env(P, *Pcold, b, lds, wv::lane());
    if (NW > 1) {
        const int w = wv::wave_index();
        env.wave_idx = w;
        if (decltype(env)::ADAPTIVE_HELPERS) {   // light envs are left to wave 0 alone: the launch is as slow as its slowest env, and that is a HEAVY one
            const PPG_CONSTANT_AS KParams *Pk = PPG_KERNARG_PTR(KParams, P);
            const int32_t *es = Pk->env_state + (size_t)b * PPG_ENV_WORDS;
            const int rows0 = (int)wv::first((uint32_t)(es[PPG_ENV_N_PRED_ROWS] + es[PPG_ENV_N_PREY_ROWS]));
            env.helpers = rows0 >= Pk->helper_min_rows;
            // every wave has read the two words before wave 0 can get to overwrite them at the end of its step: all waves of the
            // workgroup are still here, so this barrier costs nothing, and all of them take the same decision
            wv::wg_barrier();
        }
        if (w != 0) {
            if (env.helpers) env.run_helper(w);
            return;
        }
    }
    if (MODE == MODE_STEP || MODE == MODE_STEP_ORDERED || MODE == MODE_STEP_KICK || MODE == MODE_STEP_ORDERED_KICK) env.run_step();
    else if (MODE == MODE_RESET) env.run_reset();
    else if (MODE == MODE_OBSERVE) env.run_observe();
    else if (MODE == MODE_VIS) env.run_vis();
    else env.run_export_grid();
}

// The cooperative step kernels (ppgc_*): see Env's COOP.  Workgroup g steps envs g * coop_e ... g * coop_e + coop_e - 1.
template <int NQ, bool GEN2, int NW, bool CH0MAP = true>
PPG_DEVICE void coop_main(const KParams &P, unsigned char *lds) {
    const PPG_CONSTANT_AS KParams *Pc = PPG_KERNARG_PTR(KParams, P);
    typedef Env<NQ, false, false, false, false, GEN2, false, false, NW, const KParams, const PPG_CONSTANT_AS KParams, true, CH0MAP> CoopEnv;
    const int w = wv::wave_index(), ln = wv::lane();
    const int ne = Pc->coop_e;
    uint32_t *lut2 = (uint32_t *)(lds + Pc->off_lut2), *ctl = (uint32_t *)(lds + Pc->off_ctl);
    int b = PPG_BLOCK_INDEX() * ne + w;
    const bool has_env = w < ne && b < P.batch;
    if (has_env && Pc->env_order) b = (int)wv::first((uint32_t)Pc->env_order[b]);   // scheduling only (ppg_rebalance)
    CoopEnv env(P, *Pc, has_env ? b : 0, lds + (size_t)(w < ne ? w : 0) * (size_t)Pc->lds_env_bytes, ln);
    env.wave_idx = w;
    env.lut2 = lut2;
    env.ctl = ctl;
    // (every wavefront that steps an env writes the whole descriptor table, identical words: it may need it for a mid-step
    // observation long before the workgroup's barrier; a workgroup always has at least one env)
    if (has_env) env.run_step();
    else if (w < ne && ln == 0) ctl[CoopEnv::CTL_SLOT + 4 * w + 2] = 0xFFFFFFFFu;   // an env slot beyond the batch
#ifdef PPG_PROFILE_PHASES
#define PPG_COOP_STAMP(i) do { if (Pc->prof && has_env) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); if (ln == 0) Pc->prof[(size_t)b * 16 + (i)] = t_; } } while (0)
#else
#define PPG_COOP_STAMP(i) do { } while (0)
#endif
    wv::wg_barrier_lds();   // (LDS only: the table stores of this wave need not have reached memory)
    PPG_COOP_STAMP(13);
    env.coop_write_all(lds);
    PPG_COOP_STAMP(14);
    if (has_env) env.finish_stores();
}

// ppg_rollout on a handle whose plan is cooperative: P.n_steps transitions in one launch.  A workgroup's envs never interact with any
// other workgroup's, so the workgroups of a launch simply run on, each at its own pace -- no launch boundary, and nothing that makes
// all of them compute (and none of them store) at the same time.  The per-env context is rebuilt every iteration from laundered
// roots (see env_main's MODE_ROLLOUT: left alone, hipcc hoists loop-invariant values out of the step loop and spills them).
template <int NQ, bool GEN2, int NW>
PPG_DEVICE void coop_main_fused(const KParams &P, unsigned char *lds) {
    typedef Env<NQ, false, false, true, false, GEN2, false, false, NW, const KParams, const PPG_CONSTANT_AS KParams, true> CoopEnv;
    const PPG_CONSTANT_AS KParams *Pk = PPG_KERNARG_PTR(KParams, P);
    const int w = wv::wave_index();
    const int ne = Pk->coop_e;
    int b = PPG_BLOCK_INDEX() * ne + w;
    const bool has_env = w < ne && b < P.batch;
    if (has_env && Pk->env_order) b = (int)wv::first((uint32_t)Pk->env_order[b]);
    const int n_it = Pk->n_steps;
    for (int it = 0; it < n_it; ++it) {
        const PPG_CONSTANT_AS KParams *Pc = PPG_KERNARG_PTR(KParams, P);
        int bb = b, lane = wv::lane();
        unsigned char *const l = lds;
        PPG_LAUNDER_S(Pc);
        PPG_LAUNDER_S(bb);
        PPG_LAUNDER_V(lane);
        // (NOT the LDS base: behind an opaque 64-bit pointer every LDS access becomes a flat instruction, and in these multi-wave
        // workgroups that faulted on the device -- HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION -- while the emulator was happy.  With
        // the three roots above laundered the loop already stays at 100 VGPRs without scratch; with none, 128 + 52 B.)
        CoopEnv env(P, *Pc, has_env ? bb : 0, l + (size_t)(w < ne ? w : 0) * (size_t)Pc->lds_env_bytes, lane);
        env.wave_idx = w;
        env.lut2 = (uint32_t *)(l + Pc->off_lut2);
        env.ctl = (uint32_t *)(l + Pc->off_ctl);
        if (has_env) env.run_step(it);
        else if (w < ne && lane == 0) env.ctl[CoopEnv::CTL_SLOT + 4 * w + 2] = 0xFFFFFFFFu;
        wv::wg_barrier_lds();
        env.coop_write_all(l);
        if (has_env) env.finish_stores();
        if (it + 1 < n_it) {
            wv::drain_loads();       // this wave's table stores have been performed before it loads the same slots back
            wv::wg_barrier_lds();    // every wave is done with the env regions before the next transitions rebuild them
        }
    }
}

}  // namespace ppg
