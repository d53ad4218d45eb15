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
    int32_t vis_env_stride;       // cells between the mask tables of env b and env b + 1: G*G, or 0 when ppg_walls_changed found every env's
                                  // wall bitmap equal to env 0's (the usual case: one wall layout for the batch) -- all envs then read
                                  // env 0's table, a few KB that stay in L2 instead of a scattered read of [B, G*G, vis_words] from HBM
    // drive-conditioned variant of the base family (drive_conditioned_environment/predpreygrass_rllib_env.py, "DRV")
    int32_t n_drive[2];           // extra constant-filled observation channels per species (DRV:70-75), <= 4
    int32_t drive_kind[2][4];     // 0 hunger_pressure, 1 reproductive_readiness, 2 prey_opportunity, 3 predator_danger_pressure,
                                  // 4 grass_opportunity (DRV:587-608)
    int32_t off_win;              // LDS offset of the window staging area (one float64 per window cell)
    int32_t off_vm;               // walls variant: LDS offset of the listed rows' line-of-sight masks, (64 + cap_prey) x vis_words words
                                  // (entry i of the predator list at i, of the prey list at 64 + i: Env::walls_stage_masks)
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

struct TagTrue { static constexpr bool value = true; };    // compile-time switches handed to generic lambdas
struct TagFalse { static constexpr bool value = false; };

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
    // CTL_READY: bit k = env slot k's transition is over, its rows are published; CTL_TICKET + 2k + species: the next piece of that
    // run nobody has taken yet (DYN)
    enum { CTL_SLOT = 0, CTL_MID = 64, CTL_READY = 80, CTL_TICKET = 84, CTL_WORDS = 128 };
    // DYN: the shared write phase without a workgroup barrier.  A wavefront whose transition is over sets its env's READY bit and
    // every wavefront of the workgroup takes pieces of the envs that are ready through LDS tickets: while one of the workgroup's
    // transitions is still running, the other env's observations are already being written by the three wavefronts that are free
    // (behind a barrier they waited -- 5-7 % of a wavefront's cycles on average, far more in the slow tail).
#ifndef PPG_COOP_DYNAMIC
#define PPG_COOP_DYNAMIC 1
#endif
// the table stores of a transition wavefront are issued right after its READY bit, before it joins the writing (they need registers
// only; with no barrier behind them they delay nobody else and complete under the write phase instead of at the workgroup's tail:
// second generation +3.3 %, walls +2.8 %, 64x64 grids and headline +-0.5 % -- profiles/r06/y_*)
#ifndef PPG_COOP_STORES_FIRST
#define PPG_COOP_STORES_FIRST 1
#endif
    static constexpr bool DYN = PPG_COOP_DYNAMIC && COOP && !FUSED;

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
    PPG_MEMBER bool wall_at(int x, int y) const { return wall_in(wallw, x, y); }
    PPG_MEMBER bool wall_in(const uint32_t *ww, int x, int y) const {   // (ww: the bitmap of an env's LDS region)
        const int c = x * P.G + y;
        return (ww[c >> 5] >> (c & 31)) & 1u;
    }
    // _line_of_sight_clear (WO:492-525) == the bresenham walk of _get_observation (WO:550-589): no wall strictly between
    // the two cells.  The reference's float error term dx/2.0 is carried doubled, as an integer.
    PPG_MEMBER bool los_clear(int x0, int y0, int x1, int y1) const { return los_clear_in(wallw, x0, y0, x1, y1); }
    PPG_MEMBER bool los_clear_in(const uint32_t *ww, int x0, int y0, int x1, int y1) const {
        const int dx = x1 > x0 ? x1 - x0 : x0 - x1, dy = y1 > y0 ? y1 - y0 : y0 - y1;
        const int sx = x1 > x0 ? 1 : -1, sy = y1 > y0 ? 1 : -1;
        int x = x0, y = y0;
        bool clear = true;
        if (dx >= dy) {
            int err2 = dx;
            while (x != x1) {
                if (!(x == x0 && y == y0) && !(x == x1 && y == y1) && wall_in(ww, x, y)) clear = false;
                err2 -= 2 * dy;
                if (err2 < 0) { y += sy; err2 += 2 * dx; }
                x += sx;
            }
        } else {
            int err2 = dy;
            while (y != y1) {
                if (!(x == x0 && y == y0) && !(x == x1 && y == y1) && wall_in(ww, x, y)) clear = false;
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

    // ---- the phases: member functions, one file per phase (round 6: a textual split -- the generated code is token-identical) ----
#include "ppg_env_load.h"       // prefetch, rows, LDS initialisation, grass
#include "ppg_env_move.h"       // actions, decay, movement in action order
#include "ppg_env_sort.h"       // dead rows out, newborns into sorted position
#include "ppg_env_observe.h"    // cell maps + observation rows (base / drive / walls), shared row lists
#include "ppg_env_coop.h"       // cooperative kernels: observations as 1 KB pieces
#include "ppg_env_engage.h"     // starvation, predator and prey engagement
#include "ppg_env_reproduce.h"  // reproduction, spawn fallback, second-generation gates
#include "ppg_env_step.h"       // rewards + stores, device reset, step_body, run_*
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
// WALLS (ppgc3_step): the walls variant of the second generation -- three cell maps (its channel 0 is the wall bitmap), rows written whole.
template <int NQ, bool GEN2, int NW, bool CH0MAP = true, bool WALLS = false>
PPG_DEVICE void coop_main(const KParams &P, unsigned char *lds) {
    static_assert(!WALLS || (GEN2 && !CH0MAP), "the cooperative walls kernel: second generation, three maps");
    const PPG_CONSTANT_AS KParams *Pc = PPG_KERNARG_PTR(KParams, P);
    typedef Env<NQ, false, false, false, false, GEN2, WALLS, false, NW, const KParams, const PPG_CONSTANT_AS KParams, true, CH0MAP> CoopEnv;
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
    if (CoopEnv::DYN) {   // READY bits and tickets start at zero (all wavefronts are still here: this barrier costs nothing)
        if (w == NW - 1 && ln < CoopEnv::CTL_WORDS - CoopEnv::CTL_READY) ctl[CoopEnv::CTL_READY + ln] = 0u;
        wv::wg_barrier_lds();
    }
    if (has_env) env.run_step();
    else if (w < ne && ln == 0) ctl[CoopEnv::CTL_SLOT + 4 * w + 2] = 0xFFFFFFFFu;   // an env slot beyond the batch
#ifdef PPG_PROFILE_PHASES
#define PPG_COOP_STAMP(i) do { if (Pc->prof && has_env) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); if (ln == 0) Pc->prof[(size_t)b * 16 + (i)] = t_; } } while (0)
#else
#define PPG_COOP_STAMP(i) do { } while (0)
#endif
    if (CoopEnv::DYN) {
        if (w < ne) wv::lds_or(ctl + CoopEnv::CTL_READY, 1u << w);
        PPG_COOP_STAMP(13);
#if PPG_COOP_STORES_FIRST
        if (has_env) env.finish_stores();
#endif
        env.coop_write_dynamic(lds);
    } else {
        wv::wg_barrier_lds();   // (LDS only: the table stores of this wave need not have reached memory)
        PPG_COOP_STAMP(13);
        env.coop_write_all(lds);
    }
    PPG_COOP_STAMP(14);
    if (has_env) env.finish_stores();
}

template <int NQ>
PPG_DEVICE void coop_walls_main(const KParams &P, unsigned char *lds) {
    if constexpr (NQ <= 2) coop_main<NQ, true, 4, false, true>(P, lds);   // (8-bit maps: up to 128 prey rows)
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
