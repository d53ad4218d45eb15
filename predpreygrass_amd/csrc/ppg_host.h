// ppg_host.h -- host side of the C ABI in include/ppg.h that does not depend on how the
// kernels are launched.  The including translation unit provides three functions:
//   static int  backend_init(ppg_handle *h, int device);            // device check + upload h->lut_host
//   static void backend_release(ppg_handle *h);
//   static int  backend_launch(ppg_handle *h, int mode, const ppg::KParams &P, void *stream);
// predpreygrass_amd/csrc/ppg_hip.hip is the product backend (hipLaunchKernelGGL on gfx950);
// tests/wave_emu/ppg_emu.cpp is a test-only backend that runs the same kernel source on the
// CPU under a lockstep wave emulator.
#pragma once

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "ppg_kernel.h"
#include "ppg_pack.h"
#include "ppg_fetch.h"

// How a handle's step is scheduled (never what it computes): wavefronts per workgroup, the row count from which helper wavefronts
// stay, and -- cooperative kernels -- how many envs share a workgroup (0 = one env per workgroup, the ppg[w]_step kernels).
struct ppg_wave_plan_t { int nw; int min_rows; int coop_e; };

struct ppg_handle {
    ppg_wave_plan_t plan;     // what ppg_step launches: decided at ppg_create / ppg_set_envs_in_flight / ppg_set_wave_plan
    ppg_wave_plan_t forced;   // ppg_set_wave_plan: nw = 0 -> automatic
    ppg::KParams coop;        // parameter block of the cooperative step kernels (their own LDS layout: padded cell maps, four per env)
    int32_t coop_ok;          // the configuration has cooperative kernels (ppg_coop_layout)
    ppg::KParams coop3;       // the same with THREE cell maps per env (KParams::ch0_map 0: kernels ppgcm_*, four-wave step launches only)
    int32_t coop3_ok;         // that layout exists for the configuration ...
    int32_t coop_prefers3;    // ... and is the one a four-wave cooperative step uses (large grids; PPG_COOP_MAPS)
    uint32_t coop3_tab_off;   // words in front of its descriptor table inside coop_tab
    std::vector<uint32_t> coop_tab_host;
    uint32_t *coop_tab_dev;   // library-owned: KParams::coop_tab of both layouts
    int32_t drive;  // drive-conditioned variant of the base family (cfg.n_drive)
    int32_t envs_in_flight;  // scheduling hint (ppg_set_envs_in_flight); 0 = the handle's own batch
    int32_t coop_wgs_per_cu;  // float64 / float32 rows: four-wave cooperative workgroups a CU takes at most (PPG_COOP_WGS_PER_CU, read at create)
    int32_t step_lds_pad;     // experiment (PPG_STEP_LDS_PAD, read at create): unused LDS added to the multi-wave kernels' launches
    int32_t *order_dev;      // library-owned [batch]: env order of ppg_rebalance (NULL until first used)
    uint32_t *vis_dev;       // library-owned [batch, G*G, vis_words]: line-of-sight masks of the walls variant (ppg_walls_changed)
    unsigned char *fetch_dev;   // library-owned staging buffer of ppg_fetch (NULL until first used)
    uint64_t fetch_cap;         // its size
    uint64_t fetch_hint;        // bytes the next ppg_fetch copies in its first (usually only) transfer
    ppg_config cfg;
    ppg_config_gen2 cfg2;
    int32_t gen2;  // created by ppg_create_gen2
    ppg_buffers bufs;
    int32_t batch;
    int32_t device;
    int32_t nq;  // prey row registers
    ppg::KParams base;
    std::vector<uint32_t> lut_host;
    uint32_t *lut_dev;
    void *backend;
    unsigned long long *prof_dev;  // diagnostic build only
    char kernel_name[48];          // ppg_step_kernel_name
    char err[256];
};

static char g_ppg_create_error[256] = "";

static int ppg_fail(ppg_handle *h, int code, const char *fmt, ...) {
    char *dst = h ? h->err : g_ppg_create_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}

static uint32_t ppg_host_lexkey(uint32_t id) {
    char s[16];
    snprintf(s, sizeof s, "%u", id);
    static const uint32_t pw[6] = {161051u, 14641u, 1331u, 121u, 11u, 1u};
    uint32_t key = 0;
    for (int i = 0; s[i] && i < 6; ++i) key += (uint32_t)(s[i] - '0' + 1) * pw[i];
    return key;
}

// Observation element descriptors (see Env::obs_row): for chunk ch, lane l the two words describe
// elements e = ch*128 + 2l and e+1 of the (4,R,R) block in C order.
static int ppg_obs_chunks(int R) { return (4 * R * R + 127) / 128; }

static int ppg_obs_chunks_c(int R, int channels) { return (channels * R * R + 127) / 128; }

// channels = 4; 5 in the walls variant with the visibility channel (its elements: bit 28 set, channel bits 0); 4 + n_drive
// in the drive-conditioned variant (drive elements: bit 29 set, bits 24-25 = index of the drive channel)
static void ppg_build_lut(int R, int G, int map_n, uint32_t *out, int channels = 4, bool drive = false) {
    const int blk = channels * R * R, off = (R - 1) / 2;
    const int W = 2 * off + 1;  // BASE:532-539: the window is x-off..x+off
    const int nwords = ppg_obs_chunks_c(R, channels) * 128;
    // odd-sized blocks (an odd number of channels x an odd window) cannot be stored as aligned element pairs: their
    // descriptors are laid out so that lane l of chunk c handles elements c*128 + l and c*128 + 64 + l (see Env::obs_row)
    const bool strided = (blk & 1) != 0;
    for (int slot = 0; slot < nwords; ++slot) {
        const int e = strided ? (slot & ~127) + ((slot & 127) >> 1) + 64 * (slot & 1) : slot;
        uint32_t d = 0;
        if (e < blk) {
            const int c = e / (R * R), rem = e % (R * R), i = rem / R, j = rem % R;
            const int cm = c < 4 ? c : 0;
            const int moff = cm * map_n + (i - off) * G + (j - off);
            d = ((uint32_t)moff & 0xFFFFu) | ((uint32_t)(i - off + 8) << 16) | ((uint32_t)(j - off + 8) << 20) |
                ((uint32_t)cm << 24) | 0x4000000u | ((i < W && j < W) ? 0x8000000u : 0u) |
                ((c >= 4 && !drive) ? 0x10000000u : 0u) |
                ((uint32_t)(cm >= 2 ? cm - 1 : 0) << 30);   // bits 30-31: value-table section of the channel (8-bit maps, Env::map_base)
            if (c >= 4 && drive) d = 0x4000000u | 0x20000000u | ((uint32_t)(c - 4) << 24);
        }
        out[slot] = d;
    }
}

static int ppg_validate_and_layout(ppg_handle *h) {
    const ppg_config &c = h->cfg;
    if (c.abi_version != PPG_ABI_VERSION) return ppg_fail(h, PPG_EINVAL, "abi_version %d != %d", c.abi_version, PPG_ABI_VERSION);
    if (h->batch < 1) return ppg_fail(h, PPG_EINVAL, "batch must be >= 1");
    if (c.grid_size < 2 || c.grid_size > 88) return ppg_fail(h, PPG_EINVAL, "grid_size %d outside 2..88", c.grid_size);
    if (c.predator_obs_range < 1 || c.predator_obs_range > 15 || c.prey_obs_range < 1 || c.prey_obs_range > 15)
        return ppg_fail(h, PPG_EINVAL, "obs ranges must be in 1..15");
    if (c.pred_capacity != 64) return ppg_fail(h, PPG_EINVAL, "pred_capacity must be 64");
    if (c.prey_capacity != 64 && c.prey_capacity != 128 && c.prey_capacity != 256)
        return ppg_fail(h, PPG_EINVAL, "prey_capacity must be 64, 128 or 256");
    if (c.n_grass < 0 || c.grass_capacity < c.n_grass || c.grass_capacity % 64 != 0 || c.grass_capacity > 4096)
        return ppg_fail(h, PPG_EINVAL, "grass_capacity must be a multiple of 64, >= n_grass, <= 4096");
    if (c.n_initial_predators < 0 || c.n_initial_predators > c.pred_capacity || c.n_initial_prey < 0 ||
        c.n_initial_prey > c.prey_capacity)
        return ppg_fail(h, PPG_EINVAL, "initial agent counts exceed the row capacities");
    if (c.n_initial_predators + c.n_initial_prey + c.n_grass > c.grid_size * c.grid_size)
        return ppg_fail(h, PPG_EINVAL, "Cannot place more unique positions than grid cells.");  // BASE:167-168
    if (c.n_possible_predators < 0 || c.n_possible_predators > 999999 || c.n_possible_prey < 0 || c.n_possible_prey > 999999)
        return ppg_fail(h, PPG_EINVAL, "n_possible_* must be in 0..999999");
    if (c.obs_dtype < 0 || c.obs_dtype > 2) return ppg_fail(h, PPG_EINVAL, "obs_dtype must be 0 (f64), 1 (f32) or 2 (bf16)");
    if (c.obs_dtype == 2 && ((h->gen2 && h->cfg2.walls) || c.n_drive[0] > 0 || c.n_drive[1] > 0))
        return ppg_fail(h, PPG_EINVAL, "bfloat16 observations (obs_dtype 2) are for the 4-channel observations of the base family and the second generation");
    if (c.max_steps < 0) return ppg_fail(h, PPG_EINVAL, "max_steps < 0");
    const ppg_buffers &b = h->bufs;
    if (!b.row_xy || !b.row_energy || !b.row_id || !b.row_key || !b.row_cumrew || !b.row_flags || !b.row_reward ||
        !b.env_state || !b.env_seed || !b.grass_xy || !b.grass_energy || !b.obs_pred || !b.obs_prey || !b.row_parent)
        return ppg_fail(h, PPG_EINVAL, "a buffer pointer is NULL");

    ppg::KParams &P = h->base;
    memset(&P, 0, sizeof P);
    P.G = c.grid_size; P.Rp = c.predator_obs_range; P.Rq = c.prey_obs_range; P.max_steps = c.max_steps;
    P.npos_pred = c.n_possible_predators; P.npos_prey = c.n_possible_prey;
    P.n_init_pred = c.n_initial_predators; P.n_init_prey = c.n_initial_prey; P.n_grass = c.n_grass;
    P.cap_pred = c.pred_capacity; P.cap_prey = c.prey_capacity; P.cap_grass = c.grass_capacity;
    P.S = c.pred_capacity + c.prey_capacity;
    P.obs_f32 = c.obs_dtype;
    P.g_magic = (uint32_t)((0x100000000ull + (uint64_t)c.grid_size - 1) / (uint64_t)c.grid_size);
    P.rp_magic = (uint32_t)((0x100000000ull + (uint64_t)c.predator_obs_range - 1) / (uint64_t)c.predator_obs_range);
    P.rq_magic = (uint32_t)((0x100000000ull + (uint64_t)c.prey_obs_range - 1) / (uint64_t)c.prey_obs_range);
    {
        const uint64_t np = (uint64_t)c.predator_obs_range * c.predator_obs_range, nq = (uint64_t)c.prey_obs_range * c.prey_obs_range;
        P.np_magic = (uint32_t)((0x100000000ull + np - 1) / np);
        P.nq_magic = (uint32_t)((0x100000000ull + nq - 1) / nq);
    }
    P.r_catch = c.reward_predator_catch_prey; P.r_eat = c.reward_prey_eat_grass;
    P.r_pstep = c.reward_predator_step; P.r_qstep = c.reward_prey_step; P.r_caught = c.penalty_prey_caught;
    P.r_repro_p = c.reproduction_reward_predator; P.r_repro_q = c.reproduction_reward_prey;
    P.loss_p = c.energy_loss_per_step_predator; P.loss_q = c.energy_loss_per_step_prey;
    P.thr_p = c.predator_creation_energy_threshold; P.thr_q = c.prey_creation_energy_threshold;
    P.e0_p = c.initial_energy_predator; P.e0_q = c.initial_energy_prey; P.e0_g = c.initial_energy_grass;
    P.gain_g = c.energy_gain_per_step_grass; P.cap_g = c.initial_energy_grass;
    if (c.reward_mode < 0 || c.reward_mode > 2) return ppg_fail(h, PPG_EINVAL, "reward_mode must be 0, 1 or 2");
    P.reward_mode = c.reward_mode;
    if (c.kickback && c.reward_mode != 0) return ppg_fail(h, PPG_EINVAL, "kickback requires reward_mode 0");
    P.kickback = c.kickback ? 1 : 0; P.kick_p = c.kickback_reward_predator; P.kick_q = c.kickback_reward_prey;
    P.season_len = c.season_length_steps; P.season_hi = c.season_high_multiplier; P.season_lo = c.season_low_multiplier;
    h->nq = c.prey_capacity / 64;

    // LDS layout, every region 16-byte aligned
    const int n = c.grid_size * c.grid_size;
    P.map_n = (n + 7) / 8 * 8;
    int off = 0;
    // cell maps: 8-bit channel-local indices for up to 128 prey rows (Env::MAP8), else 16-bit indices into the value table
    const int map_elem = h->nq <= 2 ? 1 : 2;
    if (map_elem == 1 && c.n_grass > 255)
        return ppg_fail(h, PPG_EINVAL, "more than 255 grass patches need prey_capacity 256 (16-bit cell maps)");
    P.off_map = off; off += 4 * P.map_n * map_elem;
    off = (off + 15) / 16 * 16;
    // value table; 8-bit maps: three zero-led sections at 0 / 129 / 258, the last one as long as there are patches (Env::grass_validx)
    P.off_val = off; off += (map_elem == 1 ? 259 + c.n_grass : 1 + P.S + P.cap_grass) * 8;
    off = (off + 15) / 16 * 16;
    // scratch: 8 bytes per row of ONE species at a time (compaction), int32 per slot (parents), a list of 1 + S words (rows to
    // observe), 256 random words (reset).  (64x64 grid, 7x7 windows: with the shorter value table 20280 bytes per env = 8 per CU.)
    const int cap_max = P.cap_pred > P.cap_prey ? P.cap_pred : P.cap_prey;
    int scr_bytes = cap_max * 8 > (1 + P.S) * 4 ? cap_max * 8 : (1 + P.S) * 4;
    if (scr_bytes < 1024) scr_bytes = 1024;
    P.off_scr = off; off += scr_bytes;
    const bool drive = !h->gen2 && (c.n_drive[0] > 0 || c.n_drive[1] > 0);
    if (!h->gen2) {
        for (int t = 0; t < 2; ++t) {
            if (c.n_drive[t] < 0 || c.n_drive[t] > 4) return ppg_fail(h, PPG_EINVAL, "n_drive must be in 0..4");
            for (int k = 0; k < c.n_drive[t]; ++k)
                if (c.drive_kind[t][k] < 0 || c.drive_kind[t][k] > 4) return ppg_fail(h, PPG_EINVAL, "unknown drive feature %d", c.drive_kind[t][k]);
        }
        if (drive && c.kickback) return ppg_fail(h, PPG_EINVAL, "drive channels cannot be combined with the kickback variant");
    }
    const int channels = (h->gen2 && h->cfg2.walls && h->cfg2.include_visibility_channel) ? 5 : 4;
    const int ch_p = drive ? 4 + c.n_drive[0] : channels, ch_q = drive ? 4 + c.n_drive[1] : channels;
    P.blk_p = ch_p * P.Rp * P.Rp; P.blk_q = ch_q * P.Rq * P.Rq;   // elements per observation block
    P.nch_p = (P.blk_p + 127) / 128; P.nch_q = (P.blk_q + 127) / 128;
    // the descriptor table lives in LDS only for the kernels that read it from there: the FASTOBS kernels (base family / second
    // generation with <= 2 predator and <= 3 prey chunks: the same rule as the backends' kernel selection) keep it in registers.
    // (64x64 grid, 7x7 windows: 24080 -> 22032 bytes per env = 7 instead of 6 envs per CU.)
    const bool lut_in_registers = P.nch_p <= 2 && P.nch_q <= 3 && !drive && !(h->gen2 && h->cfg2.walls)
#ifdef PPG_WAVE_EMU   // (tests: exercise the LDS-descriptor observation path on configurations that would keep it in registers)
                                  && !getenv("PPG_EMU_FORCE_GENERIC_OBS")
#endif
        ;
    P.off_lut = off; off += lut_in_registers ? 0 : (P.nch_p + P.nch_q) * 128 * 4;
    if (drive) {  // staging area for the window sums of the drive features
        for (int t = 0; t < 2; ++t) {
            P.n_drive[t] = c.n_drive[t]; P.hunger_safe[t] = c.hunger_safe_energy[t];
            for (int k = 0; k < 4; ++k) P.drive_kind[t][k] = c.drive_kind[t][k];
        }
        P.norm_prey_opp = c.prey_opportunity_normalizer; P.norm_pred_danger = c.predator_danger_normalizer;
        P.norm_grass_opp = c.grass_opportunity_normalizer;
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;
        off = (off + 15) / 16 * 16;
        P.off_win = off; off += 4 * 4 * rmax * rmax * 8;   // four channel planes per wave of a multi-wave workgroup
    }
    if (h->gen2 && h->cfg2.walls) {  // wall bitmap
        P.n_wall_words = (n + 31) / 32;
        off = (off + 15) / 16 * 16;
        P.off_wall = off; off += P.n_wall_words * 4;
        // line-of-sight staging: one float per window cell, one area per wave of a multi-wave workgroup (4)
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;
        off = (off + 15) / 16 * 16;
        P.off_win = off; off += 4 * rmax * rmax * 4;
        // the line-of-sight masks of the rows listed for the shared observation writing (Env::walls_stage_masks)
        const int offp = (P.Rp - 1) / 2, offq = (P.Rq - 1) / 2, posp = P.Rp - 1 - offp, posq = P.Rq - 1 - offq;
        const int vw = (offp > offq ? offp : offq) + (posp > posq ? posp : posq) + 1;   // KParams::vis_w (ppg_validate_and_layout_gen2)
        P.off_vm = off; off += (64 + P.cap_prey) * ((vw * vw + 31) / 32) * 4;
    }
    P.lds_bytes = off;
    if (P.lds_bytes > 64 * 1024) return ppg_fail(h, PPG_EINVAL, "configuration needs %d bytes of LDS per wave (> 64 KiB)", P.lds_bytes);

    P.row_xy = b.row_xy; P.row_e = b.row_energy; P.row_id = b.row_id; P.row_key = b.row_key;
    P.row_cum = b.row_cumrew; P.row_flags = b.row_flags; P.row_reward = b.row_reward;
    P.env_state = b.env_state; P.env_seed = b.env_seed; P.grass_xy = b.grass_xy; P.grass_e = b.grass_energy;
    P.obs_pred = b.obs_pred; P.obs_prey = b.obs_prey; P.row_parent = b.row_parent; P.row_lastrep = b.row_lastrep;
    P.batch = h->batch;

    if (3 * P.map_n + 8 * c.grid_size + 8 > 32767) return ppg_fail(h, PPG_EINVAL, "grid too large for 16-bit map offsets");
    h->lut_host.assign((size_t)(P.nch_p + P.nch_q) * 128, 0u);
    ppg_build_lut(P.Rp, P.G, P.map_n, h->lut_host.data(), ch_p, drive);
    ppg_build_lut(P.Rq, P.G, P.map_n, h->lut_host.data() + (size_t)P.nch_p * 128, ch_q, drive);
    return PPG_OK;
}

// Second generation: express the common part through the base layout code, then add the gen-2 parameters.
static int ppg_validate_and_layout_gen2(ppg_handle *h) {
    const ppg_config_gen2 &g = h->cfg2;
    if (g.abi_version != PPG_ABI_VERSION) return ppg_fail(h, PPG_EINVAL, "abi_version %d != %d", g.abi_version, PPG_ABI_VERSION);
    int total = 0;
    for (int p = 0; p < 4; ++p) {
        if (g.n_possible[p] < 0 || g.n_possible[p] > 65535) return ppg_fail(h, PPG_EINVAL, "n_possible[%d] outside 0..65535", p);
        if (g.n_initial[p] < 0) return ppg_fail(h, PPG_EINVAL, "n_initial[%d] < 0", p);
        total += g.n_possible[p] > g.n_initial[p] ? g.n_possible[p] : g.n_initial[p];
    }
    if (total > 32767) return ppg_fail(h, PPG_EINVAL, "more than 32767 agent ids per episode (creation numbers are 15 bits)");
    const bool has_t2 = g.n_possible[1] > 0 || g.n_possible[3] > 0 || g.n_initial[1] > 0 || g.n_initial[3] > 0;
    const int ar1 = g.type_1_action_range, ar2 = has_t2 ? g.type_2_action_range : 1;
    if (ar1 < 1 || ar1 > 7 || !(ar1 & 1) || ar2 < 1 || ar2 > 7 || !(ar2 & 1))
        return ppg_fail(h, PPG_EINVAL, "action ranges must be odd and in 1..7");
    if (g.reproduction_cooldown_steps < 0 || g.reproduction_cooldown_steps > 1000000)
        return ppg_fail(h, PPG_EINVAL, "reproduction_cooldown_steps outside 0..1000000");
    if (!h->bufs.row_lastrep) return ppg_fail(h, PPG_EINVAL, "row_lastrep is NULL");
    if (g.walls && (!h->bufs.wall_bits || !h->bufs.row_info)) return ppg_fail(h, PPG_EINVAL, "walls need wall_bits and row_info");
    ppg_config &c = h->cfg;
    memset(&c, 0, sizeof c);
    c.abi_version = g.abi_version; c.grid_size = g.grid_size; c.predator_obs_range = g.predator_obs_range;
    c.prey_obs_range = g.prey_obs_range; c.max_steps = g.max_steps;
    c.n_possible_predators = g.n_possible[0] + g.n_possible[1]; c.n_possible_prey = g.n_possible[2] + g.n_possible[3];
    c.n_initial_predators = g.n_initial[0] + g.n_initial[1]; c.n_initial_prey = g.n_initial[2] + g.n_initial[3];
    c.n_grass = g.n_grass; c.pred_capacity = g.pred_capacity; c.prey_capacity = g.prey_capacity;
    c.grass_capacity = g.grass_capacity; c.obs_dtype = g.obs_dtype;
    c.energy_loss_per_step_predator = g.energy_loss_per_step_predator; c.energy_loss_per_step_prey = g.energy_loss_per_step_prey;
    c.predator_creation_energy_threshold = g.predator_creation_energy_threshold;
    c.prey_creation_energy_threshold = g.prey_creation_energy_threshold;
    c.initial_energy_predator = g.initial_energy_predator; c.initial_energy_prey = g.initial_energy_prey;
    c.initial_energy_grass = g.initial_energy_grass; c.energy_gain_per_step_grass = g.energy_gain_per_step_grass;
    c.season_high_multiplier = c.season_low_multiplier = 1.0;
    int rc = ppg_validate_and_layout(h);
    if (rc != PPG_OK) return rc;
    ppg::KParams &P = h->base;
    P.gen2 = 1;
    for (int p = 0; p < 4; ++p) { P.npos2[p] = g.n_possible[p]; P.ninit2[p] = g.n_initial[p]; }
    P.ar[0] = ar1; P.ar[1] = ar2;
    P.ar_inv[0] = (uint32_t)((65536 + ar1 - 1) / ar1); P.ar_inv[1] = (uint32_t)((65536 + ar2 - 1) / ar2);
    P.cooldown = g.reproduction_cooldown_steps;
    for (int t = 0; t < 2; ++t) {
        P.r2_catch[t] = g.reward_predator_catch_prey[t]; P.r2_eat[t] = g.reward_prey_eat_grass[t];
        P.r2_pstep[t] = g.reward_predator_step[t]; P.r2_qstep[t] = g.reward_prey_step[t];
        P.r2_caught[t] = g.penalty_prey_caught[t]; P.r2_repro_p[t] = g.reproduction_reward_predator[t];
        P.r2_repro_q[t] = g.reproduction_reward_prey[t];
    }
    P.move_factor = g.move_energy_cost_factor; P.cap_gain_prey = g.max_energy_gain_per_prey;
    P.cap_gain_grass = g.max_energy_gain_per_grass; P.max_e_pred = g.max_energy_predator; P.max_e_prey = g.max_energy_prey;
    P.cap_g = g.max_energy_grass; P.eff_transfer = g.energy_transfer_efficiency; P.eff_repro = g.reproduction_energy_efficiency;
    P.chance_p = g.reproduction_chance_predator; P.chance_q = g.reproduction_chance_prey;
    P.mut_p = g.mutation_rate_predator; P.mut_q = g.mutation_rate_prey;
    P.walls = g.walls ? 1 : 0;
    P.vis_channel = (g.walls && g.include_visibility_channel) ? 1 : 0;
    P.los_move = (g.walls && g.respect_los_for_movement) ? 1 : 0;
    P.mask_obs = (g.walls && g.mask_observation_with_visibility) ? 1 : 0;
    P.wall_bits = h->bufs.wall_bits; P.row_info = h->bufs.row_info;
    if (g.walls) {
        // window offsets that any observation can ask about: -off .. R-1-off per species (an even R reaches one cell further up)
        const int offp = (P.Rp - 1) / 2, offq = (P.Rq - 1) / 2;
        const int neg = offp > offq ? offp : offq;
        const int posp = P.Rp - 1 - offp, posq = P.Rq - 1 - offq;
        const int pos = posp > posq ? posp : posq;
        P.vis_neg = neg; P.vis_w = neg + pos + 1; P.vis_words = (P.vis_w * P.vis_w + 31) / 32;
    }
    return PPG_OK;
}

// Cooperative step kernels (Env's COOP): eligibility and LDS layout of one env's region.  The maps are padded by the larger
// window's reach, so the map offsets of an observation block's elements are position-independent (Env::coop_build_lut).
// One layout of the cooperative kernels' env region: with a channel-0 map (four maps) or without (three; Env::coop_pieces computes
// channel 0).  false: the configuration cannot have it.
static bool ppg_coop_layout_with(ppg_handle *h, int ch0_map, ppg::KParams &P, std::vector<uint32_t> &tab) {
    const ppg_config &c = h->cfg;
    P = h->base;
    const int offp = (P.Rp - 1) / 2, offq = (P.Rq - 1) / 2;
    P.ch0_map = ch0_map;
    P.pad = offp > offq ? offp : offq;
    P.Gp = P.G + 2 * P.pad;
    P.map_n = (P.Gp * P.Gp + 7) / 8 * 8;
    const int n_maps = 3 + ch0_map;
    if (3 * P.map_n + P.pad * P.Gp + P.pad > 32767) return false;       // 16-bit map offsets
    // The device reset lays its two arrays over the maps: three maps must hold G*G cells + the K placed entities, 16 bits each (Env::do_reset)
    if (!ch0_map && 2 * (((c.grid_size * c.grid_size + 7) & ~7) + c.n_initial_predators + c.n_initial_prey + c.n_grass) > 3 * P.map_n) return false;
    int off = 0;
    P.off_map = off; off += n_maps * P.map_n;
    off = (off + 15) / 16 * 16;
    P.off_val = off; off += (196 + c.n_grass) * 8;      // packed sections (Env::SEC_Q, SEC_G)
    off = (off + 15) / 16 * 16;
    const int cap_max = P.cap_pred > P.cap_prey ? P.cap_pred : P.cap_prey;
    int scr_bytes = cap_max * 8 > (64 + P.cap_prey) * 4 ? cap_max * 8 : (64 + P.cap_prey) * 4;   // compaction / the two row lists
    if (scr_bytes < 1024) scr_bytes = 1024;                                                       // (reset: 256 random words)
    P.off_scr = off; off += scr_bytes;
    P.off_lut = off;
    if (h->gen2 && h->cfg2.walls) {   // walls (ppgc3_step): each env region carries its wall bitmap and four line-of-sight staging areas
        off = (off + 15) / 16 * 16;
        P.off_wall = off; off += P.n_wall_words * 4;
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;
        off = (off + 15) / 16 * 16;
        P.off_win = off; off += 4 * rmax * rmax * 4;
        P.off_vm = off; off += (64 + P.cap_prey) * P.vis_words * 4;
    }
    P.lds_env_bytes = (off + 15) / 16 * 16;
    P.bp_magic = (uint32_t)((0x100000000ull + (uint64_t)P.blk_p - 1) / (uint64_t)P.blk_p);
    P.bq_magic = (uint32_t)((0x100000000ull + (uint64_t)P.blk_q - 1) / (uint64_t)P.blk_q);
    // KParams::coop_tab: the observation descriptors of both species (then, with a channel-0 map, that map for an empty grid)
    tab.assign((size_t)(P.blk_p + P.blk_q) + (ch0_map ? (size_t)P.map_n / 4 : 0), 0u);
    for (int t = 0; t < 2; ++t) {
        const int R = t ? P.Rq : P.Rp, o = (R - 1) / 2;
        uint32_t *out = tab.data() + (t ? P.blk_p : 0);
        for (int e = 0; e < (t ? P.blk_q : P.blk_p); ++e) {
            const int ch = e / (R * R), i = (e % (R * R)) / R, j = e % R;
            if (ch == 0 && !ch0_map) {   // "outside the grid" (BASE:520-523): the window offsets, no map
                out[e] = 0xFFFF0000u | ((uint32_t)(i - o + 8) << 4) | (uint32_t)(j - o + 8);
                continue;
            }
            const int moff = (ch - 1 + ch0_map) * P.map_n + (i - o) * P.Gp + (j - o);
            const int section = ch == 2 ? 66 : ch == 3 ? 66 + 129 : 0;     // Env::map_base of the cooperative kernels (SEC_Q, SEC_G)
            out[e] = ((uint32_t)moff & 0xFFFFu) | ((uint32_t)section << 16);
        }
    }
    if (ch0_map) {
        unsigned char *m0 = (unsigned char *)(tab.data() + P.blk_p + P.blk_q);
        for (int x = 0; x < P.Gp; ++x)
            for (int y = 0; y < P.Gp; ++y)
                if (x < P.pad || x >= P.pad + P.G || y < P.pad || y >= P.pad + P.G) m0[x * P.Gp + y] = 65;   // Env::ONE_IDX
    }
    return true;
}
// bytes of the workgroup's descriptor table (the walls variant writes whole rows through obs_row_walls_in: none)
static int ppg_coop_table_bytes(const ppg::KParams &P) { return P.walls ? 0 : ((P.blk_p + P.blk_q) * 4 + 15) / 16 * 16; }
static int ppg_coop_lds_bytes_of(const ppg::KParams &P, int e) {
    return e * P.lds_env_bytes + ppg_coop_table_bytes(P) + 128 * 4;   // env regions, descriptor table, Env::CTL_WORDS
}
// Cooperative step kernels (Env's COOP): eligibility and LDS layout of one env's region.  The maps are padded by the larger
// window's reach, so the map offsets of an observation block's elements are position-independent.  Four maps (channel 0's halo points
// at a constant 1.0: every element is one uniform lookup) where four two-env workgroups then fit a CU's LDS -- every grid up to
// about 45x45; else three maps and channel 0 computed per element (64x64 grids: 24.1 -> 18.1 KB per env, three -> four workgroups per
// CU, +10 % -- profiles/r06/a_*; the arithmetic costs the second generation's float32 rows on a 25x25 grid 11 %, so they keep the map).
// Both layouts are kept: the three-map one serves four-wave step launches only (kernels ppgcm_*); rollouts, the 6 / 8 / 16-wave and
// the 64-register builds use the four-map layout whatever the grid.  PPG_COOP_MAPS=3 / 4 (read at create): force either (tests, A/B runs).
static void ppg_coop_layout(ppg_handle *h) {
    h->coop_ok = 0; h->coop3_ok = 0; h->coop_prefers3 = 0; h->coop3_tab_off = 0;
    const ppg_config &c = h->cfg;
    const bool walls = h->gen2 && h->cfg2.walls;
    if (h->drive || c.kickback || h->nq > 2) return;                    // (8-bit maps; generic 4-channel observations only)
    std::vector<uint32_t> t3;
    if (walls) {   // (round 6) ppgc3_step: the three-map layout only -- channel 0 of its observations is the wall bitmap; whole rows
                   // through obs_row_walls_in, which reads the maps for in-grid cells only: even windows too
        h->coop_ok = ppg_coop_layout_with(h, 0, h->coop, h->coop_tab_host) ? 1 : 0;
        return;
    }
    if (!(c.predator_obs_range & 1) || !(c.prey_obs_range & 1)) return;  // even windows keep the element-descriptor kernels
    if (!ppg_coop_layout_with(h, 1, h->coop, h->coop_tab_host)) return;
    h->coop_ok = 1;
    if (ppg_coop_layout_with(h, 0, h->coop3, t3)) {
        h->coop3_ok = 1;
        h->coop3_tab_off = (uint32_t)h->coop_tab_host.size();
        h->coop_tab_host.insert(h->coop_tab_host.end(), t3.begin(), t3.end());
        const char *force = getenv("PPG_COOP_MAPS");
        h->coop_prefers3 = ppg_coop_lds_bytes_of(h->coop, 2) * 4 > 160 * 1024;
        if (force && force[0] == '3') h->coop_prefers3 = 1;
        if (force && force[0] == '4') h->coop_prefers3 = 0;
    }
}
// dynamic LDS of a cooperative workgroup of `e` envs: env regions, descriptor table, control words
// (of the layout a four-wave cooperative step launch uses: what the plan's occupancy rules are about)
static int ppg_coop_lds_bytes(const ppg_handle *h, int e) { return ppg_coop_lds_bytes_of(h->coop_prefers3 ? h->coop3 : h->coop, e); }

// How many wavefronts step one env (wave 0 runs the transition; all of them write the final observations), and from how many
// agent rows on the helper wavefronts of an env stay (lighter envs are left to wave 0: Env::helpers).  Measured on MI355X:
//  - up to 512 envs in flight the GPU is nearly empty: 8 waves per env (16 up to 256 envs); up to ~3072: 4 waves (256 envs 1.8x,
//    1024 1.7x, 2048 1.3x);
//  - walls / drive variants are bound by per-row work: 4 waves at every batch size (4096 envs: 1.7-1.9x), always all of
//    them: a second, wave-0-only copy of their large observation code in the same kernel cost 10-15 % (Env::ADAPTIVE_HELPERS);
//  - a FULL GPU (> 3072 envs in flight) runs as fast as the slowest env of a launch lets it, and that is always a heavy one.  Base
//    family: the cooperative kernels (two envs per four-wave workgroup, Env's COOP) where the configuration has them and four
//    workgroups fit a CU's LDS (64x64 grids with 7x7 windows just do); else a PAIR of waves per env (8-bit maps; a pair fills the 16
//    wave slots, +27 % over one wave); second generation (float32 observations: no longer store-bound): four waves, helpers only for envs
//    with >= 72 rows -- the stragglers -- +25 % (48.0 -> 60.2 M env-steps/s), where helpers for every env cost 6 %.
// The plan is computed when the handle is created and when ppg_set_envs_in_flight / ppg_set_wave_plan change its inputs -- never
// per step.  ppg_set_wave_plan overrides it (tests, A/B tools).
static ppg_wave_plan_t ppg_wave_plan(const ppg_handle *h) {
    const int lds_envs = h->base.lds_bytes > 0 ? (160 * 1024) / h->base.lds_bytes : 16;
    const int in_flight = h->envs_in_flight > 0 ? h->envs_in_flight : h->batch;
    const bool walls = h->gen2 && h->cfg2.walls;
    ppg_wave_plan_t p = {1, 0, 0};
    if (h->forced.nw > 0) {
        p = h->forced;
        if (walls || h->drive) {   // (pair and four-wave kernels, helpers always stay; walls: the four-wave cooperative kernel)
            p.nw = p.nw == 2 ? 2 : p.nw > 1 ? 4 : 1; p.min_rows = 0;
            if (!(walls && p.nw == 4)) p.coop_e = 0;
        }
        if (h->cfg.kickback) { p.nw = 1; p.coop_e = 0; }
        if (p.coop_e > 0 && !h->coop_ok) p.coop_e = 0;
        if (p.coop_e > 0) {
            if (p.nw != 4 && p.nw != 6 && p.nw != 8 && p.nw != 16) p.nw = 4;
            if (h->gen2) p.nw = 4;   // (the second generation has four-wave cooperative kernels only)
            if (p.coop_e > p.nw) p.coop_e = p.nw;
            p.min_rows = 0;
        } else {
            if (p.nw != 1 && p.nw != 2 && p.nw != 4 && p.nw != 8 && p.nw != 16) p.nw = 4;
            if (p.nw == 2 && h->gen2 && !walls) p.nw = 4;                                      // no pair kernels for the second generation without walls
            if (p.nw == 16 && (h->gen2 || h->base.nch_p > 2 || h->base.nch_q > 3 || h->nq > 2)) p.nw = 8;  // sixteen waves: base family, register descriptors, <= 128 prey rows
        }
        return p;
    }
    if (h->cfg.kickback) return p;   // (single-wave kernels only)
    if (h->drive) {
        // a full GPU: two waves per env keep twice the envs in flight at 128 registers (183 vs 192 us per 4096-env step, round 5);
        // the walls variant's observation phase is the longer part of its env, four waves stay faster there (95 vs 106 us)
        p.nw = in_flight > 3072 ? 2 : 4;
    } else if (walls) {
        p.nw = 4;
        // (round 6) a full GPU: the cooperative walls kernel, two envs per four-wave workgroup at 113 registers without scratch (the
        // four-wave kernel runs at a 64-register cap with 37 spilled) -- 77.3 against 86.5 us per 4096-env step, alternating processes
        // on one box; three envs per workgroup 79.6, four 88.7; at 96 registers (five workgroups per CU) 79.8 (profiles/r06/u_*).  Both
        // with the rows' window cells written as one run (Env::obs_cells_walls), which alone took the four-wave kernel from 92.1 to 86.5.
        if (in_flight > 3072 && h->coop_ok && ppg_coop_lds_bytes(h, 2) * 4 <= 160 * 1024) p.coop_e = 2;
    } else if (in_flight <= 512) {
        p.nw = 8;
        // up to 256 envs one workgroup per CU is all there is: sixteen waves (base family, register-descriptor observation path):
        // 256 envs 11.3 -> 12.1 M env-steps/s; at 512 envs eight are faster (20.5 vs 17.8 M)
        // (not with 256 prey rows: a 1024-thread workgroup caps the registers at 128 and four prey row registers spill there)
        if (in_flight <= 256 && !h->gen2 && h->base.nch_p <= 2 && h->base.nch_q <= 3 && h->nq <= 2) p.nw = 16;
    } else if (in_flight <= 3072) {
        p.nw = 4;
    } else if (h->gen2 && !(h->coop_ok && ppg_coop_lds_bytes(h, 2) * 6 <= 160 * 1024)) {
        p.nw = 4;
        p.min_rows = 72;
    } else if (h->coop_ok && ppg_coop_lds_bytes(h, 2) * (h->gen2 ? 6 : 4) <= 160 * 1024) {
        // (base family from FOUR workgroups per CU on -- round 6, 64x64 grids / BASELINE config 4: three cell maps per env instead of
        // four make it 4 x 2 envs per CU, 60.0 us per 4096-env step against 63.8 for the pair kernel at the same eight envs per CU,
        // interleaved in one process: profiles/r06/a_*)
        // (second generation, float32 observations, 3 sub-batches on one box: 56.5 us per 4096-env step against 68.1 for its
        // four-wave kernel with helpers from 72 rows; four envs per workgroup 71.7, three 63.7)
        // cooperative kernels, TWO envs per four-wave workgroup (six workgroups = 24 waves = 12 envs per CU): two waves run a
        // transition each, the other two wait at the barrier, then all four write both envs' observations as whole 1 KB pieces.
        // Interleaved A/B on one box (tools/ab_plans.py, 4096 envs, 2 / 3 sub-batches): 68.3 / 71.2 us per step against 74.0 /
        // 76.9 for the pair kernel; four envs per workgroup (every wave runs a transition) 72.2 / 75.9, three 70.1 / 74.6.
        p.nw = 4;
        p.coop_e = 2;
    } else {
        p.nw = lds_envs <= 4 ? 4 : 2;
    }
    return p;
}

static int backend_init(ppg_handle *h, int device);
static void backend_release(ppg_handle *h);
static int backend_launch(ppg_handle *h, int mode, const ppg::KParams &P, void *stream);
// (re)compute h->order_dev (allocating it on first use): envs sorted by descending weight_pred * predator rows +
// weight_prey * prey rows, ties by env index -- always a permutation of 0..batch-1
static int backend_rebalance(ppg_handle *h, int weight_pred, int weight_prey, void *stream);
// snapshot plumbing: a copy between host memory and a caller-owned device tensor, enqueued on `stream`; and the wait
static int backend_copy(ppg_handle *h, void *dst, const void *src, size_t bytes, bool to_device, void *stream);
static int backend_alloc(ppg_handle *h, void **out, size_t bytes);   // device memory owned by the handle
static int backend_sync(ppg_handle *h, void *stream);
// the two launches of ppg_pack (ppg_pack.h)
static int backend_pack(ppg_handle *h, const ppg::PackParams &K, void *stream);
// the launch of ppg_fetch (ppg_fetch.h), and the release of a buffer from backend_alloc
static int backend_fetch(ppg_handle *h, const ppg::FetchParams &K, void *stream);
static void backend_free(ppg_handle *h, void *p);

// The state tensors of one env, in image order (include/ppg.h: ppg_state_header): pointer, bytes per env.
struct ppg_state_field { void *base; size_t bytes; };
static int ppg_state_fields(const ppg_handle *h, ppg_state_field (&f)[16]) {
    const ppg_buffers &b = h->bufs;
    const size_t S = (size_t)h->base.S, NG = (size_t)h->base.cap_grass, W = (size_t)h->base.n_wall_words;
    const bool walls = h->gen2 && h->cfg2.walls;
    int n = 0;
    f[n++] = {b.row_xy, S * 2}; f[n++] = {b.row_energy, S * 8}; f[n++] = {b.row_id, S * 4}; f[n++] = {b.row_key, S * 4};
    f[n++] = {b.row_cumrew, S * 8}; f[n++] = {b.row_flags, S}; f[n++] = {b.row_reward, S * 8};
    f[n++] = {b.row_parent, S * 4};
    f[n++] = {b.env_state, (size_t)PPG_ENV_WORDS * 4}; f[n++] = {b.env_seed, 8};
    f[n++] = {b.grass_xy, NG * 2}; f[n++] = {b.grass_energy, NG * 8};
    if (h->gen2) f[n++] = {b.row_lastrep, S * 4};
    if (walls) { f[n++] = {b.row_info, S}; f[n++] = {b.wall_bits, W * 4}; }
    return n;
}

extern "C" {

int ppg_abi_version(void) { return PPG_ABI_VERSION; }

uint32_t ppg_lexkey(uint32_t id) { return ppg_host_lexkey(id); }

static int ppg_create_common(const ppg_config *cfg, const ppg_config_gen2 *cfg2, int32_t batch, int32_t device,
                             const ppg_buffers *bufs, ppg_handle **out) {
    if ((!cfg && !cfg2) || !bufs || !out) return ppg_fail(nullptr, PPG_EINVAL, "null argument");
    ppg_handle *h = new (std::nothrow) ppg_handle();
    if (!h) return ppg_fail(nullptr, PPG_ENOMEM, "out of host memory");
    h->gen2 = cfg2 ? 1 : 0;
    if (cfg) h->cfg = *cfg; else h->cfg2 = *cfg2;
    h->bufs = *bufs; h->batch = batch; h->device = device;
    h->lut_dev = nullptr; h->backend = nullptr; h->prof_dev = nullptr; h->err[0] = 0;
    h->coop_ok = 0; h->coop3_ok = 0; h->coop_prefers3 = 0; h->coop3_tab_off = 0; h->coop_tab_dev = nullptr;
    h->forced = {0, 0, 0};
    h->envs_in_flight = 0;
    {   // scheduling knobs of the experiments, read ONCE per handle (never per step)
        const char *ev = getenv("PPG_COOP_WGS_PER_CU"), *pad = getenv("PPG_STEP_LDS_PAD");
        h->coop_wgs_per_cu = ev ? atoi(ev) : 5;
        h->step_lds_pad = pad ? atoi(pad) : 0;
    }
    h->order_dev = nullptr;
    h->vis_dev = nullptr;
    h->fetch_dev = nullptr; h->fetch_cap = 0; h->fetch_hint = 0;
    h->drive = (cfg && (cfg->n_drive[0] > 0 || cfg->n_drive[1] > 0)) ? 1 : 0;
    int rc = cfg2 ? ppg_validate_and_layout_gen2(h) : ppg_validate_and_layout(h);
    if (rc == PPG_OK) ppg_coop_layout(h);
    if (rc == PPG_OK) rc = backend_init(h, device);
    if (rc != PPG_OK) {
        memcpy(g_ppg_create_error, h->err, sizeof g_ppg_create_error);
        backend_release(h);
        delete h;
        return rc;
    }
    h->base.obs_lut = h->lut_dev;
    h->coop.obs_lut = h->lut_dev;
    h->coop.coop_tab = h->coop_tab_dev;
    h->coop3.obs_lut = h->lut_dev;
    h->coop3.coop_tab = h->coop_tab_dev ? h->coop_tab_dev + h->coop3_tab_off : nullptr;
    h->plan = ppg_wave_plan(h);
    *out = h;
    return PPG_OK;
}

int ppg_create(const ppg_config *cfg, int32_t batch, int32_t device, const ppg_buffers *bufs, ppg_handle **out) {
    return ppg_create_common(cfg, nullptr, batch, device, bufs, out);
}

int ppg_create_gen2(const ppg_config_gen2 *cfg, int32_t batch, int32_t device, const ppg_buffers *bufs, ppg_handle **out) {
    return ppg_create_common(nullptr, cfg, batch, device, bufs, out);
}

int ppg_destroy(ppg_handle *h) {
    if (!h) return PPG_OK;
    backend_release(h);
    delete h;
    return PPG_OK;
}

int ppg_reset(ppg_handle *h, const uint64_t *seeds, uint32_t episode, void *stream) {
    if (!h) return PPG_EINVAL;
    ppg::KParams P = h->base;
    P.mode = ppg::MODE_RESET; P.seeds = seeds; P.reset_episode = episode;
    return backend_launch(h, ppg::MODE_RESET, P, stream);
}

int ppg_observe(ppg_handle *h, void *stream);

int ppg_reset_from_state(ppg_handle *h, const ppg_init_state *init, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!init || !init->grass_xy || (!init->pred_xy && h->cfg.n_initial_predators > 0) || (!init->prey_xy && h->cfg.n_initial_prey > 0))
        return ppg_fail(h, PPG_EINVAL, "ppg_reset_from_state: an array of ppg_init_state is NULL");
    if (h->gen2) return ppg_fail(h, PPG_EINVAL, "ppg_reset_from_state takes base-family handles (second generation: write the tensors, then ppg_observe)");
    const ppg::KParams &P = h->base;
    const size_t B = (size_t)h->batch, S = (size_t)P.S, NG = (size_t)P.cap_grass;
    const int P0 = h->cfg.n_initial_predators, Q0 = h->cfg.n_initial_prey, n_grass = h->cfg.n_grass, G = P.G;
    std::vector<uint16_t> xy(B * S, 0), gxy(B * NG, 0);
    std::vector<double> en(B * S, 0.0), zeros(B * S, 0.0), ge(B * NG, 0.0);
    std::vector<int32_t> ids(B * S, 0), par(B * S, -1), es(B * PPG_ENV_WORDS, 0);
    std::vector<uint32_t> keys(B * S, 0);
    std::vector<uint8_t> fl(B * S, 0);
    std::vector<int32_t> owner((size_t)G * G);
    for (size_t b = 0; b < B; ++b) {
        for (int t = 0; t < 2; ++t) {
            const int n = t ? Q0 : P0, lo = t ? P.cap_pred : 0;
            const uint16_t *src = (t ? init->prey_xy : init->pred_xy) + b * (size_t)n;
            std::fill(owner.begin(), owner.end(), -1);
            for (int i = 0; i < n; ++i) {
                const int x = src[i] >> 8, y = src[i] & 255;
                if (x >= G || y >= G) return ppg_fail(h, PPG_EINVAL, "env %zu: %s %d at (%d, %d) is outside the %dx%d grid", b, t ? "prey" : "predator", i, x, y, G, G);
                const size_t s = b * S + lo + i;
                xy[s] = src[i]; en[s] = t ? h->cfg.initial_energy_prey : h->cfg.initial_energy_predator;
                ids[s] = i; keys[s] = ppg_host_lexkey((uint32_t)i);
                owner[(size_t)x * G + y] = i;    // grid[type, pos] = energy in id order: the last writer owns the cell (BASE:190-200)
            }
            for (int c = 0; c < G * G; ++c) if (owner[c] >= 0) fl[b * S + lo + owner[c]] = PPG_ROW_OWNS;
        }
        std::fill(owner.begin(), owner.end(), -1);
        for (int p = 0; p < n_grass; ++p) {
            const uint16_t c = init->grass_xy[b * (size_t)n_grass + p];
            const int x = c >> 8, y = c & 255;
            if (x >= G || y >= G) return ppg_fail(h, PPG_EINVAL, "env %zu: grass patch %d at (%d, %d) is outside the grid", b, p, x, y);
            if (owner[(size_t)x * G + y] >= 0) return ppg_fail(h, PPG_EINVAL, "env %zu: grass positions must be unique", b);
            owner[(size_t)x * G + y] = p;
            gxy[b * NG + p] = c; ge[b * NG + p] = h->cfg.initial_energy_grass;
        }
        int32_t *w = &es[b * PPG_ENV_WORDS];
        w[PPG_ENV_N_PRED_ROWS] = P0; w[PPG_ENV_N_PREY_ROWS] = Q0; w[PPG_ENV_NEXT_PRED_ID] = P0; w[PPG_ENV_NEXT_PREY_ID] = Q0;   // BASE:153-154
        w[PPG_ENV_N_PRED_ALIVE] = P0; w[PPG_ENV_N_PREY_ALIVE] = Q0;                                                             // BASE:210-211
        w[PPG_ENV_FLAGS] = PPG_ENVF_WAS_RESET | PPG_ENVF_LIST_IS_ROW_ORDER;
        w[PPG_ENV_EPISODE] = (int32_t)init->episode;
    }
    const ppg_buffers &bf = h->bufs;
    const struct { void *dst; const void *src; size_t bytes; } copies[] = {
        {bf.row_xy, xy.data(), B * S * 2}, {bf.row_energy, en.data(), B * S * 8}, {bf.row_id, ids.data(), B * S * 4},
        {bf.row_key, keys.data(), B * S * 4}, {bf.row_flags, fl.data(), B * S}, {bf.row_cumrew, zeros.data(), B * S * 8},
        {bf.row_reward, zeros.data(), B * S * 8}, {bf.row_parent, par.data(), B * S * 4}, {bf.env_state, es.data(), B * PPG_ENV_WORDS * 4},
        {bf.grass_xy, gxy.data(), B * NG * 2}, {bf.grass_energy, ge.data(), B * NG * 8}};
    for (const auto &c : copies) {
        const int rc = backend_copy(h, c.dst, c.src, c.bytes, true, stream);
        if (rc != PPG_OK) return rc;
    }
    int rc = backend_sync(h, stream);   // (the staging vectors die with this call)
    if (rc != PPG_OK) return rc;
    return ppg_observe(h, stream);
}

int ppg_observe(ppg_handle *h, void *stream) {
    if (!h) return PPG_EINVAL;
    ppg::KParams P = h->base;
    P.mode = ppg::MODE_OBSERVE;
    return backend_launch(h, ppg::MODE_OBSERVE, P, stream);
}

// the parameter block of a row-order step as the handle's wave plan wants it launched
static bool ppg_coop_high_occupancy(const ppg_handle *h, int coop_e);
// (fused: ppg_rollout -- the fused kernels exist for the four-map layout only, like the 6 / 8 / 16-wave and the 64-register builds)
static ppg::KParams ppg_planned_step_params(const ppg_handle *h, bool fused = false) {
    const ppg_wave_plan_t &wp = h->plan;
    const bool three = wp.coop_e > 0 && h->coop_prefers3 && wp.nw == 4 && !fused && !ppg_coop_high_occupancy(h, wp.coop_e);
    ppg::KParams P = wp.coop_e > 0 ? (three ? h->coop3 : h->coop) : h->base;
    if (wp.coop_e > 0) {   // cooperative kernels: wp.coop_e env regions, then the workgroup's descriptor table and control words
        P.coop_e = wp.coop_e;
        P.off_lut2 = wp.coop_e * P.lds_env_bytes;
        P.off_ctl = P.off_lut2 + ppg_coop_table_bytes(P);
        P.lds_bytes = ppg_coop_lds_bytes_of(P, wp.coop_e);
        // float64 / float32 rows: at most FIVE four-wave workgroups (ten envs) per CU although registers and LDS admit six -- a CU
        // then has fewer scattered write streams open at a time: 66.8 -> 68.2 M env-steps/s on the headline workload, the driver's
        // command 62.1 -> 63.0 M (profiles/r05/n_coop_workgroups_per_cu.txt; four: the same; three: -13 %).  Done by asking for LDS
        // the kernel does not use.  bfloat16 rows go the other way (ppgch_step: eight per CU).  PPG_COOP_WGS_PER_CU=0: no limit.
        // (obs_f32: the handle's normalised row dtype, both generations -- 0 float64, 1 float32, 2 / 3 bfloat16)
        if (wp.nw == 4 && h->base.obs_f32 < 2) {
            const int per_cu = h->coop_wgs_per_cu;
            if (per_cu > 0 && per_cu < 16) {
                const int floor_bytes = 160 * 1024 / (per_cu + 1) + 16;
                if (P.lds_bytes < floor_bytes) P.lds_bytes = floor_bytes;
            }
        }
        P.env_order = h->base.env_order;
        P.vis_masks = h->base.vis_masks;
        P.vis_env_stride = h->base.vis_env_stride;
    }
    else P.lds_bytes += h->step_lds_pad;   // experiment (profiles/r05/s_*): fewer envs per CU for the multi-wave kernels
    P.helper_min_rows = wp.min_rows;
    return P;
}

int ppg_step(ppg_handle *h, const int8_t *actions, uint32_t flags, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!actions && !(flags & PPG_STEP_RANDOM_ACTIONS)) return ppg_fail(h, PPG_EINVAL, "actions is NULL without PPG_STEP_RANDOM_ACTIONS");
    if (flags & ~(PPG_STEP_RANDOM_ACTIONS | PPG_STEP_AUTO_RESET)) return ppg_fail(h, PPG_EINVAL, "unknown step flags 0x%x", flags);
    const int mode = h->cfg.kickback ? ppg::MODE_STEP_KICK : ppg::MODE_STEP;
    ppg::KParams P = ppg_planned_step_params(h);
    P.mode = mode; P.actions = actions; P.flags = flags; P.prof = h->prof_dev; P.n_steps = 1;
    return backend_launch(h, mode, P, stream);
}

int ppg_rollout(ppg_handle *h, int32_t n_steps, const int8_t *actions, uint32_t flags, void *stream) {
    if (!h) return PPG_EINVAL;
    if (n_steps < 1) return ppg_fail(h, PPG_EINVAL, "n_steps must be >= 1");
    if (h->cfg.kickback) return ppg_fail(h, PPG_EINVAL, "ppg_rollout does not support the kickback variant");
    // second generation: only as the fused form of its cooperative kernel (reproduction uniforms from the device's Philox streams,
    // as in ppg_step); the walls variant has no cooperative kernel
    if (h->gen2 && (h->cfg2.walls || !(h->plan.coop_e > 0 && h->plan.nw == 4)))
        return ppg_fail(h, PPG_EINVAL, "ppg_rollout on a second-generation handle needs a cooperative four-wave plan (ppg_set_wave_plan) and no walls");
    if (h->drive) return ppg_fail(h, PPG_EINVAL, "ppg_rollout does not support the drive-conditioned variant");
    if (!actions && !(flags & PPG_STEP_RANDOM_ACTIONS)) return ppg_fail(h, PPG_EINVAL, "actions is NULL without PPG_STEP_RANDOM_ACTIONS");
    if (flags & ~(PPG_STEP_RANDOM_ACTIONS | PPG_STEP_AUTO_RESET)) return ppg_fail(h, PPG_EINVAL, "unknown step flags 0x%x", flags);
    // a handle whose plan is cooperative with four-wave workgroups runs the fused form of that kernel; else one wave per env
    ppg::KParams P = (h->plan.coop_e > 0 && h->plan.nw == 4) ? ppg_planned_step_params(h, true) : h->base;
    P.mode = ppg::MODE_ROLLOUT; P.actions = actions; P.flags = flags; P.prof = h->prof_dev; P.n_steps = n_steps;
    return backend_launch(h, ppg::MODE_ROLLOUT, P, stream);
}

int ppg_step_many(ppg_handle *const *handles, int32_t n, const int8_t *const *actions, uint32_t flags,
                  void *const *streams) {
    if (!handles || !streams || n < 1) return PPG_EINVAL;
    for (int32_t k = 0; k < n; ++k) {
        int rc = ppg_step(handles[k], actions ? actions[k] : nullptr, flags, streams[k]);
        if (rc != PPG_OK) return rc;
    }
    return PPG_OK;
}

int ppg_step_ordered(ppg_handle *h, const int8_t *actions, const uint8_t *act_rank, uint32_t flags, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!actions) return ppg_fail(h, PPG_EINVAL, "actions is NULL");
    if (flags & ~PPG_STEP_AUTO_RESET) return ppg_fail(h, PPG_EINVAL, "ppg_step_ordered takes only PPG_STEP_AUTO_RESET");
    ppg::KParams P = h->base;
    const int mode = act_rank ? (h->cfg.kickback ? ppg::MODE_STEP_ORDERED_KICK : ppg::MODE_STEP_ORDERED)
                              : (h->cfg.kickback ? ppg::MODE_STEP_KICK : ppg::MODE_STEP);
    P.mode = mode; P.actions = actions; P.act_rank = act_rank; P.flags = flags; P.prof = h->prof_dev; P.n_steps = 1;
    return backend_launch(h, mode, P, stream);
}

int ppg_step_uniforms(ppg_handle *h, const int8_t *actions, const uint8_t *act_rank, const double *uniforms,
                      int32_t uniforms_per_env, uint32_t flags, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!h->gen2) return ppg_fail(h, PPG_EINVAL, "ppg_step_uniforms needs a handle from ppg_create_gen2");
    if (!uniforms || uniforms_per_env < 1) return ppg_fail(h, PPG_EINVAL, "uniforms is NULL or empty");
    if (!actions && !(flags & PPG_STEP_RANDOM_ACTIONS)) return ppg_fail(h, PPG_EINVAL, "actions is NULL without PPG_STEP_RANDOM_ACTIONS");
    if (flags & ~(PPG_STEP_RANDOM_ACTIONS | PPG_STEP_AUTO_RESET)) return ppg_fail(h, PPG_EINVAL, "unknown step flags 0x%x", flags);
    if (act_rank && (flags & PPG_STEP_RANDOM_ACTIONS)) return ppg_fail(h, PPG_EINVAL, "act_rank with PPG_STEP_RANDOM_ACTIONS");
    ppg::KParams P = act_rank ? h->base : ppg_planned_step_params(h);   // (row order: the kernel the wave plan picks, like ppg_step)
    const int mode = act_rank ? ppg::MODE_STEP_ORDERED : ppg::MODE_STEP;
    P.mode = mode; P.actions = actions; P.act_rank = act_rank; P.flags = flags; P.prof = h->prof_dev; P.n_steps = 1;
    P.uniforms = uniforms; P.uniforms_per_env = uniforms_per_env;
    return backend_launch(h, mode, P, stream);
}

int ppg_rebalance(ppg_handle *h, void *stream) {
    if (!h) return PPG_EINVAL;
    const ppg::KParams &P = h->base;
    const int rc = backend_rebalance(h, P.nch_p, P.nch_q, stream);   // weight = 128-element chunks per observation
    if (rc == PPG_OK) h->base.env_order = h->order_dev;
    return rc;
}

int ppg_walls_changed(ppg_handle *h, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!(h->gen2 && h->cfg2.walls)) return ppg_fail(h, PPG_EINVAL, "ppg_walls_changed needs a walls handle (ppg_config_gen2.walls)");
    const ppg::KParams &B = h->base;
    if (!h->vis_dev) {
        const int rc = backend_alloc(h, (void **)&h->vis_dev, (size_t)h->batch * B.G * B.G * B.vis_words * sizeof(uint32_t));
        if (rc != PPG_OK) return rc;
    }
    ppg::KParams P = h->base;
    P.mode = ppg::MODE_VIS; P.vis_masks = h->vis_dev;
    int rc = backend_launch(h, ppg::MODE_VIS, P, stream);
    if (rc != PPG_OK) return rc;
    // One wall layout for the whole batch (the usual case)?  Then every env reads env 0's masks: a table of a few KB that stays in
    // L2 instead of a scattered read from [batch, G*G, vis_words] in HBM.  Decided here, on the host, from the bitmaps themselves
    // (batch x n_wall_words words: 320 KB for 4096 envs of 25x25) -- setting walls is not on the step path.
    const size_t words = (size_t)B.n_wall_words;
    uint32_t *bits = (uint32_t *)malloc((size_t)h->batch * words * sizeof(uint32_t));
    if (!bits) return ppg_fail(h, PPG_ENOMEM, "ppg_walls_changed: no host memory for the bitmap comparison");
    rc = backend_copy(h, bits, h->bufs.wall_bits, (size_t)h->batch * words * sizeof(uint32_t), false, stream);
    if (rc == PPG_OK) rc = backend_sync(h, stream);
    bool same = rc == PPG_OK;
    for (int b = 1; same && b < h->batch; ++b) same = memcmp(bits, bits + (size_t)b * words, words * sizeof(uint32_t)) == 0;
    free(bits);
    if (rc != PPG_OK) return rc;
    h->base.vis_masks = h->vis_dev;   // every later launch reads the masks instead of walking the lines
    if (getenv("PPG_VIS_PER_ENV")) same = false;   // (A/B switch: every env reads its own table whatever the bitmaps say)
    h->base.vis_env_stride = same ? 0 : B.G * B.G;
    return rc;
}

int ppg_get_buffers(const ppg_handle *h, ppg_buffers *out) {
    if (!h || !out) return PPG_EINVAL;
    *out = h->bufs;
    return PPG_OK;
}

int ppg_set_envs_in_flight(ppg_handle *h, int32_t envs_in_flight) {
    if (!h) return PPG_EINVAL;
    if (envs_in_flight < 0) return ppg_fail(h, PPG_EINVAL, "envs_in_flight < 0");
    h->envs_in_flight = envs_in_flight;
    h->plan = ppg_wave_plan(h);
    return PPG_OK;
}

int ppg_set_wave_plan(ppg_handle *h, int32_t waves, int32_t helper_min_rows, int32_t coop_envs) {
    if (!h) return PPG_EINVAL;
    if (waves < 0 || helper_min_rows < 0 || coop_envs < 0) return ppg_fail(h, PPG_EINVAL, "negative wave plan");
    h->forced = {waves, helper_min_rows, coop_envs};
    h->plan = ppg_wave_plan(h);
    return PPG_OK;
}

int ppg_get_wave_plan(const ppg_handle *h, int32_t *waves, int32_t *helper_min_rows, int32_t *coop_envs) {
    if (!h) return PPG_EINVAL;
    if (waves) *waves = h->plan.nw;
    if (helper_min_rows) *helper_min_rows = h->plan.min_rows;
    if (coop_envs) *coop_envs = h->plan.coop_e;
    return PPG_OK;
}

int ppg_export_grid(ppg_handle *h, double *grid_out, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!grid_out) return ppg_fail(h, PPG_EINVAL, "grid_out is NULL");
    ppg::KParams P = h->base;
    P.mode = ppg::MODE_EXPORT_GRID; P.grid_out = grid_out;
    return backend_launch(h, ppg::MODE_EXPORT_GRID, P, stream);
}

int32_t ppg_lds_bytes(const ppg_handle *h) { return h ? h->base.lds_bytes : 0; }

// bfloat16 rows on the four-wave cooperative kernel: its 64-register build when eight workgroups fit a CU's LDS
static bool ppg_coop_high_occupancy(const ppg_handle *h, int coop_e) {
    return !h->gen2 && h->cfg.obs_dtype >= 2 && ppg_coop_lds_bytes_of(h->coop, coop_e) * 8 <= 160 * 1024 && !getenv("PPG_COOP_NO_HIGH_OCCUPANCY");
}

const char *ppg_step_kernel_name(ppg_handle *h) {
    if (!h) return "";
    const ppg_wave_plan_t wp = h->plan;
    if (wp.coop_e > 0) {   // ppgc_step_q<NQ> (4 waves) / ppgc8_ / ppgc16_
        if (h->gen2 && h->cfg2.walls) { snprintf(h->kernel_name, sizeof h->kernel_name, "ppgc3_step_q%d", h->nq); return h->kernel_name; }
        const bool three = !ppg_planned_step_params(h).ch0_map;
        snprintf(h->kernel_name, sizeof h->kernel_name, "ppgc%s%s_step_q%d", three ? "m" : "",
                 h->gen2 ? "2" : wp.nw == 8 ? "8" : wp.nw == 16 ? "16" : wp.nw == 6 ? "6" : ppg_coop_high_occupancy(h, wp.coop_e) ? "h" : "", h->nq);
        return h->kernel_name;
    }
    const bool walls = h->gen2 && h->cfg2.walls, fast = h->base.nch_p <= 2 && h->base.nch_q <= 3 && !walls && !h->drive;
    const char *family = h->drive ? "4" : walls ? "3" : h->gen2 ? "2" : "";
    // ppg[w|w8|wp]<family>_step... the names of ppg_kernel_list.h: ppgw_step / ppgw8_step / ppgwp_step, ppgw2_step / ppgw28_step, ppgw3_step, ppgw4_step
    const char *waves = wp.nw == 1 ? "" : wp.nw == 2 ? "wp" : wp.nw == 8 ? "w8" : wp.nw == 16 ? "w16" : "w";
    char fam[8];
    if (wp.nw == 8 && h->gen2 && !walls) snprintf(fam, sizeof fam, "w28");
    else snprintf(fam, sizeof fam, "%s%s", wp.nw == 8 ? "w8" : waves, family);
    snprintf(h->kernel_name, sizeof h->kernel_name, "ppg%s_step_%sq%d%s", fam, h->cfg.kickback && wp.nw == 1 ? "kick_" : "", h->nq,
             (fast || walls || h->drive) ? "" : "g");
    return h->kernel_name;
}

uint64_t ppg_state_bytes(const ppg_handle *h) {
    if (!h) return 0;
    ppg_state_field f[16];
    const int n = ppg_state_fields(h, f);
    uint64_t tot = sizeof(ppg_state_header);
    for (int i = 0; i < n; ++i) tot += (f[i].bytes + 7) / 8 * 8;
    return tot;
}

static void ppg_fill_state_header(const ppg_handle *h, ppg_state_header &H) {
    memset(&H, 0, sizeof H);
    H.magic = PPG_STATE_MAGIC; H.version = PPG_STATE_VERSION; H.bytes = (uint32_t)ppg_state_bytes(h);
    H.gen2 = h->gen2 ? 1u : 0u; H.walls = (h->gen2 && h->cfg2.walls) ? 1u : 0u;
    H.grid_size = (uint32_t)h->base.G; H.pred_capacity = (uint32_t)h->base.cap_pred; H.prey_capacity = (uint32_t)h->base.cap_prey;
    H.grass_capacity = (uint32_t)h->base.cap_grass; H.n_wall_words = (uint32_t)h->base.n_wall_words;
}

int ppg_export_state(ppg_handle *h, int32_t env, void *blob, uint64_t *size, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!size) return ppg_fail(h, PPG_EINVAL, "size is NULL");
    const uint64_t need = ppg_state_bytes(h);
    if (!blob) { *size = need; return PPG_OK; }
    if (env < 0 || env >= h->batch) return ppg_fail(h, PPG_EINVAL, "env %d not in 0..%d", env, h->batch - 1);
    if (*size < need) return ppg_fail(h, PPG_EINVAL, "blob holds %llu bytes, the image needs %llu", (unsigned long long)*size, (unsigned long long)need);
    ppg_state_header H;
    ppg_fill_state_header(h, H);
    memcpy(blob, &H, sizeof H);
    ppg_state_field f[16];
    const int n = ppg_state_fields(h, f);
    unsigned char *dst = (unsigned char *)blob + sizeof H;
    for (int i = 0; i < n; ++i) {
        const int rc = backend_copy(h, dst, (const unsigned char *)f[i].base + (size_t)env * f[i].bytes, f[i].bytes, false, stream);
        if (rc != PPG_OK) return rc;
        dst += (f[i].bytes + 7) / 8 * 8;
    }
    *size = need;
    return backend_sync(h, stream);
}

int ppg_import_state(ppg_handle *h, int32_t env, const void *blob, uint64_t size, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!blob) return ppg_fail(h, PPG_EINVAL, "blob is NULL");
    if (env < 0 || env >= h->batch) return ppg_fail(h, PPG_EINVAL, "env %d not in 0..%d", env, h->batch - 1);
    ppg_state_header want, got;
    ppg_fill_state_header(h, want);
    if (size < sizeof got) return ppg_fail(h, PPG_EINVAL, "blob of %llu bytes is shorter than a header", (unsigned long long)size);
    memcpy(&got, blob, sizeof got);
    if (got.magic != PPG_STATE_MAGIC || got.version != PPG_STATE_VERSION)
        return ppg_fail(h, PPG_EINVAL, "not a ppg state image (magic 0x%x, version %u)", got.magic, got.version);
    if (memcmp(&got, &want, sizeof got) != 0 || size < want.bytes)
        return ppg_fail(h, PPG_EINVAL, "state image was taken from a handle with another geometry (grid %u, rows %u+%u, grass %u, gen2 %u, walls %u)",
                        got.grid_size, got.pred_capacity, got.prey_capacity, got.grass_capacity, got.gen2, got.walls);
    ppg_state_field f[16];
    const int n = ppg_state_fields(h, f);
    const unsigned char *src = (const unsigned char *)blob + sizeof got;
    for (int i = 0; i < n; ++i) {
        const int rc = backend_copy(h, (unsigned char *)f[i].base + (size_t)env * f[i].bytes, src, f[i].bytes, true, stream);
        if (rc != PPG_OK) return rc;
        src += (f[i].bytes + 7) / 8 * 8;
    }
    {
        const int rc = backend_sync(h, stream);   // the blob may be freed by the caller as soon as this returns
        if (rc != PPG_OK) return rc;
    }
    if (h->gen2 && h->cfg2.walls && h->base.vis_masks) return ppg_walls_changed(h, stream);   // the image carried a wall bitmap
    return PPG_OK;
}

static int ppg_pack_geometry(const ppg_handle *h, uint32_t flags, int &blk_p, int &blk_q, int &src_elem, int &dst_elem) {
    const ppg::KParams &P = h->base;
    const bool drive = h->drive != 0;
    const int channels = (h->gen2 && h->cfg2.walls && h->cfg2.include_visibility_channel) ? 5 : 4;
    blk_p = (drive ? 4 + P.n_drive[0] : channels) * P.Rp * P.Rp;
    blk_q = (drive ? 4 + P.n_drive[1] : channels) * P.Rq * P.Rq;
    if (flags & PPG_PACK_NO_OBS) blk_p = blk_q = 0;   // an image without observation sections
    src_elem = P.obs_f32 == 2 ? 2 : P.obs_f32 ? 4 : 8;
    dst_elem = ((flags & PPG_PACK_F32) && src_elem == 8) ? 4 : src_elem;
    return PPG_OK;
}

uint64_t ppg_pack_bytes(const ppg_handle *h, int32_t n_envs, int64_t n_pred_rows, int64_t n_prey_rows, uint32_t flags) {
    if (!h || n_envs < 0 || n_pred_rows < 0 || n_prey_rows < 0) return 0;
    int bp, bq, se, de;
    ppg_pack_geometry(h, flags, bp, bq, se, de);
    return ppg::pack_layout((uint64_t)n_envs, (uint64_t)n_pred_rows, (uint64_t)n_prey_rows, (uint64_t)bp, (uint64_t)bq, (uint64_t)de).total;
}

int ppg_pack(ppg_handle *const *handles, int32_t n, void *out, uint64_t capacity, uint32_t flags, void *stream) {
    if (!handles || n < 1 || !handles[0]) return PPG_EINVAL;
    ppg_handle *h0 = handles[0];
    if (n > PPG_PACK_MAX_HANDLES) return ppg_fail(h0, PPG_EINVAL, "ppg_pack takes at most %d handles", PPG_PACK_MAX_HANDLES);
    if (!out || ((uintptr_t)out & 15u)) return ppg_fail(h0, PPG_EINVAL, "out must be a 16-byte aligned device pointer");
    if (flags & ~(PPG_PACK_F32 | PPG_PACK_NO_OBS)) return ppg_fail(h0, PPG_EINVAL, "unknown pack flags 0x%x", flags);
    ppg::PackParams K;
    memset(&K, 0, sizeof K);
    ppg_pack_geometry(h0, flags, K.blk_pred, K.blk_prey, K.src_elem, K.dst_elem);
    K.S = h0->base.S; K.cap_pred = h0->base.cap_pred; K.cap_prey = h0->base.cap_prey;
    K.n_handles = n;
    int total = 0;
    for (int k = 0; k < n; ++k) {
        const ppg_handle *h = handles[k];
        if (!h) return ppg_fail(h0, PPG_EINVAL, "handle %d is NULL", k);
        int bp, bq, se, de;
        ppg_pack_geometry(h, flags, bp, bq, se, de);
        if (bp != K.blk_pred || bq != K.blk_prey || se != K.src_elem || h->base.S != K.S || h->base.cap_pred != K.cap_pred || h->device != h0->device)
            return ppg_fail(h0, PPG_EINVAL, "handle %d has another geometry / device than handle 0", k);
        K.env_base[k] = total;
        total += h->batch;
        K.env_state[k] = h->bufs.env_state; K.row_id[k] = h->bufs.row_id; K.row_reward[k] = h->bufs.row_reward;
        K.row_flags[k] = h->bufs.row_flags;
        K.obs_pred[k] = (const unsigned char *)h->bufs.obs_pred; K.obs_prey[k] = (const unsigned char *)h->bufs.obs_prey;
    }
    for (int k = n; k <= PPG_PACK_MAX_HANDLES; ++k) K.env_base[k] = total;
    K.n_envs = total;
    K.capacity = capacity;
    K.out = (unsigned char *)out;
    // the header, the env words and the row offsets are always written: the caller learns the size an overflowing image needs
    const uint64_t fixed = ppg::pack_layout((uint64_t)total, 0, 0, 0, 0, 4).id_p;
    if (capacity < fixed) return ppg_fail(h0, PPG_EINVAL, "capacity %llu is below the fixed part of the image (%llu bytes for %d envs)",
                                          (unsigned long long)capacity, (unsigned long long)fixed, total);
    return backend_pack(h0, K, stream);
}

static void ppg_fetch_geometry(const ppg_handle *h, uint32_t &rec, uint32_t &bp, uint32_t &bq) {
    ppg_state_field f[16];
    const int n = ppg_state_fields(h, f);
    uint64_t r = 0;
    for (int i = 0; i < n; ++i) r += (f[i].bytes + 7) / 8 * 8;
    rec = (uint32_t)((r + 15) / 16 * 16);
    int blk_p, blk_q, se, de;
    ppg_pack_geometry(h, 0, blk_p, blk_q, se, de);
    bp = (uint32_t)(blk_p * se); bq = (uint32_t)(blk_q * se);
}

uint64_t ppg_fetch_bytes(const ppg_handle *h, int32_t n_envs, int64_t n_pred_rows, int64_t n_prey_rows) {
    if (!h || n_envs < 0 || n_pred_rows < 0 || n_prey_rows < 0) return 0;
    uint32_t rec, bp, bq;
    ppg_fetch_geometry(h, rec, bp, bq);
    return sizeof(ppg_fetch_header) + (uint64_t)n_envs * rec + (uint64_t)n_pred_rows * bp + (uint64_t)n_prey_rows * bq + (uint64_t)n_envs * 32;
}

int ppg_fetch(ppg_handle *h, int32_t env0, int32_t n_envs, void *host, uint64_t capacity, void *stream) {
    if (!h) return PPG_EINVAL;
    if (!host || ((uintptr_t)host & 15u)) return ppg_fail(h, PPG_EINVAL, "host must be a 16-byte aligned host pointer");
    if (env0 < 0 || n_envs < 1 || env0 + n_envs > h->batch) return ppg_fail(h, PPG_EINVAL, "envs [%d, %d) not in 0..%d", env0, env0 + n_envs, h->batch);
    ppg::FetchParams K;
    memset(&K, 0, sizeof K);
    ppg_fetch_geometry(h, K.record_bytes, K.blk_pred_bytes, K.blk_prey_bytes);
    const uint64_t fixed = sizeof(ppg_fetch_header) + (uint64_t)n_envs * K.record_bytes;
    if (capacity < fixed) return ppg_fail(h, PPG_EINVAL, "capacity %llu is below the fixed part of the image (%llu bytes for %d envs)",
                                          (unsigned long long)capacity, (unsigned long long)fixed, n_envs);
    ppg_state_field f[16];
    K.n_fields = ppg_state_fields(h, f);
    for (int i = 0; i < K.n_fields; ++i) { K.field[i] = (const unsigned char *)f[i].base; K.field_bytes[i] = (uint32_t)f[i].bytes; }
    K.env0 = env0; K.n_envs = n_envs;
    K.env_state = h->bufs.env_state;
    K.obs_pred = (const unsigned char *)h->bufs.obs_pred; K.obs_prey = (const unsigned char *)h->bufs.obs_prey;
    K.cap_pred = h->base.cap_pred; K.cap_prey = h->base.cap_prey;
    K.capacity = capacity;
    if (h->fetch_cap < capacity) {   // the staging buffer grows with the largest image asked for
        if (h->fetch_dev) { const int rc = backend_sync(h, stream); if (rc != PPG_OK) return rc; backend_free(h, h->fetch_dev); }
        h->fetch_dev = nullptr; h->fetch_cap = 0;
        const int rc = backend_alloc(h, (void **)&h->fetch_dev, (size_t)capacity);
        if (rc != PPG_OK) return rc;
        h->fetch_cap = capacity;
    }
    K.out = h->fetch_dev;
    int rc = backend_fetch(h, K, stream);
    if (rc != PPG_OK) return rc;
    // ONE transfer in the common case: as many bytes as the last image took plus a margin (births add rows); the header then says
    // whether a second transfer has to bring the rest
    uint64_t first = h->fetch_hint ? h->fetch_hint : capacity;
    if (first < fixed) first = fixed;
    if (first > capacity) first = capacity;
    rc = backend_copy(h, host, h->fetch_dev, (size_t)first, false, stream);
    if (rc == PPG_OK) rc = backend_sync(h, stream);
    if (rc != PPG_OK) return rc;
    const ppg_fetch_header *H = (const ppg_fetch_header *)host;
    if (H->magic != PPG_FETCH_MAGIC) return ppg_fail(h, PPG_EHIP, "ppg_fetch: the image has no header (magic 0x%x)", H->magic);
    if (!H->overflow && H->bytes_used > first) {
        rc = backend_copy(h, (unsigned char *)host + first, h->fetch_dev + first, (size_t)(H->bytes_used - first), false, stream);
        if (rc == PPG_OK) rc = backend_sync(h, stream);
        if (rc != PPG_OK) return rc;
    }
    h->fetch_hint = H->bytes_used + H->bytes_used / 4 + 4096;
    return PPG_OK;
}

#ifdef PPG_PROFILE_PHASES
// diagnostic build only: device buffer [B,16] of shader-clock stamps (see PPG_STAMP)
int ppg_debug_set_profile_buffer(ppg_handle *h, unsigned long long *dev) { h->prof_dev = dev; return PPG_OK; }
#endif

const char *ppg_last_error(const ppg_handle *h) { return h ? h->err : g_ppg_create_error; }

}  // extern "C"
