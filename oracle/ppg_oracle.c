/*
 * ppg_oracle.c -- CPU restatement of the reference PredPreyGrass environment.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ppg_oracle.h).  Parity status:
 * PINNED against the reference through tests/golden/ (see the header).
 *
 * The restatement is deliberately literal: a dense (4,G,G) float64 grid,
 * insertion-ordered dictionaries, Python's list.sort() on the agent-id strings.
 * It shares no data structure with the HIP path (sparse slot tables, ownership
 * bits), so agreement between the two is evidence, not tautology.
 *
 * "BASE:n" = line n of
 * /root/reference/predpreygrass/non_evolutionary/base_environment/predpreygrass_rllib_env.py
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -o libppg_oracle.so ppg_oracle.c
 * (-ffp-contract=off: every energy update must be a separately rounded IEEE
 * double operation, as in CPython.)
 */
#include "ppg_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* state                                                              */
/* ------------------------------------------------------------------ */

struct ppo_env {
    ppo_config c;
    int G;
    double *grid; /* grid_world_state, index [ch][x][y], BASE:118-124 */

    /* self.agents (list of id strings) as (type,id) pairs, BASE:73,143 */
    int n_agents;
    int *ag_type, *ag_id;

    /* self.agent_positions: insertion-ordered dict, BASE:111,146.  Entries are
     * appended at insertion, flagged absent on `del`; ids are never reused so
     * an id is inserted at most once per episode. */
    int n_entries;
    int *ent_type, *ent_id, *ent_x, *ent_y;
    char *ent_present;
    int *ent_index[2]; /* [type][id] -> entry index, or -1 */

    double *energy[2];   /* self.agent_energies */
    double *cumrew[2];   /* self.cumulative_rewards */
    double *e_before[2]; /* dense variants: energy_before (DENSE:242-245) */
    char *has_before[2];
    double *bonus[2];    /* dense_additive: reproduction_bonus */
    int *parent[2];      /* kickback variant: self.agent_parent (id of the same-type parent, -1 = none) */
    char *just_ate[2];   /* self.agents_just_ate */

    int n_grass;
    int *grass_x, *grass_y; /* self.grass_positions (dict in grass_k order) */
    double *grass_e;        /* self.grass_energies */

    int n_pending; /* self._pending_removal */
    int *pend_type, *pend_id;

    int next_idx[2];  /* _next_predator_idx / _next_prey_idx */
    int current_step;
    int cur_num[2];   /* current_num_predators / current_num_prey */

    uint64_t seed;
    uint32_t episode;

    /* the four dicts step() builds, keyed by agent id */
    char *has_obs[2], *has_rew[2], *has_term[2], *has_trunc[2];
    int *obs_at[2];
    double *rew[2];
    char *term[2], *trunc[2];

    double *arena; /* every _get_observation() result of this call */
    size_t arena_len, arena_cap;

    ppo_record *rec;
    int rec_cap;

    /* rollout helper state */
    int ro_done;
    int ro_started;
    int ro_n_live;          /* live agents in the order of the last returned dict */
    int *ro_live_type, *ro_live_id, *ro_live_row;
};

static int npos(const ppo_env *e, int type) {
    return type == PPO_PREDATOR ? e->c.n_possible_predators : e->c.n_possible_prey;
}
static int obs_range(const ppo_env *e, int type) {
    return type == PPO_PREDATOR ? e->c.predator_obs_range : e->c.prey_obs_range; /* BASE:515 */
}
static double *cell(ppo_env *e, int ch, int x, int y) {
    return &e->grid[((size_t)ch * e->G + x) * e->G + y];
}
static void agent_name(int type, int id, char *buf) {
    sprintf(buf, "%s_%d", type == PPO_PREDATOR ? "predator" : "prey", id); /* BASE:71-72 */
}

ppo_env *ppo_create(const ppo_config *cfg) {
    if (cfg->num_obs_channels != 4 || cfg->grid_size < 1) return NULL;
    ppo_env *e = (ppo_env *)calloc(1, sizeof(*e));
    e->c = *cfg;
    e->G = cfg->grid_size;
    int tot = cfg->n_possible_predators + cfg->n_possible_prey;
    if (cfg->n_initial_active_predator + cfg->n_initial_active_prey > tot) tot =
        cfg->n_initial_active_predator + cfg->n_initial_active_prey;
    e->grid = (double *)calloc((size_t)4 * e->G * e->G, sizeof(double));
    e->ag_type = (int *)calloc(tot + 8, sizeof(int));
    e->ag_id = (int *)calloc(tot + 8, sizeof(int));
    e->ent_type = (int *)calloc(tot + 8, sizeof(int));
    e->ent_id = (int *)calloc(tot + 8, sizeof(int));
    e->ent_x = (int *)calloc(tot + 8, sizeof(int));
    e->ent_y = (int *)calloc(tot + 8, sizeof(int));
    e->ent_present = (char *)calloc(tot + 8, 1);
    e->pend_type = (int *)calloc(tot + 8, sizeof(int));
    e->pend_id = (int *)calloc(tot + 8, sizeof(int));
    for (int t = 0; t < 2; ++t) {
        int n = npos(e, t);
        int ninit = t == PPO_PREDATOR ? cfg->n_initial_active_predator : cfg->n_initial_active_prey;
        if (ninit > n) n = ninit;
        n += 1;
        e->ent_index[t] = (int *)malloc(n * sizeof(int));
        e->energy[t] = (double *)calloc(n, sizeof(double));
        e->cumrew[t] = (double *)calloc(n, sizeof(double));
        e->e_before[t] = (double *)calloc(n, sizeof(double));
        e->has_before[t] = (char *)calloc(n, 1);
        e->bonus[t] = (double *)calloc(n, sizeof(double));
        e->parent[t] = (int *)malloc(n * sizeof(int));
        for (int i = 0; i < n; ++i) e->parent[t][i] = -1;
        e->just_ate[t] = (char *)calloc(n, 1);
        e->has_obs[t] = (char *)calloc(n, 1);
        e->has_rew[t] = (char *)calloc(n, 1);
        e->has_term[t] = (char *)calloc(n, 1);
        e->has_trunc[t] = (char *)calloc(n, 1);
        e->obs_at[t] = (int *)calloc(n, sizeof(int));
        e->rew[t] = (double *)calloc(n, sizeof(double));
        e->term[t] = (char *)calloc(n, 1);
        e->trunc[t] = (char *)calloc(n, 1);
        for (int i = 0; i < n; ++i) e->ent_index[t][i] = -1;
    }
    e->n_grass = cfg->initial_num_grass;
    e->grass_x = (int *)calloc(e->n_grass + 1, sizeof(int));
    e->grass_y = (int *)calloc(e->n_grass + 1, sizeof(int));
    e->grass_e = (double *)calloc(e->n_grass + 1, sizeof(double));
    e->ro_live_type = (int *)calloc(tot + 8, sizeof(int));
    e->ro_live_id = (int *)calloc(tot + 8, sizeof(int));
    e->ro_live_row = (int *)calloc(tot + 8, sizeof(int));
    e->rec_cap = tot + 8;
    e->rec = (ppo_record *)calloc(e->rec_cap, sizeof(ppo_record));
    e->arena_cap = 1 << 16;
    e->arena = (double *)malloc(e->arena_cap * sizeof(double));
    e->ro_done = 1;
    return e;
}

void ppo_destroy(ppo_env *e) {
    if (!e) return;
    free(e->grid); free(e->ag_type); free(e->ag_id);
    free(e->ent_type); free(e->ent_id); free(e->ent_x); free(e->ent_y); free(e->ent_present);
    free(e->pend_type); free(e->pend_id);
    for (int t = 0; t < 2; ++t) {
        free(e->ent_index[t]); free(e->energy[t]); free(e->cumrew[t]); free(e->just_ate[t]);
        free(e->e_before[t]); free(e->has_before[t]); free(e->bonus[t]); free(e->parent[t]);
        free(e->has_obs[t]); free(e->has_rew[t]); free(e->has_term[t]); free(e->has_trunc[t]);
        free(e->obs_at[t]); free(e->rew[t]); free(e->term[t]); free(e->trunc[t]);
    }
    free(e->grass_x); free(e->grass_y); free(e->grass_e);
    free(e->ro_live_type); free(e->ro_live_id); free(e->ro_live_row);
    free(e->rec); free(e->arena);
    free(e);
}

void ppo_set_seed(ppo_env *e, uint64_t seed, uint32_t episode) {
    e->seed = seed;
    e->episode = episode;
}

/* ------------------------------------------------------------------ */
/* observation: BASE:511-539                                          */
/* ------------------------------------------------------------------ */

static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* _get_observation(agent) for an agent standing at (xp,yp). */
static void observe_at(const ppo_env *e, int xp, int yp, int R, double *obs) {
    const int G = e->G;
    /* _obs_clip, BASE:528-539 */
    int off = (R - 1) / 2;
    int xld = xp - off, xhd = xp + off;
    int yld = yp - off, yhd = yp + off;
    int xlo = clipi(xld, 0, G - 1), xhi = clipi(xhd, 0, G - 1);
    int ylo = clipi(yld, 0, G - 1), yhi = clipi(yhd, 0, G - 1);
    int xolo = abs(clipi(xld, -off, 0)), yolo = abs(clipi(yld, -off, 0));
    int xohi = xolo + (xhi - xlo), yohi = yolo + (yhi - ylo);
    /* returned as half-open ranges: xlo,xhi+1, ... */
    xhi += 1; yhi += 1; xohi += 1; yohi += 1;
    (void)xhi; (void)yhi;
    /* BASE:518-524 */
    for (int i = 0; i < 4 * R * R; ++i) obs[i] = 0.0;
    for (int i = 0; i < R * R; ++i) obs[i] = 1.0; /* observation[0].fill(1) */
    for (int i = xolo; i < xohi; ++i)
        for (int j = yolo; j < yohi; ++j) {
            obs[(0 * R + i) * R + j] = 0.0;
            int gx = xlo + (i - xolo), gy = ylo + (j - yolo);
            for (int ch = 1; ch < 4; ++ch)
                obs[((size_t)ch * R + i) * R + j] = e->grid[((size_t)ch * G + gx) * G + gy];
        }
}

/* np.sum of n contiguous float64 values: numpy's pairwise summation (numpy/core/src/umath/loops_utils.h.src,
 * pairwise_sum: plain loop below 8 elements, eight interleaved accumulators up to 128, halves above).  The drive
 * features sum a whole (R,R) channel of the observation (DRV:601-608), so the order of the additions matters. */
static double np_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum(a, n2) + np_sum(a + n2, n - n2);
    }
}

static int n_channels(const ppo_env *e, int type) { return 4 + e->c.n_drive[type]; }

/* _safe_clip01, DRV:612-615 */
static double safe_clip01(double v) {
    if (!(v - v == 0.0)) return 0.0;   /* not finite */
    return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
}

/* _get_drive_features / _get_drive_feature (DRV:577-610) appended to an observation whose world channels are filled */
static void add_drive_channels(const ppo_env *e, int type, double energy, int R, double *obs) {
    const ppo_config *c = &e->c;
    for (int k = 0; k < c->n_drive[type]; ++k) {
        double v = 0.0;
        switch (c->drive_kind[type][k]) {
            case PPO_DRIVE_HUNGER: v = safe_clip01(1.0 - energy / c->hunger_safe_energy[type]); break;
            case PPO_DRIVE_REPRO:
                v = safe_clip01(energy / (type == PPO_PREDATOR ? c->predator_creation_energy_threshold
                                                               : c->prey_creation_energy_threshold));
                break;
            case PPO_DRIVE_PREY_OPP: v = safe_clip01(np_sum(obs + 2 * R * R, R * R) / c->prey_opportunity_normalizer); break;
            case PPO_DRIVE_PRED_DANGER: v = safe_clip01(np_sum(obs + 1 * R * R, R * R) / c->predator_danger_normalizer); break;
            case PPO_DRIVE_GRASS_OPP: v = safe_clip01(np_sum(obs + 3 * R * R, R * R) / c->grass_opportunity_normalizer); break;
        }
        for (int i = 0; i < R * R; ++i) obs[(size_t)(4 + k) * R * R + i] = v;   /* DRV:566-569 */
    }
}

static int entry_of(const ppo_env *e, int type, int id) {
    if (id < 0 || id > npos(e, type)) return -1;
    int k = e->ent_index[type][id];
    if (k < 0 || !e->ent_present[k]) return -1;
    return k;
}

int ppo_observe(const ppo_env *e, int32_t type, int32_t id, double *dst) {
    int k = entry_of(e, type, id);
    if (k < 0) return -1;
    observe_at(e, e->ent_x[k], e->ent_y[k], obs_range(e, type), dst);
    add_drive_channels(e, type, e->energy[type][id], obs_range(e, type), dst);
    return 0;
}

/* observations[agent] = self._get_observation(agent) */
static void put_obs(ppo_env *e, int type, int id) {
    int R = obs_range(e, type);
    size_t len = (size_t)n_channels(e, type) * R * R;
    if (e->arena_len + len > e->arena_cap) {
        while (e->arena_len + len > e->arena_cap) e->arena_cap *= 2;
        e->arena = (double *)realloc(e->arena, e->arena_cap * sizeof(double));
    }
    int k = entry_of(e, type, id);
    observe_at(e, e->ent_x[k], e->ent_y[k], R, e->arena + e->arena_len);
    add_drive_channels(e, type, e->energy[type][id], R, e->arena + e->arena_len);
    e->obs_at[type][id] = (int)e->arena_len;
    e->has_obs[type][id] = 1;
    e->arena_len += len;
}

static void clear_call_dicts(ppo_env *e) {
    for (int t = 0; t < 2; ++t) {
        int n = npos(e, t) + 1;
        memset(e->has_obs[t], 0, n); memset(e->has_rew[t], 0, n);
        memset(e->has_term[t], 0, n); memset(e->has_trunc[t], 0, n);
    }
    e->arena_len = 0;
}

/* ------------------------------------------------------------------ */
/* dict helpers                                                       */
/* ------------------------------------------------------------------ */

static void positions_insert(ppo_env *e, int type, int id, int x, int y) {
    int k = e->n_entries++;
    e->ent_type[k] = type; e->ent_id[k] = id; e->ent_x[k] = x; e->ent_y[k] = y;
    e->ent_present[k] = 1;
    e->ent_index[type][id] = k;
}
static void positions_delete(ppo_env *e, int type, int id) {
    int k = e->ent_index[type][id];
    e->ent_present[k] = 0;
}
static int in_pending(const ppo_env *e, int type, int id) {
    for (int i = 0; i < e->n_pending; ++i)
        if (e->pend_type[i] == type && e->pend_id[i] == id) return 1;
    return 0;
}

static int cmp_names(const void *a, const void *b) {
    const int *pa = (const int *)a, *pb = (const int *)b;
    char na[32], nb[32];
    agent_name(pa[0], pa[1], na);
    agent_name(pb[0], pb[1], nb);
    return strcmp(na, nb);
}
/* self.agents.sort(), BASE:468 -- lexicographic on the id strings */
static void agents_sort(ppo_env *e) {
    int n = e->n_agents;
    int *tmp = (int *)malloc((size_t)n * 2 * sizeof(int) + 8);
    for (int i = 0; i < n; ++i) { tmp[2 * i] = e->ag_type[i]; tmp[2 * i + 1] = e->ag_id[i]; }
    qsort(tmp, n, 2 * sizeof(int), cmp_names);
    for (int i = 0; i < n; ++i) { e->ag_type[i] = tmp[2 * i]; e->ag_id[i] = tmp[2 * i + 1]; }
    free(tmp);
}

/* Build the returned dicts: filter by self.agents order, BASE:459-462 */
static int emit_records(ppo_env *e, ppo_step_out *out) {
    int n = 0;
    for (int i = 0; i < e->n_agents; ++i) {
        int t = e->ag_type[i], id = e->ag_id[i];
        /* In the reference each of the four dicts is filtered on its own; every
         * id in self.agents has an entry in all four on every path (asserted). */
        if (!e->has_obs[t][id] || !e->has_rew[t][id] || !e->has_term[t][id] || !e->has_trunc[t][id])
            return -9;
        ppo_record *r = &e->rec[n++];
        int R = obs_range(e, t);
        r->type = t; r->id = id;
        r->reward = e->rew[t][id];
        r->terminated = e->term[t][id];
        r->truncated = e->trunc[t][id];
        r->obs_offset = e->obs_at[t][id];
        r->obs_len = n_channels(e, t) * R * R;
    }
    out->n_records = n;
    out->records = e->rec;
    out->obs = e->arena;
    return 0;
}

/* ------------------------------------------------------------------ */
/* reset: BASE:129-217 with the placement supplied by the caller      */
/* ------------------------------------------------------------------ */

int ppo_reset_from_placement(ppo_env *e, const int32_t *pred_xy, const int32_t *prey_xy,
                             const int32_t *grass_xy, ppo_step_out *out) {
    const ppo_config *c = &e->c;
    const int G = e->G;
    if (c->n_initial_active_predator + c->n_initial_active_prey + c->initial_num_grass > G * G)
        return -4; /* ValueError, BASE:167-168 */
    e->current_step = 0;                                              /* BASE:134 */
    memset(e->grid, 0, (size_t)4 * G * G * sizeof(double));           /* BASE:138 */
    /* BASE:143-147 */
    e->n_agents = 0;
    for (int i = 0; i < c->n_initial_active_predator; ++i) {
        e->ag_type[e->n_agents] = PPO_PREDATOR; e->ag_id[e->n_agents++] = i;
    }
    for (int j = 0; j < c->n_initial_active_prey; ++j) {
        e->ag_type[e->n_agents] = PPO_PREY; e->ag_id[e->n_agents++] = j;
    }
    e->n_entries = 0;
    for (int t = 0; t < 2; ++t) {
        int n = npos(e, t) + 1;
        for (int i = 0; i < n; ++i) { e->ent_index[t][i] = -1; e->parent[t][i] = -1; }  /* KICK:178 */
        memset(e->just_ate[t], 0, n);
    }
    for (int i = 0; i < e->n_agents; ++i) e->cumrew[e->ag_type[i]][e->ag_id[i]] = 0; /* BASE:150 */
    e->n_pending = 0;                                                 /* BASE:152 */
    e->next_idx[PPO_PREDATOR] = c->n_initial_active_predator;         /* BASE:153 */
    e->next_idx[PPO_PREY] = c->n_initial_active_prey;                 /* BASE:154 */
    /* BASE:190-200 */
    for (int i = 0; i < e->n_agents; ++i) {
        int t = e->ag_type[i], id = e->ag_id[i];
        const int32_t *xy = t == PPO_PREDATOR ? pred_xy : prey_xy;
        int x = xy[2 * id], y = xy[2 * id + 1];
        if (x < 0 || x >= G || y < 0 || y >= G) return -5;
        positions_insert(e, t, id, x, y);
        double e0 = t == PPO_PREDATOR ? c->initial_energy_predator : c->initial_energy_prey;
        e->energy[t][id] = e0;
        *cell(e, t == PPO_PREDATOR ? 1 : 2, x, y) = e0;
    }
    /* BASE:203-208 */
    for (int k = 0; k < e->n_grass; ++k) {
        int x = grass_xy[2 * k], y = grass_xy[2 * k + 1];
        if (x < 0 || x >= G || y < 0 || y >= G) return -5;
        e->grass_x[k] = x; e->grass_y[k] = y;
        e->grass_e[k] = c->initial_energy_grass;
        *cell(e, 3, x, y) = c->initial_energy_grass;
    }
    e->cur_num[PPO_PREY] = c->n_initial_active_prey;                  /* BASE:210 */
    e->cur_num[PPO_PREDATOR] = c->n_initial_active_predator;          /* BASE:211 */
    /* BASE:215: observations for all agents; reset returns (observations, {}) */
    clear_call_dicts(e);
    for (int i = 0; i < e->n_agents; ++i) {
        int t = e->ag_type[i], id = e->ag_id[i];
        put_obs(e, t, id);
        e->rew[t][id] = 0.0; e->has_rew[t][id] = 1;   /* not part of reset's return; */
        e->term[t][id] = 0; e->has_term[t][id] = 1;   /* filled so records are uniform */
        e->trunc[t][id] = 0; e->has_trunc[t][id] = 1;
    }
    if (out) {
        out->terminated_all = 0; out->truncated_all = 0;
        out->fallback_spawns = 0; out->failed_spawns = 0;
        return emit_records(e, out);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Philox4x32-10 and the build-defined RNG contract                   */
/* ------------------------------------------------------------------ */

#define PPO_TAG_ACT 0x41435431u
#define PPO_TAG_RST 0x52535431u
#define PPO_TAG_SPW 0x53505731u

void ppo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static uint32_t philox_word(uint64_t seed, uint32_t tag, uint32_t a, uint32_t b, uint32_t c, uint32_t d, int w) {
    uint32_t ctr[4] = {a, b, c, d};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ tag};
    uint32_t out[4];
    ppo_philox4x32_10(ctr, key, out);
    return out[w];
}

/* Row `row` of type `type` lives in wave lane (row & 63) of register 0
 * (predators) or 1 + (row >> 6) (prey); one Philox call per lane yields the
 * words for registers 4q..4q+3. */
int32_t ppo_random_action(uint64_t seed, uint32_t episode, uint32_t step, int32_t type, int32_t row) {
    uint32_t lane = (uint32_t)row & 63u;
    uint32_t reg = type == PPO_PREDATOR ? 0u : 1u + ((uint32_t)row >> 6);
    uint32_t w = philox_word(seed, PPO_TAG_ACT, step, lane + 64u * (reg >> 2), 0u, episode, (int)(reg & 3u));
    return (int32_t)(((uint64_t)w * 9u) >> 32);
}

int ppo_reset_philox(ppo_env *e, uint64_t seed, uint32_t episode, ppo_step_out *out) {
    const ppo_config *c = &e->c;
    const int G = e->G, n = G * G;
    int K = c->n_initial_active_predator + c->n_initial_active_prey + c->initial_num_grass;
    if (K > n) return -4;
    ppo_set_seed(e, seed, episode);
    int *perm = (int *)malloc((size_t)n * sizeof(int));
    int32_t *xy = (int32_t *)malloc((size_t)2 * K * sizeof(int32_t) + 8);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < K; ++k) {
        uint32_t r = philox_word(seed, PPO_TAG_RST, (uint32_t)k >> 2, 0u, 0u, episode, k & 3);
        int j = k + (int)(((uint64_t)r * (uint32_t)(n - k)) >> 32);
        int a = perm[k], b = perm[j];
        perm[j] = a; perm[k] = b;
        xy[2 * k] = b / G; xy[2 * k + 1] = b % G;
    }
    int P = c->n_initial_active_predator, Q = c->n_initial_active_prey;
    int rc = ppo_reset_from_placement(e, xy, xy + 2 * P, xy + 2 * (P + Q), out);
    free(perm); free(xy);
    return rc;
}

/* ------------------------------------------------------------------ */
/* step: BASE:219-473                                                 */
/* ------------------------------------------------------------------ */

/* _find_available_spawn_position, BASE:738-766.  Returns 0 and (*nx,*ny), or
 * -1 when no cell is free.  *fallback is set when BASE:759-764 is reached. */
static int find_spawn(ppo_env *e, int x, int y, int child_type, int child_id, int *nx, int *ny, int *fallback) {
    const int G = e->G;
    static const int d[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}}; /* BASE:749 */
    for (int k = 0; k < 4; ++k) {
        int cx = x + d[k][0], cy = y + d[k][1];
        if (!(0 <= cx && cx < G && 0 <= cy && cy < G)) continue;      /* BASE:750 */
        int occ = 0;                                                  /* BASE:754 */
        for (int i = 0; i < e->n_entries; ++i)
            if (e->ent_present[i] && e->ent_x[i] == cx && e->ent_y[i] == cy) { occ = 1; break; }
        if (!occ) { *nx = cx; *ny = cy; return 0; }                   /* BASE:756-757 */
    }
    /* BASE:759-764: the reference draws from the unseeded global np.random
     * over an arbitrarily ordered set; there is nothing to be bit-exact with.
     * Build contract: k-th free cell in x-major order, k from Philox. */
    *fallback = 1;
    char *occ = (char *)calloc((size_t)G * G, 1);
    for (int i = 0; i < e->n_entries; ++i)
        if (e->ent_present[i]) occ[e->ent_x[i] * G + e->ent_y[i]] = 1;
    int nfree = 0;
    for (int i = 0; i < G * G; ++i) nfree += !occ[i];
    if (nfree == 0) { free(occ); return -1; }                         /* BASE:766 */
    uint32_t r = philox_word(e->seed, PPO_TAG_SPW, (uint32_t)e->current_step, (uint32_t)child_id,
                             (uint32_t)child_type, e->episode, 0);
    int k = (int)(((uint64_t)r * (uint32_t)nfree) >> 32);
    for (int i = 0; i < G * G; ++i)
        if (!occ[i] && k-- == 0) { *nx = i / G; *ny = i % G; break; }
    free(occ);
    return 0;
}

int ppo_step(ppo_env *e, int32_t n_act, const int32_t *act_type, const int32_t *act_id,
             const int32_t *act, ppo_step_out *out) {
    const ppo_config *c = &e->c;
    const int G = e->G;
    clear_call_dicts(e);                                              /* BASE:220 */
    out->fallback_spawns = 0; out->failed_spawns = 0;

    /* BASE:222-225 */
    for (int p = 0; p < e->n_pending; ++p)
        for (int i = 0; i < e->n_agents; ++i)
            if (e->ag_type[i] == e->pend_type[p] && e->ag_id[i] == e->pend_id[p]) {
                memmove(e->ag_type + i, e->ag_type + i + 1, (size_t)(e->n_agents - i - 1) * sizeof(int));
                memmove(e->ag_id + i, e->ag_id + i + 1, (size_t)(e->n_agents - i - 1) * sizeof(int));
                e->n_agents--;
                break;
            }
    e->n_pending = 0;

    /* BASE:228-238: truncation */
    if (e->current_step >= c->max_steps) {
        for (int i = 0; i < e->n_agents; ++i) {
            int t = e->ag_type[i], id = e->ag_id[i];
            put_obs(e, t, id);
            e->rew[t][id] = 0.0; e->has_rew[t][id] = 1;
            e->trunc[t][id] = 1; e->has_trunc[t][id] = 1;
            e->term[t][id] = 0; e->has_term[t][id] = 1;
        }
        out->truncated_all = 1; out->terminated_all = 0;
        return emit_records(e, out);
    }

    /* the reference would raise KeyError at BASE:246/249 (dead agent) or
     * BASE:502 (bad action) part-way through; report it before mutating. */
    for (int a = 0; a < n_act; ++a) {
        if (entry_of(e, act_type[a], act_id[a]) < 0) return -2;
        if (act[a] < 0 || act[a] > 8) return -3;
    }

    for (int t = 0; t < 2; ++t) memset(e->just_ate[t], 0, npos(e, t) + 1); /* BASE:241 */
    const int dense = c->reward_mode != 0;
    if (dense) { /* energy_before = dict(self.agent_energies), DENSE:242-245 */
        for (int t = 0; t < 2; ++t) { memset(e->has_before[t], 0, npos(e, t) + 1); }
        for (int i = 0; i < e->n_entries; ++i)
            if (e->ent_present[i]) {
                int t = e->ent_type[i], id = e->ent_id[i];
                e->e_before[t][id] = e->energy[t][id]; e->has_before[t][id] = 1; e->bonus[t][id] = 0.0;
            }
    }

    /* Step 1, BASE:244-250 */
    for (int a = 0; a < n_act; ++a) {
        int t = act_type[a], id = act_id[a];
        int k = e->ent_index[t][id];
        if (t == PPO_PREDATOR) {
            e->energy[t][id] -= c->energy_loss_per_step_predator;
            *cell(e, 1, e->ent_x[k], e->ent_y[k]) = e->energy[t][id];
        } else {
            e->energy[t][id] -= c->energy_loss_per_step_prey;
            *cell(e, 2, e->ent_x[k], e->ent_y[k]) = e->energy[t][id];
        }
    }
    /* BASE:252-256; seasonal variant: base_environment_seasonal/predpreygrass_rllib_env.py:224-234,268-271
     * (square wave on current_step; the product is rounded once, like the Python expression) */
    double gain = c->energy_gain_per_step_grass;
    if (c->season_length_steps > 0) {
        int phase = (e->current_step / c->season_length_steps) % 2;
        gain = c->energy_gain_per_step_grass * (phase == 0 ? c->season_high_multiplier : c->season_low_multiplier);
    }
    for (int g = 0; g < e->n_grass; ++g) {
        double v = e->grass_e[g] + gain;
        /* Python min(a, b): b if b < a else a */
        e->grass_e[g] = (c->initial_energy_grass < v) ? c->initial_energy_grass : v;
        *cell(e, 3, e->grass_x[g], e->grass_y[g]) = e->grass_e[g];
    }

    /* Step 2, BASE:259-276 */
    for (int a = 0; a < n_act; ++a) {
        int t = act_type[a], id = act_id[a];
        int k = entry_of(e, t, id);
        if (k < 0) continue;                                          /* BASE:260 */
        int ox = e->ent_x[k], oy = e->ent_y[k];
        /* _get_move, BASE:495-509 */
        int ch = t == PPO_PREDATOR ? 1 : 2;                           /* BASE:499 */
        int dx = act[a] / 3 - 1, dy = act[a] % 3 - 1;                 /* BASE:96-106 */
        int nx = clipi(ox + dx, 0, G - 1), ny = clipi(oy + dy, 0, G - 1); /* BASE:503-505 */
        if (*cell(e, ch, nx, ny) > 0) { nx = ox; ny = oy; }           /* BASE:506-507 */
        e->ent_x[k] = nx; e->ent_y[k] = ny;                           /* BASE:263 */
        /* move_cost is the constant 0, BASE:264-265,493 */
        *cell(e, ch, ox, oy) = 0;                                     /* BASE:268/272 */
        *cell(e, ch, nx, ny) = e->energy[t][id];                      /* BASE:269/273 */
    }

    /* Step 3, BASE:279-380 */
    for (int i = 0; i < e->n_agents; ++i) {
        int t = e->ag_type[i], id = e->ag_id[i];
        int k = entry_of(e, t, id);
        if (k < 0) continue;                                          /* BASE:281-282 */
        if (e->energy[t][id] <= 0) {                                  /* BASE:284 */
            put_obs(e, t, id);                                        /* BASE:287 */
            e->rew[t][id] = 0; e->has_rew[t][id] = 1;                 /* BASE:288 */
            if (dense) { /* DENSE:291-292 */
                e->rew[t][id] = e->energy[t][id] - e->e_before[t][id];
                e->cumrew[t][id] += e->rew[t][id];
            }
            e->term[t][id] = 1; e->has_term[t][id] = 1;
            e->trunc[t][id] = 0; e->has_trunc[t][id] = 1;
            e->cur_num[t] -= 1;
            *cell(e, t == PPO_PREDATOR ? 1 : 2, e->ent_x[k], e->ent_y[k]) = 0; /* BASE:293/297 */
            positions_delete(e, t, id);                               /* BASE:299-300 */
            e->parent[t][id] = -1;                                    /* KICK:325 */
            continue;
        } else if (t == PPO_PREDATOR) {
            int px = e->ent_x[k], py = e->ent_y[k];
            int caught = -1;                                          /* BASE:305-312 */
            for (int j = 0; j < e->n_entries; ++j)
                if (e->ent_present[j] && e->ent_type[j] == PPO_PREY && e->ent_x[j] == px && e->ent_y[j] == py) {
                    caught = j; break;
                }
            if (caught >= 0) {
                int cid = e->ent_id[caught];
                e->just_ate[t][id] = 1;                               /* BASE:319 */
                if (!dense) {
                    e->rew[t][id] = c->reward_predator_catch_prey; e->has_rew[t][id] = 1; /* BASE:322 */
                    e->cumrew[t][id] += e->rew[t][id];                /* BASE:323 */
                }
                e->energy[t][id] += e->energy[PPO_PREY][cid];         /* BASE:324 */
                *cell(e, 1, px, py) = e->energy[t][id];               /* BASE:325 */
                put_obs(e, PPO_PREY, cid);                            /* BASE:327 */
                e->rew[PPO_PREY][cid] = c->penalty_prey_caught; e->has_rew[PPO_PREY][cid] = 1;
                if (dense) e->rew[PPO_PREY][cid] = 0.0 - e->e_before[PPO_PREY][cid]; /* DENSE:328-329 */
                e->cumrew[PPO_PREY][cid] += e->rew[PPO_PREY][cid];    /* BASE:329 */
                e->term[PPO_PREY][cid] = 1; e->has_term[PPO_PREY][cid] = 1;   /* BASE:332 */
                e->trunc[PPO_PREY][cid] = 0; e->has_trunc[PPO_PREY][cid] = 1;
                e->cur_num[PPO_PREY] -= 1;
                *cell(e, 2, e->ent_x[caught], e->ent_y[caught]) = 0;  /* BASE:335 */
                positions_delete(e, PPO_PREY, cid);                   /* BASE:336-338 */
                e->parent[PPO_PREY][cid] = -1;                        /* KICK:364 */
            } else if (!dense) {
                e->rew[t][id] = c->reward_predator_step; e->has_rew[t][id] = 1; /* BASE:341 */
            }
            put_obs(e, t, id);                                        /* BASE:343 */
            if (!dense) e->cumrew[t][id] += e->rew[t][id];            /* BASE:344 */
            e->term[t][id] = 0; e->has_term[t][id] = 1;
            e->trunc[t][id] = 0; e->has_trunc[t][id] = 1;
        } else {
            if (!e->has_term[t][id] || !e->term[t][id]) {             /* BASE:348 */
                int px = e->ent_x[k], py = e->ent_y[k];
                int g = -1;                                           /* BASE:351-358 */
                for (int j = 0; j < e->n_grass; ++j)
                    if (e->grass_x[j] == px && e->grass_y[j] == py) { g = j; break; }
                if (g >= 0) {
                    e->just_ate[t][id] = 1;                           /* BASE:362 */
                    if (!dense) {
                        e->rew[t][id] = c->reward_prey_eat_grass; e->has_rew[t][id] = 1; /* BASE:365 */
                        e->cumrew[t][id] += e->rew[t][id];            /* BASE:366 */
                    }
                    e->energy[t][id] += e->grass_e[g];                /* BASE:367 */
                    *cell(e, 2, px, py) = e->energy[t][id];           /* BASE:368 */
                    *cell(e, 3, e->grass_x[g], e->grass_y[g]) = 0;    /* BASE:371 */
                    e->grass_e[g] = 0;                                /* BASE:372 */
                } else if (!dense) {
                    e->rew[t][id] = c->reward_prey_step; e->has_rew[t][id] = 1; /* BASE:375 */
                }
                put_obs(e, t, id);                                    /* BASE:377 */
                if (!dense) e->cumrew[t][id] += e->rew[t][id];        /* BASE:378 */
                e->term[t][id] = 0; e->has_term[t][id] = 1;
                e->trunc[t][id] = 0; e->has_trunc[t][id] = 1;
            }
        }
    }

    /* Step 4, BASE:383 */
    e->n_pending = 0;
    for (int i = 0; i < e->n_agents; ++i) {
        int t = e->ag_type[i], id = e->ag_id[i];
        if (e->has_term[t][id] && e->term[t][id]) {
            e->pend_type[e->n_pending] = t; e->pend_id[e->n_pending++] = id;
        }
    }

    /* Step 5, BASE:389-448 (iterates a copy of self.agents) */
    int n_before = e->n_agents;
    for (int i = 0; i < n_before; ++i) {
        int t = e->ag_type[i], id = e->ag_id[i];
        if (in_pending(e, t, id)) continue;                           /* BASE:390-391 */
        double thr = t == PPO_PREDATOR ? c->predator_creation_energy_threshold
                                       : c->prey_creation_energy_threshold;
        double e0 = t == PPO_PREDATOR ? c->initial_energy_predator : c->initial_energy_prey;
        if (e->energy[t][id] >= thr) {                                /* BASE:393/422 */
            if (e->next_idx[t] < npos(e, t)) {                        /* BASE:395/424 */
                int cid = e->next_idx[t];
                int k = e->ent_index[t][id];
                int nx = 0, ny = 0, fb = 0;
                int rc = find_spawn(e, e->ent_x[k], e->ent_y[k], t, cid, &nx, &ny, &fb); /* BASE:399-400 */
                out->fallback_spawns += fb;
                if (rc < 0) { out->failed_spawns += 1; continue; }    /* reference: TypeError */
                e->next_idx[t] += 1;                                  /* BASE:397/426 */
                e->ag_type[e->n_agents] = t; e->ag_id[e->n_agents++] = cid; /* BASE:398/427 */
                positions_insert(e, t, cid, nx, ny);                  /* BASE:401 */
                e->energy[t][cid] = e0;                               /* BASE:403 */
                e->energy[t][id] -= e0;                               /* BASE:404 */
                int ch = t == PPO_PREDATOR ? 1 : 2;
                *cell(e, ch, nx, ny) = e0;                            /* BASE:405 */
                *cell(e, ch, e->ent_x[k], e->ent_y[k]) = e->energy[t][id]; /* BASE:406 */
                e->cur_num[t] += 1;
                e->rew[t][cid] = 0; e->has_rew[t][cid] = 1;           /* BASE:408 */
                if (!dense) {
                    e->rew[t][id] = t == PPO_PREDATOR ? c->reproduction_reward_predator
                                                      : c->reproduction_reward_prey; /* BASE:409/438 */
                    e->has_rew[t][id] = 1;
                    e->cumrew[t][id] += e->rew[t][id];                /* BASE:411 */
                } else if (c->reward_mode == 2) {  /* dense_additive: reproduction_bonus[agent] */
                    e->bonus[t][id] = t == PPO_PREDATOR ? c->reproduction_reward_predator : c->reproduction_reward_prey;
                }
                e->cumrew[t][cid] = 0;                                /* BASE:410 */
                if (c->kickback) {
                    e->parent[t][cid] = id;                           /* KICK:434/475 */
                    int gp = e->parent[t][id];                        /* KICK:443 */
                    if (gp >= 0 && entry_of(e, t, gp) >= 0) {
                        double kb = t == PPO_PREDATOR ? c->kickback_reward_predator : c->kickback_reward_prey;
                        e->rew[t][gp] = (e->has_rew[t][gp] ? e->rew[t][gp] : 0.0) + kb;   /* KICK:446 */
                        e->has_rew[t][gp] = 1;
                        e->cumrew[t][gp] = e->cumrew[t][gp] + kb;     /* KICK:447 */
                    }
                }
                put_obs(e, t, cid);                                   /* BASE:412 */
                e->term[t][cid] = 0; e->has_term[t][cid] = 1;
                e->trunc[t][cid] = 0; e->has_trunc[t][cid] = 1;
            }
        }
    }

    /* Step 5b of the dense variants (DENSE:440-449; additive :463-470) */
    if (dense) {
        for (int k = 0; k < e->n_entries; ++k) {
            int t = e->ent_type[k], id = e->ent_id[k];
            if (!e->ent_present[k] || !e->has_before[t][id]) continue;
            double d = e->energy[t][id] - e->e_before[t][id];
            if (c->reward_mode == 2) d = d + e->bonus[t][id];
            e->rew[t][id] = d; e->has_rew[t][id] = 1;
            e->cumrew[t][id] += e->rew[t][id];
        }
    }

    /* Step 6, BASE:451-453 */
    for (int i = 0; i < e->n_agents; ++i) {
        int t = e->ag_type[i], id = e->ag_id[i];
        if (entry_of(e, t, id) >= 0) put_obs(e, t, id);
    }

    /* BASE:456-466 */
    out->terminated_all = (e->cur_num[PPO_PREY] <= 0 || e->cur_num[PPO_PREDATOR] <= 0);
    out->truncated_all = 0;
    int rc = emit_records(e, out);
    agents_sort(e);                                                   /* BASE:468 */
    e->current_step += 1;                                             /* BASE:471 */
    return rc;
}

/* ------------------------------------------------------------------ */
/* accessors                                                          */
/* ------------------------------------------------------------------ */

const double *ppo_grid(const ppo_env *e) { return e->grid; }
int32_t ppo_current_step(const ppo_env *e) { return e->current_step; }
int32_t ppo_num_alive(const ppo_env *e, int32_t type) { return e->cur_num[type]; }
int32_t ppo_next_id(const ppo_env *e, int32_t type) { return e->next_idx[type]; }
int32_t ppo_agents_len(const ppo_env *e) { return e->n_agents; }
void ppo_agents_get(const ppo_env *e, int32_t *types, int32_t *ids) {
    for (int i = 0; i < e->n_agents; ++i) { types[i] = e->ag_type[i]; ids[i] = e->ag_id[i]; }
}
int32_t ppo_agent_alive(const ppo_env *e, int32_t type, int32_t id) { return entry_of(e, type, id) >= 0; }
int32_t ppo_agent_get(const ppo_env *e, int32_t type, int32_t id, int32_t *x, int32_t *y,
                      double *energy, double *cum, int32_t *just_ate) {
    int k = entry_of(e, type, id);
    if (k < 0) return -1;
    *x = e->ent_x[k]; *y = e->ent_y[k];
    *energy = e->energy[type][id];
    *cum = e->cumrew[type][id];
    *just_ate = e->just_ate[type][id];
    return 0;
}
void ppo_grass_get(const ppo_env *e, int32_t *xy, double *energy) {
    for (int k = 0; k < e->n_grass; ++k) {
        xy[2 * k] = e->grass_x[k]; xy[2 * k + 1] = e->grass_y[k];
        energy[k] = e->grass_e[k];
    }
}

/* ------------------------------------------------------------------ */
/* random rollout under the live-agent protocol    */
/* ------------------------------------------------------------------ */

static void rollout_note_live(ppo_env *e, const ppo_step_out *o) {
    /* Per-type output row = position among the records of that type, taken
     * as [survivors..., newborns...]: the returned dict is ordered predator
     * survivors, prey survivors, predator newborns, prey newborns, so counting
     * per type in dict order gives exactly that. */
    int rows[2] = {0, 0};
    e->ro_n_live = 0;
    for (int i = 0; i < o->n_records; ++i) {
        const ppo_record *r = &o->records[i];
        int row = rows[r->type]++;
        if (!r->terminated) {
            int n = e->ro_n_live++;
            e->ro_live_type[n] = r->type;
            e->ro_live_id[n] = r->id;
            e->ro_live_row[n] = row;
        }
    }
    e->ro_done = o->terminated_all || o->truncated_all;
}

int64_t ppo_rollout_random(ppo_env *e, uint64_t seed, int64_t n_calls, ppo_step_out *last) {
    ppo_step_out o;
    memset(&o, 0, sizeof o);
    int cap = e->rec_cap;
    int32_t *at = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t *ai = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t *aa = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int64_t done_calls = 0;
    if (!e->ro_started) { e->ro_started = 1; e->ro_done = 1; e->seed = seed; e->episode = (uint32_t)-1; }
    for (; done_calls < n_calls; ++done_calls) {
        if (e->ro_done) {
            if (ppo_reset_philox(e, seed, e->episode + 1u, &o) != 0) break;
            rollout_note_live(e, &o);
            continue;
        }
        /* action dict in the order of the previous observation dict (live agents only) */
        int n = e->ro_n_live;
        for (int i = 0; i < n; ++i) {
            at[i] = e->ro_live_type[i]; ai[i] = e->ro_live_id[i];
            aa[i] = ppo_random_action(seed, e->episode, (uint32_t)e->current_step, at[i], e->ro_live_row[i]);
        }
        if (ppo_step(e, n, at, ai, aa, &o) != 0) break;
        rollout_note_live(e, &o);
    }
    free(at); free(ai); free(aa);
    if (last) *last = o;
    return done_calls;
}
