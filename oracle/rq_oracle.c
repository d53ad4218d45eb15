/*
 * rq_oracle.c -- CPU restatement of the reference's second-generation ("red queen") PredPreyGrass env.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rq_oracle.h).  Parity status: PINNED through tests/golden/rq_*.npz.
 *
 * Deliberately literal: a dense (4,G,G) float32 grid, insertion-ordered dictionaries, Python's list.sort()
 * on the "type_<t>_<species>_<id>" strings.  It shares no data structure with the HIP path.
 *
 * "RQ:n" = line n of /root/reference/predpreygrass/non_evolutionary/red_queen/predpreygrass_rllib_env.py
 * "WO:n" = line n of /root/reference/predpreygrass/non_evolutionary/walls_occlusion/predpreygrass_rllib_env.py
 * (the same env plus static walls, line-of-sight masking and per-agent infos; selected by config.walls)
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -o librq_oracle.so rq_oracle.c -lm
 */
#include "rq_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct rqo_env {
    rqo_config c;
    int G;
    float *grid; /* grid_world_state float32 [ch][x][y], RQ:137-139 */
    char *wall;  /* WO: wall_positions as a G*G bitmap */
    int *block_reason; /* WO: _last_move_block_reason of this call, per flat agent index */

    int off[5];  /* flat index of (pool, id) = off[pool] + id */
    int tot;

    int n_agents; /* self.agents */
    int *ag_pool, *ag_id;

    int n_entries; /* self.agent_positions: insertion-ordered dict */
    int *ent_pool, *ent_id, *ent_x, *ent_y;
    char *ent_present;
    int *ent_index; /* flat -> entry or -1 */

    double *energy, *cumrew;
    int *age, *last_repro;
    char *just_ate;

    int n_grass;
    int *grass_x, *grass_y;
    double *grass_e;

    int n_pending;
    int *pend_pool, *pend_id;

    int next_idx[4];
    int current_step;
    int active[2]; /* active_num_predators, active_num_prey */

    uint64_t seed;
    uint32_t episode;

    char *has_obs, *has_rew, *has_term, *has_trunc;
    int *obs_at;
    double *rew;
    char *term, *trunc;

    float *arena;
    size_t arena_len, arena_cap;

    rqo_record *rec;
    int rec_cap;

    /* uniform stream of the current call */
    const double *uni;
    int n_uni, uni_pos, uni_dry;

    int ro_done, ro_started, ro_n_live;
    int *ro_live_pool, *ro_live_id, *ro_live_row;
};

static int flat(const rqo_env *e, int pool, int id) { return e->off[pool] + id; }
static int pool_cap(const rqo_env *e, int pool) {
    int n = e->c.n_possible[pool];
    if (e->c.n_initial[pool] > n) n = e->c.n_initial[pool];
    return n + 1;
}
static int obs_range(const rqo_env *e, int pool) {
    return RQO_IS_PREY(pool) ? e->c.prey_obs_range : e->c.predator_obs_range; /* RQ:349 */
}
static float *cell(rqo_env *e, int ch, int x, int y) { return &e->grid[((size_t)ch * e->G + x) * e->G + y]; }
static void agent_name(int pool, int id, char *buf) { /* RQ:131,727,812 */
    sprintf(buf, "type_%d_%s_%d", RQO_TYPE_OF(pool), RQO_IS_PREY(pool) ? "prey" : "predator", id);
}

rqo_env *rqo_create(const rqo_config *cfg) {
    if (cfg->num_obs_channels != 4 || cfg->grid_size < 1) return NULL;
    rqo_env *e = (rqo_env *)calloc(1, sizeof(*e));
    e->c = *cfg;
    e->G = cfg->grid_size;
    e->off[0] = 0;
    for (int p = 0; p < 4; ++p) e->off[p + 1] = e->off[p] + pool_cap(e, p);
    int tot = e->tot = e->off[4];
    e->grid = (float *)calloc((size_t)4 * e->G * e->G, sizeof(float));
    e->wall = (char *)calloc((size_t)e->G * e->G, 1);
    e->block_reason = (int *)calloc(tot + 8, sizeof(int));
    e->ag_pool = (int *)calloc(tot + 8, sizeof(int));
    e->ag_id = (int *)calloc(tot + 8, sizeof(int));
    e->ent_pool = (int *)calloc(tot + 8, sizeof(int));
    e->ent_id = (int *)calloc(tot + 8, sizeof(int));
    e->ent_x = (int *)calloc(tot + 8, sizeof(int));
    e->ent_y = (int *)calloc(tot + 8, sizeof(int));
    e->ent_present = (char *)calloc(tot + 8, 1);
    e->ent_index = (int *)malloc((tot + 8) * sizeof(int));
    e->pend_pool = (int *)calloc(tot + 8, sizeof(int));
    e->pend_id = (int *)calloc(tot + 8, sizeof(int));
    e->energy = (double *)calloc(tot + 8, sizeof(double));
    e->cumrew = (double *)calloc(tot + 8, sizeof(double));
    e->age = (int *)calloc(tot + 8, sizeof(int));
    e->last_repro = (int *)calloc(tot + 8, sizeof(int));
    e->just_ate = (char *)calloc(tot + 8, 1);
    e->has_obs = (char *)calloc(tot + 8, 1);
    e->has_rew = (char *)calloc(tot + 8, 1);
    e->has_term = (char *)calloc(tot + 8, 1);
    e->has_trunc = (char *)calloc(tot + 8, 1);
    e->obs_at = (int *)calloc(tot + 8, sizeof(int));
    e->rew = (double *)calloc(tot + 8, sizeof(double));
    e->term = (char *)calloc(tot + 8, 1);
    e->trunc = (char *)calloc(tot + 8, 1);
    for (int i = 0; i < tot + 8; ++i) e->ent_index[i] = -1;
    e->n_grass = cfg->initial_num_grass;
    e->grass_x = (int *)calloc(e->n_grass + 1, sizeof(int));
    e->grass_y = (int *)calloc(e->n_grass + 1, sizeof(int));
    e->grass_e = (double *)calloc(e->n_grass + 1, sizeof(double));
    e->ro_live_pool = (int *)calloc(tot + 8, sizeof(int));
    e->ro_live_id = (int *)calloc(tot + 8, sizeof(int));
    e->ro_live_row = (int *)calloc(tot + 8, sizeof(int));
    e->rec_cap = tot + 8;
    e->rec = (rqo_record *)calloc(e->rec_cap, sizeof(rqo_record));
    e->arena_cap = 1 << 16;
    e->arena = (float *)malloc(e->arena_cap * sizeof(float));
    e->ro_done = 1;
    return e;
}

void rqo_destroy(rqo_env *e) {
    if (!e) return;
    free(e->grid); free(e->wall); free(e->block_reason); free(e->ag_pool); free(e->ag_id);
    free(e->ent_pool); free(e->ent_id); free(e->ent_x); free(e->ent_y); free(e->ent_present); free(e->ent_index);
    free(e->pend_pool); free(e->pend_id);
    free(e->energy); free(e->cumrew); free(e->age); free(e->last_repro); free(e->just_ate);
    free(e->has_obs); free(e->has_rew); free(e->has_term); free(e->has_trunc);
    free(e->obs_at); free(e->rew); free(e->term); free(e->trunc);
    free(e->grass_x); free(e->grass_y); free(e->grass_e);
    free(e->ro_live_pool); free(e->ro_live_id); free(e->ro_live_row);
    free(e->rec); free(e->arena);
    free(e);
}

void rqo_set_seed(rqo_env *e, uint64_t seed, uint32_t episode) { e->seed = seed; e->episode = episode; }

/* ------------------------------------------------------------------ */
/* observation: RQ:345-373                                            */
/* ------------------------------------------------------------------ */

static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* _line_of_sight_clear (WO:492-525) / the bresenham generator inside _get_observation (WO:550-575): no wall on the
 * cells strictly between start and end.  The error term is the reference's float (dx / 2.0). */
static int los_clear(const rqo_env *e, int x0, int y0, int x1, int y1) {
    const int G = e->G;
    int dx = abs(x1 - x0), dy = abs(y1 - y0);
    int x = x0, y = y0;
    int sx = x1 > x0 ? 1 : -1, sy = y1 > y0 ? 1 : -1;
    if (dx >= dy) {
        double err = dx / 2.0;
        while (x != x1) {
            if (!(x == x0 && y == y0) && !(x == x1 && y == y1) && e->wall[x * G + y]) return 0;
            err -= dy;
            if (err < 0) { y += sy; err += dx; }
            x += sx;
        }
    } else {
        double err = dy / 2.0;
        while (y != y1) {
            if (!(x == x0 && y == y0) && !(x == x1 && y == y1) && e->wall[x * G + y]) return 0;
            err -= dx;
            if (err < 0) { x += sx; err += dy; }
            y += sy;
        }
    }
    return 1;
}

static int n_channels(const rqo_env *e) { return 4 + (e->c.walls && e->c.include_visibility_channel ? 1 : 0); }

/* _get_observation of the walls env, WO:527-601 */
static void observe_at_walls(const rqo_env *e, int xp, int yp, int R, float *obs) {
    const int G = e->G;
    int off = (R - 1) / 2;
    int xld = xp - off, xhd = xp + off, yld = yp - off, yhd = yp + off;
    int xlo = clipi(xld, 0, G - 1), xhi = clipi(xhd, 0, G - 1);
    int ylo = clipi(yld, 0, G - 1), yhi = clipi(yhd, 0, G - 1);
    int xolo = abs(clipi(xld, -off, 0)), yolo = abs(clipi(yld, -off, 0));
    int xohi = xolo + (xhi - xlo) + 1, yohi = yolo + (yhi - ylo) + 1;
    const int C = n_channels(e);
    for (int i = 0; i < C * R * R; ++i) obs[i] = 0.0f;                 /* WO:535 */
    for (int i = xolo; i < xohi; ++i)
        for (int j = yolo; j < yohi; ++j) {
            int gx = xlo + (i - xolo), gy = ylo + (j - yolo);
            if (e->wall[gx * G + gy]) obs[(0 * R + i) * R + j] = 1.0f; /* WO:537-541 */
            for (int ch = 1; ch < 4; ++ch)                             /* WO:543 */
                obs[((size_t)ch * R + i) * R + j] = e->grid[((size_t)ch * G + gx) * G + gy];
        }
    if (!(e->c.include_visibility_channel || e->c.mask_observation_with_visibility)) return;
    float *vis = (float *)calloc((size_t)R * R, sizeof(float));        /* WO:548 */
    for (int lx = 0; lx < R; ++lx)
        for (int ly = 0; ly < R; ++ly) {                               /* WO:577-589 */
            int gx = xlo + (lx - xolo), gy = ylo + (ly - yolo);
            if (!(0 <= gx && gx < G && 0 <= gy && gy < G)) continue;
            vis[lx * R + ly] = los_clear(e, xp, yp, gx, gy) ? 1.0f : 0.0f;
        }
    if (e->c.mask_observation_with_visibility)                         /* WO:591-594: float32 multiply */
        for (int ch = 1; ch < 4; ++ch)
            for (int i = 0; i < R * R; ++i) obs[(size_t)ch * R * R + i] *= vis[i];
    if (e->c.include_visibility_channel)                               /* WO:596-598 */
        for (int i = 0; i < R * R; ++i) obs[(size_t)4 * R * R + i] = vis[i];
    free(vis);
}

static void observe_at(const rqo_env *e, int xp, int yp, int R, float *obs) {
    const int G = e->G;
    if (e->c.walls) { observe_at_walls(e, xp, yp, R, obs); return; }
    int off = (R - 1) / 2;                                             /* RQ:366 */
    int xld = xp - off, xhd = xp + off, yld = yp - off, yhd = yp + off;
    int xlo = clipi(xld, 0, G - 1), xhi = clipi(xhd, 0, G - 1);
    int ylo = clipi(yld, 0, G - 1), yhi = clipi(yhd, 0, G - 1);
    int xolo = abs(clipi(xld, -off, 0)), yolo = abs(clipi(yld, -off, 0));
    int xohi = xolo + (xhi - xlo) + 1, yohi = yolo + (yhi - ylo) + 1;  /* RQ:372-373 */
    for (int i = 0; i < 4 * R * R; ++i) obs[i] = 0.0f;                 /* RQ:352-355 */
    for (int i = 0; i < R * R; ++i) obs[i] = 1.0f;                     /* RQ:356 */
    for (int i = xolo; i < xohi; ++i)
        for (int j = yolo; j < yohi; ++j) {
            obs[(0 * R + i) * R + j] = 0.0f;                           /* RQ:357 */
            int gx = xlo + (i - xolo), gy = ylo + (j - yolo);
            for (int ch = 1; ch < 4; ++ch)                             /* RQ:358 */
                obs[((size_t)ch * R + i) * R + j] = e->grid[((size_t)ch * G + gx) * G + gy];
        }
}

static int entry_of(const rqo_env *e, int pool, int id) {
    if (pool < 0 || pool > 3 || id < 0 || id >= pool_cap(e, pool)) return -1;
    int k = e->ent_index[flat(e, pool, id)];
    if (k < 0 || !e->ent_present[k]) return -1;
    return k;
}

int rqo_observe(const rqo_env *e, int32_t pool, int32_t id, float *dst) {
    int k = entry_of(e, pool, id);
    if (k < 0) return -1;
    observe_at(e, e->ent_x[k], e->ent_y[k], obs_range(e, pool), dst);
    return 0;
}

static void put_obs(rqo_env *e, int pool, int id) {
    int R = obs_range(e, pool);
    size_t len = (size_t)n_channels(e) * R * R;
    if (e->arena_len + len > e->arena_cap) {
        while (e->arena_len + len > e->arena_cap) e->arena_cap *= 2;
        e->arena = (float *)realloc(e->arena, e->arena_cap * sizeof(float));
    }
    int k = entry_of(e, pool, id);
    observe_at(e, e->ent_x[k], e->ent_y[k], R, e->arena + e->arena_len);
    int f = flat(e, pool, id);
    e->obs_at[f] = (int)e->arena_len;
    e->has_obs[f] = 1;
    e->arena_len += len;
}

static void clear_call_dicts(rqo_env *e) {
    for (int i = 0; i < e->tot; ++i) e->block_reason[i] = RQO_BLOCK_NO_INFO;   /* WO:308-309 */
    memset(e->has_obs, 0, e->tot); memset(e->has_rew, 0, e->tot);
    memset(e->has_term, 0, e->tot); memset(e->has_trunc, 0, e->tot);
    e->arena_len = 0;
}

static void positions_insert(rqo_env *e, int pool, int id, int x, int y) {
    int k = e->n_entries++;
    e->ent_pool[k] = pool; e->ent_id[k] = id; e->ent_x[k] = x; e->ent_y[k] = y;
    e->ent_present[k] = 1;
    e->ent_index[flat(e, pool, id)] = k;
}
static void positions_delete(rqo_env *e, int pool, int id) { e->ent_present[e->ent_index[flat(e, pool, id)]] = 0; }
static int in_pending(const rqo_env *e, int pool, int id) {
    for (int i = 0; i < e->n_pending; ++i)
        if (e->pend_pool[i] == pool && e->pend_id[i] == id) return 1;
    return 0;
}

static int cmp_names(const void *a, const void *b) {
    const int *pa = (const int *)a, *pb = (const int *)b;
    char na[48], nb[48];
    agent_name(pa[0], pa[1], na);
    agent_name(pb[0], pb[1], nb);
    return strcmp(na, nb);
}
static void agents_sort(rqo_env *e) { /* RQ:270 */
    int n = e->n_agents;
    int *tmp = (int *)malloc((size_t)n * 2 * sizeof(int) + 8);
    for (int i = 0; i < n; ++i) { tmp[2 * i] = e->ag_pool[i]; tmp[2 * i + 1] = e->ag_id[i]; }
    qsort(tmp, n, 2 * sizeof(int), cmp_names);
    for (int i = 0; i < n; ++i) { e->ag_pool[i] = tmp[2 * i]; e->ag_id[i] = tmp[2 * i + 1]; }
    free(tmp);
}

static int emit_records(rqo_env *e, rqo_step_out *out) { /* RQ:262-265 */
    int n = 0;
    for (int i = 0; i < e->n_agents; ++i) {
        int p = e->ag_pool[i], id = e->ag_id[i], f = flat(e, p, id);
        if (!e->has_obs[f] || !e->has_rew[f] || !e->has_term[f] || !e->has_trunc[f]) return -9;
        rqo_record *r = &e->rec[n++];
        int R = obs_range(e, p);
        r->pool = p; r->id = id;
        r->reward = e->rew[f];
        r->terminated = e->term[f];
        r->truncated = e->trunc[f];
        r->obs_offset = e->obs_at[f];
        r->obs_len = n_channels(e) * R * R;
        r->block_reason = e->c.walls ? e->block_reason[f] : RQO_BLOCK_NO_INFO;
    }
    out->n_records = n;
    out->records = e->rec;
    out->obs = e->arena;
    return 0;
}

/* _register_new_agent, RQ:987-999 (the parts that feed back into step()) */
static void register_new_agent(rqo_env *e, int pool, int id) {
    int f = flat(e, pool, id);
    e->age[f] = 0;                                                     /* RQ:993 */
    e->last_repro[f] = -e->c.reproduction_cooldown_steps;              /* RQ:999 */
}

/* ------------------------------------------------------------------ */
/* reset: RQ:88-195 with the placement supplied by the caller         */
/* ------------------------------------------------------------------ */

int rqo_set_walls(rqo_env *e, int32_t n, const int32_t *xy) {
    const int G = e->G;
    memset(e->wall, 0, (size_t)G * G);
    for (int i = 0; i < n; ++i) {
        int x = xy[2 * i], y = xy[2 * i + 1];
        if (x < 0 || x >= G || y < 0 || y >= G) return -5;
        e->wall[x * G + y] = 1;
    }
    return 0;
}

int rqo_reset_from_placement(rqo_env *e, const int32_t *pred_xy, const int32_t *prey_xy,
                             const int32_t *grass_xy, rqo_step_out *out) {
    const rqo_config *c = &e->c;
    const int G = e->G;
    int total = c->initial_num_grass;
    for (int p = 0; p < 4; ++p) total += c->n_initial[p];
    if (total > G * G) return -4;                                      /* RQ:881-882 */
    e->current_step = 0;                                               /* RQ:90 */
    memset(e->grid, 0, (size_t)4 * G * G * sizeof(float));             /* RQ:138-139 */
    if (e->c.walls)
        for (int i = 0; i < G * G; ++i)
            if (e->wall[i]) e->grid[i] = 1.0f;                         /* WO:271-273: walls painted into channel 0 */
    e->n_agents = 0; e->n_entries = 0; e->n_pending = 0;
    for (int i = 0; i < e->tot; ++i) e->ent_index[i] = -1;
    memset(e->just_ate, 0, e->tot);
    /* RQ:125-133: predators type 1, predators type 2, prey type 1, prey type 2 */
    for (int p = 0; p < 4; ++p) {
        e->next_idx[p] = c->n_initial[p];                              /* RQ:129 */
        for (int i = 0; i < c->n_initial[p]; ++i) {
            e->ag_pool[e->n_agents] = p; e->ag_id[e->n_agents++] = i;
            register_new_agent(e, p, i);
        }
    }
    /* RQ:161-180: predator_list / prey_list keep self.agents order; consecutive slices of the positions */
    int np = 0, nq = 0;
    for (int i = 0; i < e->n_agents; ++i) {
        int p = e->ag_pool[i], id = e->ag_id[i];
        int x, y;
        if (!RQO_IS_PREY(p)) { x = pred_xy[2 * np]; y = pred_xy[2 * np + 1]; np++; }
        else { x = prey_xy[2 * nq]; y = prey_xy[2 * nq + 1]; nq++; }
        if (x < 0 || x >= G || y < 0 || y >= G) return -5;
        positions_insert(e, p, id, x, y);
        double e0 = RQO_IS_PREY(p) ? c->initial_energy_prey : c->initial_energy_predator;
        e->energy[flat(e, p, id)] = e0;
        *cell(e, RQO_IS_PREY(p) ? 2 : 1, x, y) = (float)e0;            /* RQ:172/179 */
        e->cumrew[flat(e, p, id)] = 0;                                 /* RQ:173/180 */
    }
    for (int k = 0; k < e->n_grass; ++k) {                             /* RQ:182-186 */
        int x = grass_xy[2 * k], y = grass_xy[2 * k + 1];
        if (x < 0 || x >= G || y < 0 || y >= G) return -5;
        e->grass_x[k] = x; e->grass_y[k] = y;
        e->grass_e[k] = c->initial_energy_grass;
        *cell(e, 3, x, y) = (float)c->initial_energy_grass;
    }
    e->active[0] = np; e->active[1] = nq;                              /* RQ:188-189 */
    clear_call_dicts(e);
    for (int i = 0; i < e->n_agents; ++i) {                            /* RQ:194 */
        int p = e->ag_pool[i], id = e->ag_id[i], f = flat(e, p, id);
        put_obs(e, p, id);
        e->rew[f] = 0.0; e->has_rew[f] = 1;
        e->term[f] = 0; e->has_term[f] = 1;
        e->trunc[f] = 0; e->has_trunc[f] = 1;
    }
    if (out) {
        out->terminated_all = 0; out->truncated_all = 0;
        out->fallback_spawns = 0; out->failed_spawns = 0; out->draws = 0;
        return emit_records(e, out);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Philox4x32-10 and the build-defined RNG contract                   */
/* ------------------------------------------------------------------ */

#define RQO_TAG_ACT 0x41435431u
#define RQO_TAG_RST 0x52535431u
#define RQO_TAG_SPW 0x53505731u
#define RQO_TAG_REP 0x52455031u

static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
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
static void philox_words(uint64_t seed, uint32_t tag, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t out[4]) {
    uint32_t ctr[4] = {a, b, c, d};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ tag};
    philox4x32_10(ctr, key, out);
}

/* draw number `draw` of step `step`: a 53-bit uniform in [0,1) like numpy's random() */
double rqo_philox_uniform(uint64_t seed, uint32_t episode, uint32_t step, uint32_t draw) {
    uint32_t w[4];
    philox_words(seed, RQO_TAG_REP, step, draw, 0u, episode, w);
    return ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) * (1.0 / 9007199254740992.0);
}

int32_t rqo_random_action(uint64_t seed, uint32_t episode, uint32_t step, int32_t species, int32_t row, int32_t n_actions) {
    uint32_t lane = (uint32_t)row & 63u;
    uint32_t reg = species == 0 ? 0u : 1u + ((uint32_t)row >> 6);
    uint32_t w[4];
    philox_words(seed, RQO_TAG_ACT, step, lane + 64u * (reg >> 2), 0u, episode, w);
    return (int32_t)(((uint64_t)w[reg & 3u] * (uint32_t)n_actions) >> 32);
}

int rqo_reset_philox(rqo_env *e, uint64_t seed, uint32_t episode, rqo_step_out *out) {
    const rqo_config *c = &e->c;
    const int G = e->G, n = G * G;
    int P = c->n_initial[0] + c->n_initial[1], Q = c->n_initial[2] + c->n_initial[3];
    int K = P + Q + c->initial_num_grass;
    /* build contract: partial Fisher-Yates over the cells that are not walls, in cell-index order */
    int *perm = (int *)malloc((size_t)n * sizeof(int));
    int nfree = 0;
    for (int i = 0; i < n; ++i)
        if (!(c->walls && e->wall[i])) perm[nfree++] = i;
    if (K > nfree) { free(perm); return -4; }
    rqo_set_seed(e, seed, episode);
    int32_t *xy = (int32_t *)malloc((size_t)2 * K * sizeof(int32_t) + 8);
    for (int k = 0; k < K; ++k) {
        uint32_t w[4];
        philox_words(seed, RQO_TAG_RST, (uint32_t)k >> 2, 0u, 0u, episode, w);
        int j = k + (int)(((uint64_t)w[k & 3] * (uint32_t)(nfree - k)) >> 32);
        int a = perm[k], b = perm[j];
        perm[j] = a; perm[k] = b;
        xy[2 * k] = b / G; xy[2 * k + 1] = b % G;
    }
    int rc = rqo_reset_from_placement(e, xy, xy + 2 * P, xy + 2 * (P + Q), out);
    free(perm); free(xy);
    return rc;
}

/* ------------------------------------------------------------------ */
/* step: RQ:197-299                                                   */
/* ------------------------------------------------------------------ */

static double next_uniform(rqo_env *e) { /* self.rng.random() */
    int d = e->uni_pos++;
    if (e->uni) {
        if (d >= e->n_uni) { e->uni_dry = 1; return 0.0; }
        return e->uni[d];
    }
    return rqo_philox_uniform(e->seed, e->episode, (uint32_t)e->current_step, (uint32_t)d);
}

/* Python's min(a, b): b if b < a else a */
static double pymin(double a, double b) { return (b < a) ? b : a; }

/* _find_available_spawn_position, RQ:375-403 */
static int find_spawn(rqo_env *e, int x, int y, int child_species, int child_id, int *nx, int *ny, int *fallback) {
    const int G = e->G;
    static const int d[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};     /* RQ:386 */
    for (int k = 0; k < 4; ++k) {
        int cx = x + d[k][0], cy = y + d[k][1];
        if (!(0 <= cx && cx < G && 0 <= cy && cy < G)) continue;       /* RQ:387 */
        int occ = 0;                                                   /* RQ:391 (occupied = all agent positions, RQ:748) */
        for (int i = 0; i < e->n_entries; ++i)
            if (e->ent_present[i] && e->ent_x[i] == cx && e->ent_y[i] == cy) { occ = 1; break; }
        if (!occ) { *nx = cx; *ny = cy; return 0; }                    /* RQ:393-394 */
    }
    /* RQ:396-401: rng.integers over list(set difference) -- the order of that list is CPython's set order, so
     * there is nothing to be bit-exact with (and it would consume from the uniform stream).  Build contract:
     * k-th free cell in x-major order, k from Philox; golden cases never reach it. */
    *fallback = 1;
    char *occ = (char *)calloc((size_t)G * G, 1);
    for (int i = 0; i < e->n_entries; ++i)
        if (e->ent_present[i]) occ[e->ent_x[i] * G + e->ent_y[i]] = 1;
    int nfree = 0;
    for (int i = 0; i < G * G; ++i) nfree += !occ[i];
    if (nfree == 0) { free(occ); return -1; }                          /* RQ:403 */
    uint32_t w[4];
    philox_words(e->seed, RQO_TAG_SPW, (uint32_t)e->current_step, (uint32_t)child_id, (uint32_t)child_species, e->episode, w);
    int k = (int)(((uint64_t)w[0] * (uint32_t)nfree) >> 32);
    for (int i = 0; i < G * G; ++i)
        if (!occ[i] && k-- == 0) { *nx = i / G; *ny = i % G; break; }
    free(occ);
    return 0;
}

static int act_range(const rqo_env *e, int pool) {
    return RQO_TYPE_OF(pool) == 1 ? e->c.type_1_action_range : e->c.type_2_action_range;
}

/* _handle_predator_reproduction / _handle_prey_reproduction, RQ:695-866 (the two differ only in names) */
static void handle_reproduction(rqo_env *e, int pool, int id, rqo_step_out *out) {
    const rqo_config *c = &e->c;
    const int prey = RQO_IS_PREY(pool);
    const int f = flat(e, pool, id);
    int cooldown = c->reproduction_cooldown_steps;                     /* RQ:696 */
    if (e->current_step - e->last_repro[f] < cooldown) return;         /* RQ:697-698 */
    double chance = prey ? c->reproduction_chance_prey : c->reproduction_chance_predator;
    if (next_uniform(e) > chance) return;                              /* RQ:701-702 */
    double thr = prey ? c->prey_creation_energy_threshold : c->predator_creation_energy_threshold;
    if (!(e->energy[f] >= thr)) return;                                /* RQ:704/789 */
    int parent_type = RQO_TYPE_OF(pool);                               /* RQ:705 */
    double mrate = prey ? c->mutation_rate_prey : c->mutation_rate_predator;
    int mutated = next_uniform(e) < mrate;                             /* RQ:708/793 */
    int new_type = mutated ? (parent_type == 1 ? 2 : 1) : parent_type; /* RQ:709-712 */
    int new_pool = (prey ? 2 : 0) + (new_type - 1);
    double rr = (prey ? c->reproduction_reward_prey : c->reproduction_reward_predator)[parent_type - 1];
    if (e->next_idx[new_pool] >= c->n_possible[new_pool]) {            /* RQ:715-725: reward even without a slot */
        e->rew[f] = rr; e->has_rew[f] = 1;
        e->cumrew[f] += e->rew[f];
        return;
    }
    int k = e->ent_index[f];
    int cid = e->next_idx[new_pool];
    int nx = 0, ny = 0, fb = 0;
    /* RQ:748-749 (looked up before the bookkeeping below; the reference crashes when it returns None) */
    int rc = find_spawn(e, e->ent_x[k], e->ent_y[k], prey, cid, &nx, &ny, &fb);
    out->fallback_spawns += fb;
    if (rc < 0) { out->failed_spawns += 1; return; }
    e->next_idx[new_pool] += 1;                                        /* RQ:728 */
    e->ag_pool[e->n_agents] = new_pool; e->ag_id[e->n_agents++] = cid; /* RQ:729 */
    e->last_repro[f] = e->current_step;                                /* RQ:737 */
    register_new_agent(e, new_pool, cid);                              /* RQ:739 */
    positions_insert(e, new_pool, cid, nx, ny);                        /* RQ:751-752 */
    double e0 = prey ? c->initial_energy_prey : c->initial_energy_predator;
    double energy_given = e0 * c->reproduction_energy_efficiency;      /* RQ:754-755 */
    int cf = flat(e, new_pool, cid);
    e->energy[cf] = energy_given;                                      /* RQ:756 */
    e->energy[f] -= e0;                                                /* RQ:757 */
    int ch = prey ? 2 : 1;
    *cell(e, ch, nx, ny) = (float)e0;                                  /* RQ:760: the grid shows the full initial energy */
    *cell(e, ch, e->ent_x[k], e->ent_y[k]) = (float)e->energy[f];      /* RQ:761 */
    e->active[prey] += 1;                                              /* RQ:763 */
    e->rew[cf] = 0; e->has_rew[cf] = 1;                                /* RQ:766 */
    e->rew[f] = rr; e->has_rew[f] = 1;                                 /* RQ:767 */
    e->cumrew[cf] = 0;                                                 /* RQ:768 */
    e->cumrew[f] += e->rew[f];                                         /* RQ:769 */
    put_obs(e, new_pool, cid);                                         /* RQ:771 */
    e->term[cf] = 0; e->has_term[cf] = 1;
    e->trunc[cf] = 0; e->has_trunc[cf] = 1;
}

int rqo_step(rqo_env *e, int32_t n_act, const int32_t *act_pool, const int32_t *act_id, const int32_t *act,
             const double *uniforms, int32_t n_uniforms, rqo_step_out *out) {
    const rqo_config *c = &e->c;
    const int G = e->G;
    clear_call_dicts(e);                                               /* RQ:198 */
    out->fallback_spawns = 0; out->failed_spawns = 0; out->draws = 0;
    e->uni = uniforms; e->n_uni = n_uniforms; e->uni_pos = 0; e->uni_dry = 0;
    memset(e->just_ate, 0, e->tot);                                    /* RQ:200 */

    for (int p = 0; p < e->n_pending; ++p)                             /* RQ:202-205 */
        for (int i = 0; i < e->n_agents; ++i)
            if (e->ag_pool[i] == e->pend_pool[p] && e->ag_id[i] == e->pend_id[p]) {
                memmove(e->ag_pool + i, e->ag_pool + i + 1, (size_t)(e->n_agents - i - 1) * sizeof(int));
                memmove(e->ag_id + i, e->ag_id + i + 1, (size_t)(e->n_agents - i - 1) * sizeof(int));
                e->n_agents--;
                break;
            }
    e->n_pending = 0;

    if (e->current_step >= c->max_steps) {                             /* RQ:449-458 */
        for (int i = 0; i < e->n_agents; ++i) {
            int p = e->ag_pool[i], id = e->ag_id[i], f = flat(e, p, id);
            put_obs(e, p, id);
            e->rew[f] = 0.0; e->has_rew[f] = 1;
            e->trunc[f] = 1; e->has_trunc[f] = 1;
            e->term[f] = 0; e->has_term[f] = 1;
        }
        out->truncated_all = 1; out->terminated_all = 0;
        return emit_records(e, out);
    }

    /* the reference would raise KeyError at RQ:323/325 part-way through; report it before mutating */
    for (int a = 0; a < n_act; ++a) {
        if (entry_of(e, act_pool[a], act_id[a]) < 0) continue;         /* dead agents are skipped, RQ:467,521 */
        int r = act_range(e, act_pool[a]);
        int delta = r >= 1 ? (r - 1) / 2 : -1;                         /* RQ:142 */
        int n = delta >= 0 ? (2 * delta + 1) * (2 * delta + 1) : 0;
        if (act[a] < 0 || act[a] >= n) return -3;
    }

    /* Step 1, RQ:462-489 */
    for (int a = 0; a < n_act; ++a) {
        int p = act_pool[a], id = act_id[a];
        int k = entry_of(e, p, id);
        if (k < 0) continue;
        int f = flat(e, p, id);
        e->energy[f] -= RQO_IS_PREY(p) ? c->energy_loss_per_step_prey : c->energy_loss_per_step_predator;
        *cell(e, RQO_IS_PREY(p) ? 2 : 1, e->ent_x[k], e->ent_y[k]) = (float)e->energy[f];
    }
    /* Step 2, RQ:497-502 (ages exist for every id ever registered) */
    for (int a = 0; a < n_act; ++a) e->age[flat(e, act_pool[a], act_id[a])] += 1;
    /* Step 3, RQ:504-514 */
    for (int g = 0; g < e->n_grass; ++g) {
        e->grass_e[g] = pymin(e->grass_e[g] + c->energy_gain_per_step_grass, c->max_energy_grass);
        *cell(e, 3, e->grass_x[g], e->grass_y[g]) = (float)e->grass_e[g];
    }
    /* Step 4, RQ:516-542 */
    for (int a = 0; a < n_act; ++a) {
        int p = act_pool[a], id = act_id[a];
        int k = entry_of(e, p, id);
        if (k < 0) continue;                                           /* RQ:521 */
        int f = flat(e, p, id);
        int ox = e->ent_x[k], oy = e->ent_y[k];
        /* _get_move, RQ:315-343; action map RQ:141-146: i -> (dx, dy), dx outer, dy inner */
        int r = act_range(e, p), delta = (r - 1) / 2, side = 2 * delta + 1;
        int dx = act[a] / side - delta, dy = act[a] % side - delta;
        int nx = clipi(ox + dx, 0, G - 1), ny = clipi(oy + dy, 0, G - 1);   /* RQ:336 */
        int ch = RQO_IS_PREY(p) ? 2 : 1;                               /* RQ:338 */
        if (!c->walls) {
            if (*cell(e, ch, nx, ny) > 0) { nx = ox; ny = oy; }        /* RQ:339-341 */
        } else {                                                       /* WO:466-488 */
            int reason = RQO_BLOCK_NONE;
            if (e->wall[nx * G + ny]) { nx = ox; ny = oy; reason = RQO_BLOCK_WALL; }
            else if (*cell(e, ch, nx, ny) > 0) { nx = ox; ny = oy; reason = RQO_BLOCK_OCCUPIED; }
            else if (c->respect_los_for_movement && !(nx == ox && ny == oy)) {
                int ddx = nx - ox, ddy = ny - oy;
                if (abs(ddx) == 1 && abs(ddy) == 1) {                  /* no corner cutting */
                    if (e->wall[(ox + ddx) * G + oy] || e->wall[ox * G + (oy + ddy)]) { nx = ox; ny = oy; reason = RQO_BLOCK_CORNER_CUT; }
                } else if (!los_clear(e, ox, oy, nx, ny)) { nx = ox; ny = oy; reason = RQO_BLOCK_LOS; }
            }
            e->block_reason[f] = reason;                               /* WO:766-778 */
        }
        e->ent_x[k] = nx; e->ent_y[k] = ny;                            /* RQ:524 */
        /* _get_movement_energy_cost, RQ:301-313 */
        double distance = sqrt((double)((nx - ox) * (nx - ox) + (ny - oy) * (ny - oy)));
        double move_cost = distance * c->move_energy_cost_factor * e->energy[f];
        e->energy[f] -= move_cost;                                     /* RQ:526 */
        *cell(e, ch, ox, oy) = 0;                                      /* RQ:537/541 */
        *cell(e, ch, nx, ny) = (float)e->energy[f];                    /* RQ:538/542 */
    }

    /* Step 5, RQ:225-233 */
    for (int i = 0; i < e->n_agents; ++i) {
        int p = e->ag_pool[i], id = e->ag_id[i], f = flat(e, p, id);
        int k = entry_of(e, p, id);
        if (k < 0) continue;                                           /* RQ:226-227 */
        int ty = RQO_TYPE_OF(p) - 1;
        if (e->energy[f] <= 0) {                                       /* RQ:228, _handle_energy_decay RQ:551-580 */
            put_obs(e, p, id);
            e->rew[f] = 0; e->has_rew[f] = 1;
            e->term[f] = 1; e->has_term[f] = 1;
            e->trunc[f] = 0; e->has_trunc[f] = 1;
            *cell(e, RQO_IS_PREY(p) ? 2 : 1, e->ent_x[k], e->ent_y[k]) = 0;  /* RQ:559 */
            e->active[RQO_IS_PREY(p)] -= 1;                            /* RQ:573/576 */
            positions_delete(e, p, id);                                /* RQ:579-580 */
        } else if (!RQO_IS_PREY(p)) {                                  /* _handle_predator_engagement RQ:582-645 */
            int px = e->ent_x[k], py = e->ent_y[k];
            int caught = -1;                                           /* RQ:584-586: first prey in dict order */
            for (int j = 0; j < e->n_entries; ++j)
                if (e->ent_present[j] && RQO_IS_PREY(e->ent_pool[j]) && e->ent_x[j] == px && e->ent_y[j] == py) {
                    caught = j; break;
                }
            if (caught >= 0) {
                int cp = e->ent_pool[caught], cid = e->ent_id[caught], cf = flat(e, cp, cid);
                e->just_ate[f] = 1;                                    /* RQ:592 */
                e->rew[f] = c->reward_predator_catch_prey[ty]; e->has_rew[f] = 1;  /* RQ:594 */
                e->cumrew[f] += e->rew[f];                             /* RQ:596 */
                double raw_gain = pymin(e->energy[cf], c->max_energy_gain_per_prey); /* RQ:598 */
                double gain = raw_gain * c->energy_transfer_efficiency;  /* RQ:599-600 */
                e->energy[f] += gain;                                  /* RQ:601 */
                e->energy[f] = pymin(e->energy[f], c->max_energy_predator); /* RQ:605-606 */
                *cell(e, 1, px, py) = (float)e->energy[f];             /* RQ:613 */
                put_obs(e, cp, cid);                                   /* RQ:615 */
                e->rew[cf] = c->penalty_prey_caught[RQO_TYPE_OF(cp) - 1]; e->has_rew[cf] = 1; /* RQ:616 */
                e->cumrew[cf] += e->rew[cf];                           /* RQ:618 */
                e->term[cf] = 1; e->has_term[cf] = 1;                  /* RQ:620-621 */
                e->trunc[cf] = 0; e->has_trunc[cf] = 1;
                e->active[1] -= 1;                                     /* RQ:622 */
                *cell(e, 2, e->ent_x[caught], e->ent_y[caught]) = 0;   /* RQ:623 */
                positions_delete(e, cp, cid);                          /* RQ:635-637 */
            } else {
                e->rew[f] = c->reward_predator_step[ty]; e->has_rew[f] = 1;  /* RQ:639 */
            }
            put_obs(e, p, id);                                         /* RQ:641 */
            e->cumrew[f] += e->rew[f];                                 /* RQ:643 */
            e->term[f] = 0; e->has_term[f] = 1;
            e->trunc[f] = 0; e->has_trunc[f] = 1;
        } else {                                                       /* _handle_prey_engagement RQ:647-693 */
            if (e->has_term[f] && e->term[f]) continue;                /* RQ:648-649 */
            int px = e->ent_x[k], py = e->ent_y[k];
            int g = -1;                                                /* RQ:652-654 */
            for (int j = 0; j < e->n_grass; ++j)
                if (e->grass_x[j] == px && e->grass_y[j] == py) { g = j; break; }
            if (g >= 0) {
                e->just_ate[f] = 1;                                    /* RQ:658 */
                e->rew[f] = c->reward_prey_eat_grass[ty]; e->has_rew[f] = 1;  /* RQ:660 */
                e->cumrew[f] += e->rew[f];                             /* RQ:663 */
                double raw_gain = pymin(e->grass_e[g], c->max_energy_gain_per_grass); /* RQ:665 */
                double gain = raw_gain * c->energy_transfer_efficiency;
                e->energy[f] += gain;                                  /* RQ:668 */
                e->energy[f] = pymin(e->energy[f], c->max_energy_prey);  /* RQ:672-673 */
                *cell(e, 2, px, py) = (float)e->energy[f];             /* RQ:680 */
                *cell(e, 3, px, py) = 0;                               /* RQ:682 */
                e->grass_e[g] = 0;                                     /* RQ:683 */
            } else {
                e->rew[f] = c->reward_prey_step[ty]; e->has_rew[f] = 1;  /* RQ:685 */
            }
            put_obs(e, p, id);                                         /* RQ:689 */
            e->cumrew[f] += e->rew[f];                                 /* RQ:691 */
            e->term[f] = 0; e->has_term[f] = 1;
            e->trunc[f] = 0; e->has_trunc[f] = 1;
        }
    }

    /* Step 6, RQ:236 */
    e->n_pending = 0;
    for (int i = 0; i < e->n_agents; ++i) {
        int p = e->ag_pool[i], id = e->ag_id[i], f = flat(e, p, id);
        if (e->has_term[f] && e->term[f]) { e->pend_pool[e->n_pending] = p; e->pend_id[e->n_pending++] = id; }
    }

    /* Step 7, RQ:248-254 (iterates a copy of self.agents) */
    int n_before = e->n_agents;
    for (int i = 0; i < n_before; ++i) {
        int p = e->ag_pool[i], id = e->ag_id[i];
        if (in_pending(e, p, id)) continue;
        handle_reproduction(e, p, id, out);
    }

    /* Step 8, RQ:257-259 */
    for (int i = 0; i < e->n_agents; ++i)
        if (entry_of(e, e->ag_pool[i], e->ag_id[i]) >= 0) put_obs(e, e->ag_pool[i], e->ag_id[i]);

    out->truncated_all = 0;                                            /* RQ:266 */
    out->terminated_all = (e->active[1] <= 0 || e->active[0] <= 0);    /* RQ:267 */
    out->draws = e->uni_pos;
    int rc = emit_records(e, out);
    agents_sort(e);                                                    /* RQ:270 */
    e->current_step += 1;                                              /* RQ:297 */
    if (e->uni_dry) return -6;
    return rc;
}

/* ------------------------------------------------------------------ */
/* accessors                                                          */
/* ------------------------------------------------------------------ */

const float *rqo_grid(const rqo_env *e) { return e->grid; }
int32_t rqo_current_step(const rqo_env *e) { return e->current_step; }
int32_t rqo_num_alive(const rqo_env *e, int32_t species) { return e->active[species]; }
int32_t rqo_next_id(const rqo_env *e, int32_t pool) { return e->next_idx[pool]; }
int32_t rqo_agents_len(const rqo_env *e) { return e->n_agents; }
void rqo_agents_get(const rqo_env *e, int32_t *pools, int32_t *ids) {
    for (int i = 0; i < e->n_agents; ++i) { pools[i] = e->ag_pool[i]; ids[i] = e->ag_id[i]; }
}
int32_t rqo_agent_alive(const rqo_env *e, int32_t pool, int32_t id) { return entry_of(e, pool, id) >= 0; }
int32_t rqo_agent_get(const rqo_env *e, int32_t pool, int32_t id, int32_t *x, int32_t *y, double *energy,
                      double *cum, int32_t *just_ate, int32_t *age, int32_t *last_reproduction) {
    int k = entry_of(e, pool, id);
    if (k < 0) return -1;
    int f = flat(e, pool, id);
    *x = e->ent_x[k]; *y = e->ent_y[k];
    *energy = e->energy[f];
    *cum = e->cumrew[f];
    *just_ate = e->just_ate[f];
    *age = e->age[f];
    *last_reproduction = e->last_repro[f];
    return 0;
}
void rqo_grass_get(const rqo_env *e, int32_t *xy, double *energy) {
    for (int k = 0; k < e->n_grass; ++k) {
        xy[2 * k] = e->grass_x[k]; xy[2 * k + 1] = e->grass_y[k];
        energy[k] = e->grass_e[k];
    }
}

/* ------------------------------------------------------------------ */
/* random rollout under the live-agent protocol                       */
/* ------------------------------------------------------------------ */

static void rollout_note_live(rqo_env *e, const rqo_step_out *o) {
    /* per-species output row = position among the records of that species in dict order: sorted survivors
     * (type 1 before type 2), then newborns in birth order -- the row order of the device tables */
    int rows[2] = {0, 0};
    e->ro_n_live = 0;
    for (int i = 0; i < o->n_records; ++i) {
        const rqo_record *r = &o->records[i];
        int row = rows[RQO_IS_PREY(r->pool)]++;
        if (!r->terminated) {
            int n = e->ro_n_live++;
            e->ro_live_pool[n] = r->pool;
            e->ro_live_id[n] = r->id;
            e->ro_live_row[n] = row;
        }
    }
    e->ro_done = o->terminated_all || o->truncated_all;
}

int64_t rqo_rollout_random(rqo_env *e, uint64_t seed, int64_t n_calls, rqo_step_out *last) {
    rqo_step_out o;
    memset(&o, 0, sizeof o);
    int cap = e->rec_cap;
    int32_t *ap = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t *ai = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t *aa = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int64_t done_calls = 0;
    if (!e->ro_started) { e->ro_started = 1; e->ro_done = 1; e->seed = seed; e->episode = (uint32_t)-1; }
    for (; done_calls < n_calls; ++done_calls) {
        if (e->ro_done) {
            if (rqo_reset_philox(e, seed, e->episode + 1u, &o) != 0) break;
            rollout_note_live(e, &o);
            continue;
        }
        int n = e->ro_n_live;
        for (int i = 0; i < n; ++i) {
            ap[i] = e->ro_live_pool[i]; ai[i] = e->ro_live_id[i];
            int r = act_range(e, ap[i]);
            aa[i] = rqo_random_action(seed, e->episode, (uint32_t)e->current_step, RQO_IS_PREY(ap[i]), e->ro_live_row[i], r * r);
        }
        if (rqo_step(e, n, ap, ai, aa, NULL, 0, &o) != 0) break;
        rollout_note_live(e, &o);
    }
    free(ap); free(ai); free(aa);
    if (last) *last = o;
    return done_calls;
}
