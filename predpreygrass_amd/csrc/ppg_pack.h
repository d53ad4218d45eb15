// ppg_pack.h -- the packed observation image of include/ppg.h (ppg_pack): what the last call of a GPU's sub-batches
// returned, compacted into ONE contiguous buffer so that a single all-gather moves the observation dict of a shard
// (the reference has no counterpart: its dicts live in one Python process, BASE:456-473).
//
// Two launches, both written against the wave primitives of wave.h (the CPU test build runs the same source):
//   pack_scan  one wavefront: exclusive prefix sums of the per-env row counts -> row_off section + header
//   pack_rows  one wavefront per env: env words, the per-row scalars and the observation blocks of the rows in use.
// The rows in use of an env are the FIRST n rows of its region of every table, so an env's observations are two
// contiguous copies (predators, prey): 16 bytes per lane when the block size allows it, float64 -> float32 on request.
#pragma once

#include <stdint.h>

#include "../../include/ppg.h"

#ifndef PPG_WAVE_EMU
#include "wave.h"
#endif

namespace ppg {

struct PackParams {
    int32_t n_handles, n_envs;
    int32_t env_base[PPG_PACK_MAX_HANDLES + 1];  // envs in front of handle k
    const int32_t *env_state[PPG_PACK_MAX_HANDLES];
    const int32_t *row_id[PPG_PACK_MAX_HANDLES];
    const double *row_reward[PPG_PACK_MAX_HANDLES];
    const uint8_t *row_flags[PPG_PACK_MAX_HANDLES];
    const unsigned char *obs_pred[PPG_PACK_MAX_HANDLES];
    const unsigned char *obs_prey[PPG_PACK_MAX_HANDLES];
    int32_t S, cap_pred, cap_prey;
    int32_t blk_pred, blk_prey;   // elements per observation block
    int32_t src_elem, dst_elem;   // bytes per observation element in the env buffers / in the image
    uint64_t capacity;
    unsigned char *out;
};

struct PackLayout {
    uint64_t env_state, row_off, id_p, id_q, rew_p, rew_q, fl_p, fl_q, obs_p, obs_q, total;
};

PPG_HOST_DEVICE uint64_t pack_align16(uint64_t v) { return (v + 15u) & ~(uint64_t)15u; }

PPG_HOST_DEVICE PackLayout pack_layout(uint64_t n_envs, uint64_t np, uint64_t nq, uint64_t blk_p, uint64_t blk_q, uint64_t elem) {
    PackLayout L;
    uint64_t o = sizeof(ppg_pack_header);
    L.env_state = o; o = pack_align16(o + n_envs * PPG_ENV_WORDS * 4);
    L.row_off = o;   o = pack_align16(o + n_envs * 8);
    L.id_p = o;      o = pack_align16(o + np * 4);
    L.id_q = o;      o = pack_align16(o + nq * 4);
    L.rew_p = o;     o = pack_align16(o + np * 8);
    L.rew_q = o;     o = pack_align16(o + nq * 8);
    L.fl_p = o;      o = pack_align16(o + np);
    L.fl_q = o;      o = pack_align16(o + nq);
    L.obs_p = o;     o = pack_align16(o + np * blk_p * elem);
    L.obs_q = o;     o = pack_align16(o + nq * blk_q * elem);
    L.total = o;
    return L;
}

struct alignas(16) Pack16 { uint32_t a, b, c, d; };
struct alignas(16) PackD2 { double x, y; };
struct alignas(8) PackF2 { float x, y; };

// handle that owns env e (n_handles <= 8: a short scan of the cumulative counts)
template <class KP>
PPG_DEVICE int pack_handle_of(const KP &K, int e) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < PPG_PACK_MAX_HANDLES; ++q) k += (q < K.n_handles && e >= K.env_base[q]) ? 1 : 0;
    return k;
}

// ---- launch 1: one wavefront ------------------------------------------------------------------------------
template <class KP>
PPG_DEVICE void pack_scan_main(const KP &K, unsigned char *lds) {
    const int ln = wv::lane();
    uint32_t *tot = (uint32_t *)lds;  // [64][2]
    const int per = (K.n_envs + 63) / 64;
    const int lo = ln * per, hi = (lo + per) < K.n_envs ? (lo + per) : K.n_envs;
    uint32_t sp = 0, sq = 0;
    for (int e = lo; e < hi; ++e) {
        const int k = pack_handle_of(K, e);
        const int32_t *es = K.env_state[k] + (size_t)(e - K.env_base[k]) * PPG_ENV_WORDS;
        sp += (uint32_t)es[PPG_ENV_N_PRED_ROWS];
        sq += (uint32_t)es[PPG_ENV_N_PREY_ROWS];
    }
    tot[2 * ln] = sp;
    tot[2 * ln + 1] = sq;
    wv::sync();
    uint32_t bp = 0, bq = 0, tp = 0, tq = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t a = tot[2 * l], b = tot[2 * l + 1];
        if (l < ln) { bp += a; bq += b; }
        tp += a; tq += b;
    }
    const PackLayout L = pack_layout((uint64_t)K.n_envs, tp, tq, (uint64_t)K.blk_pred, (uint64_t)K.blk_prey, (uint64_t)K.dst_elem);
    uint32_t *row_off = (uint32_t *)(K.out + L.row_off);
    for (int e = lo; e < hi; ++e) {
        const int k = pack_handle_of(K, e);
        const int32_t *es = K.env_state[k] + (size_t)(e - K.env_base[k]) * PPG_ENV_WORDS;
        row_off[2 * e] = bp;
        row_off[2 * e + 1] = bq;
        bp += (uint32_t)es[PPG_ENV_N_PRED_ROWS];
        bq += (uint32_t)es[PPG_ENV_N_PREY_ROWS];
    }
    if (ln == 0) {
        ppg_pack_header *H = (ppg_pack_header *)K.out;
        H->magic = PPG_PACK_MAGIC; H->version = PPG_PACK_VERSION;
        H->n_envs = (uint32_t)K.n_envs; H->n_pred_rows = tp; H->n_prey_rows = tq;
        H->obs_elem_bytes = (uint32_t)K.dst_elem;
        H->blk_pred = (uint32_t)K.blk_pred; H->blk_prey = (uint32_t)K.blk_prey;
        H->bytes_used = L.total; H->capacity = K.capacity;
        H->overflow = L.total > K.capacity ? 1u : 0u;
        H->env_words = PPG_ENV_WORDS;
        H->reserved[0] = 0; H->reserved[1] = 0;
    }
}

// n_elems observation elements from src to dst (contiguous), this wavefront's 64 lanes together
template <class KP>
PPG_DEVICE void pack_copy_obs(const KP &K, const unsigned char *src, unsigned char *dst, uint64_t n_elems, int ln) {
    const uint64_t dst_addr = (uint64_t)(uintptr_t)dst, src_addr = (uint64_t)(uintptr_t)src;
    if (K.src_elem == K.dst_elem) {
        const uint64_t bytes = n_elems * (uint64_t)K.src_elem;
        if (((dst_addr | src_addr | bytes) & 15u) == 0) {
            const Pack16 *s = (const Pack16 *)src;
            Pack16 *d = (Pack16 *)dst;
            const uint64_t n16 = bytes >> 4;
            uint64_t i = (uint64_t)ln;
            for (; i + 192 < n16; i += 256) {   // four loads in flight per lane
                const Pack16 v0 = s[i], v1 = s[i + 64], v2 = s[i + 128], v3 = s[i + 192];
                d[i] = v0; d[i + 64] = v1; d[i + 128] = v2; d[i + 192] = v3;
            }
            for (; i < n16; i += 64) d[i] = s[i];
        } else if (K.src_elem == 8) {
            const double *s = (const double *)src;
            double *d = (double *)dst;
            for (uint64_t i = (uint64_t)ln; i < n_elems; i += 64) d[i] = s[i];
        } else if (K.src_elem == 2) {   // bfloat16 rows
            const uint16_t *s = (const uint16_t *)src;
            uint16_t *d = (uint16_t *)dst;
            for (uint64_t i = (uint64_t)ln; i < n_elems; i += 64) d[i] = s[i];
        } else {
            const float *s = (const float *)src;
            float *d = (float *)dst;
            for (uint64_t i = (uint64_t)ln; i < n_elems; i += 64) d[i] = s[i];
        }
        return;
    }
    // float64 in the env buffers, float32 in the image
    if (((src_addr & 15u) | (dst_addr & 7u) | (n_elems & 1u)) == 0) {
        const PackD2 *s = (const PackD2 *)src;
        PackF2 *d = (PackF2 *)dst;
        const uint64_t n2 = n_elems >> 1;
        uint64_t i = (uint64_t)ln;
        for (; i + 192 < n2; i += 256) {
            const PackD2 v0 = s[i], v1 = s[i + 64], v2 = s[i + 128], v3 = s[i + 192];
            PackF2 f0, f1, f2, f3;
            f0.x = (float)v0.x; f0.y = (float)v0.y; f1.x = (float)v1.x; f1.y = (float)v1.y;
            f2.x = (float)v2.x; f2.y = (float)v2.y; f3.x = (float)v3.x; f3.y = (float)v3.y;
            d[i] = f0; d[i + 64] = f1; d[i + 128] = f2; d[i + 192] = f3;
        }
        for (; i < n2; i += 64) { const PackD2 v = s[i]; PackF2 f; f.x = (float)v.x; f.y = (float)v.y; d[i] = f; }
    } else {
        const double *s = (const double *)src;
        float *d = (float *)dst;
        for (uint64_t i = (uint64_t)ln; i < n_elems; i += 64) d[i] = (float)s[i];
    }
}

// ---- launch 2: one wavefront per env ------------------------------------------------------------------------
template <class KP>
PPG_DEVICE void pack_rows_main(const KP &K) {
    const int e = PPG_BLOCK_INDEX();
    if (e >= K.n_envs) return;
    const int ln = wv::lane();
    const int k = (int)wv::first((uint32_t)pack_handle_of(K, e));
    const int b = e - K.env_base[k];
    const int32_t *es = K.env_state[k] + (size_t)b * PPG_ENV_WORDS;
    const ppg_pack_header *H = (const ppg_pack_header *)K.out;
    const uint32_t np = wv::first(H->n_pred_rows), nq = wv::first(H->n_prey_rows);
    const bool overflow = wv::first(H->overflow) != 0u;
    const PackLayout L = pack_layout((uint64_t)K.n_envs, np, nq, (uint64_t)K.blk_pred, (uint64_t)K.blk_prey, (uint64_t)K.dst_elem);
    if (ln < PPG_ENV_WORDS) ((int32_t *)(K.out + L.env_state))[(size_t)e * PPG_ENV_WORDS + ln] = es[ln];
    if (overflow) return;
    const int n_p = (int)wv::first((uint32_t)es[PPG_ENV_N_PRED_ROWS]), n_q = (int)wv::first((uint32_t)es[PPG_ENV_N_PREY_ROWS]);
    const uint32_t *row_off = (const uint32_t *)(K.out + L.row_off);
    const uint64_t off_p = wv::first(row_off[2 * e]), off_q = wv::first(row_off[2 * e + 1]);
    const size_t rb = (size_t)b * K.S;
    for (int r = ln; r < n_p; r += 64) {
        ((int32_t *)(K.out + L.id_p))[off_p + r] = K.row_id[k][rb + r];
        ((double *)(K.out + L.rew_p))[off_p + r] = K.row_reward[k][rb + r];
        (K.out + L.fl_p)[off_p + r] = K.row_flags[k][rb + r];
    }
    for (int r = ln; r < n_q; r += 64) {
        ((int32_t *)(K.out + L.id_q))[off_q + r] = K.row_id[k][rb + K.cap_pred + r];
        ((double *)(K.out + L.rew_q))[off_q + r] = K.row_reward[k][rb + K.cap_pred + r];
        (K.out + L.fl_q)[off_q + r] = K.row_flags[k][rb + K.cap_pred + r];
    }
    pack_copy_obs(K, K.obs_pred[k] + (size_t)b * K.cap_pred * K.blk_pred * K.src_elem,
                  K.out + L.obs_p + off_p * (uint64_t)K.blk_pred * K.dst_elem, (uint64_t)n_p * K.blk_pred, ln);
    pack_copy_obs(K, K.obs_prey[k] + (size_t)b * K.cap_prey * K.blk_prey * K.src_elem,
                  K.out + L.obs_q + off_q * (uint64_t)K.blk_prey * K.dst_elem, (uint64_t)n_q * K.blk_prey, ln);
}

}  // namespace ppg
