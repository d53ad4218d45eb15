// ppg_fetch.h -- the host view of include/ppg.h (ppg_fetch): what the last call returned for a run of envs, gathered into ONE
// contiguous image so that the dict classes (predpreygrass_amd/env.py: PredPreyGrass.step, BASE:219,473) pay one device->host copy
// and one stream synchronisation per call instead of one per tensor.
//
// One launch, one wavefront per env, written against the wave primitives of wave.h (the CPU test build runs the same source):
// every wavefront sums the observation bytes of the envs in front of it (the envs of a fetch are few: one, or a vector env's dozens),
// copies its env's slice of every state tensor into the env's record and its observation blocks IN USE behind the records.
#pragma once

#include <stdint.h>

#include "../../include/ppg.h"
#include "ppg_pack.h"

namespace ppg {

constexpr int FETCH_MAX_FIELDS = 16;

struct FetchParams {
    int32_t env0, n_envs, n_fields;
    uint32_t record_bytes;                        // one env's record: its slice of every field, each padded to 8 bytes, the sum to 16
    const unsigned char *field[FETCH_MAX_FIELDS]; // state tensors in ppg_state_fields() order
    uint32_t field_bytes[FETCH_MAX_FIELDS];       // bytes per env
    const int32_t *env_state;
    const unsigned char *obs_pred, *obs_prey;
    int32_t cap_pred, cap_prey;
    uint32_t blk_pred_bytes, blk_prey_bytes;      // bytes of one observation block
    uint64_t capacity;
    unsigned char *out;
};

// bytes of env e's observation section: predator blocks in use, then prey blocks in use, each run padded to 16 bytes
template <class KP>
PPG_DEVICE uint64_t fetch_section_bytes(const KP &K, int e, uint32_t &np, uint32_t &nq) {
    const int32_t *es = K.env_state + (size_t)(K.env0 + e) * PPG_ENV_WORDS;
    np = (uint32_t)es[PPG_ENV_N_PRED_ROWS];
    nq = (uint32_t)es[PPG_ENV_N_PREY_ROWS];
    return pack_align16((uint64_t)np * K.blk_pred_bytes) + pack_align16((uint64_t)nq * K.blk_prey_bytes);
}

// n bytes from src to dst, the wavefront's 64 lanes together (16 bytes per lane where the alignment allows it)
PPG_DEVICE void fetch_copy(const unsigned char *src, unsigned char *dst, uint64_t n, int ln) {
    const uint64_t a = (uint64_t)(uintptr_t)src | (uint64_t)(uintptr_t)dst;
    if (((a | n) & 15u) == 0) {
        const Pack16 *s = (const Pack16 *)src;
        Pack16 *d = (Pack16 *)dst;
        const uint64_t n16 = n >> 4;
        uint64_t i = (uint64_t)ln;
        for (; i + 192 < n16; i += 256) {   // four loads in flight per lane
            const Pack16 v0 = s[i], v1 = s[i + 64], v2 = s[i + 128], v3 = s[i + 192];
            d[i] = v0; d[i + 64] = v1; d[i + 128] = v2; d[i + 192] = v3;
        }
        for (; i < n16; i += 64) d[i] = s[i];
    } else if (((a | n) & 3u) == 0) {
        const uint32_t *s = (const uint32_t *)src;
        uint32_t *d = (uint32_t *)dst;
        for (uint64_t i = (uint64_t)ln; i < (n >> 2); i += 64) d[i] = s[i];
    } else {
        for (uint64_t i = (uint64_t)ln; i < n; i += 64) dst[i] = src[i];
    }
}

template <class KP>
PPG_DEVICE void fetch_main(const KP &K, unsigned char *lds) {
    const int e = PPG_BLOCK_INDEX();
    if (e >= K.n_envs) return;
    const int ln = wv::lane();
    // this env's offset = the sections of the envs in front of it; the image's size = all of them
    uint64_t before = 0, total = 0;
    for (int q = ln; q < K.n_envs; q += 64) {
        uint32_t a, b;
        const uint64_t s = fetch_section_bytes(K, q, a, b);
        total += s;
        if (q < e) before += s;
    }
    uint64_t *red = (uint64_t *)lds;   // [64][2]
    red[2 * ln] = before;
    red[2 * ln + 1] = total;
    wv::sync();
    before = 0; total = 0;
    for (int l = 0; l < 64; ++l) { before += red[2 * l]; total += red[2 * l + 1]; }
    const uint64_t fixed = sizeof(ppg_fetch_header) + (uint64_t)K.n_envs * K.record_bytes;
    const uint64_t used = fixed + total;
    const bool overflow = used > K.capacity;
    if (e == 0 && ln == 0) {
        ppg_fetch_header *H = (ppg_fetch_header *)K.out;
        H->magic = PPG_FETCH_MAGIC; H->version = PPG_FETCH_VERSION;
        H->env0 = (uint32_t)K.env0; H->n_envs = (uint32_t)K.n_envs;
        H->record_bytes = K.record_bytes;
        H->blk_pred_bytes = K.blk_pred_bytes; H->blk_prey_bytes = K.blk_prey_bytes;
        H->overflow = overflow ? 1u : 0u;
        H->bytes_used = used; H->capacity = K.capacity;
        H->reserved[0] = H->reserved[1] = H->reserved[2] = H->reserved[3] = 0u;
    }
    // the record: always written (the caller validated capacity >= the fixed part)
    unsigned char *rec = K.out + sizeof(ppg_fetch_header) + (size_t)e * K.record_bytes;
    uint32_t off = 0;
    for (int f = 0; f < K.n_fields; ++f) {
        const uint32_t nb = K.field_bytes[f];
        fetch_copy(K.field[f] + (size_t)(K.env0 + e) * nb, rec + off, nb, ln);
        off += (nb + 7u) & ~7u;
    }
    if (overflow) return;
    uint32_t np, nq;
    (void)fetch_section_bytes(K, e, np, nq);
    unsigned char *dst = K.out + fixed + before;
    fetch_copy(K.obs_pred + (size_t)(K.env0 + e) * K.cap_pred * K.blk_pred_bytes, dst, (uint64_t)np * K.blk_pred_bytes, ln);
    dst += pack_align16((uint64_t)np * K.blk_pred_bytes);
    fetch_copy(K.obs_prey + (size_t)(K.env0 + e) * K.cap_prey * K.blk_prey_bytes, dst, (uint64_t)nq * K.blk_prey_bytes, ln);
}

}  // namespace ppg
