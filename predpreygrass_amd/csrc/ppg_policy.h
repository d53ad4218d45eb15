// ppg_policy.h -- policy inference next to the env (include/ppg.h: ppg_policy_*).
//
// The network the reference trains for each species (base_environment/tune_ppo_base_environment.py:106-141):
//     (4,R,R) observation -> conv3x3(4->16) ReLU -> conv3x3(16->32) ReLU -> conv3x3(32->64) ReLU   ("same" padding, stride 1)
//                         -> flatten -> Linear(64 R^2 -> 256) ReLU -> Linear(256 -> 256) ReLU -> Linear(256 -> n_actions)
// evaluated for every agent row in use, reading the observation rows where ppg_step wrote them and writing one int8 action per
// row.  gfx950 only: every layer is a GEMM on the matrix cores, v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate).
//
// ORIENTATION (all six layers): M = output features / channels (the A operand = weights), N = samples or positions (the B
// operand = activations).  With the 32x32 result layout (column = lane & 31, rows in the 16 registers: row = i + 8g + 4h for
// register 4g+i, h = lane >> 5) a lane then holds, for ONE sample/position, four groups of four consecutive features: the
// epilogue writes them as four 8-byte bf16x4 vectors into an activation image whose inner dimension is the feature index --
// which is exactly the 16-byte-per-lane B fragment the next layer reads (k = 8h + j: eight consecutive features).
//
// ONE persistent launch per species.  A workgroup (4 wavefronts) takes tiles of 128 samples:
//   A. convolutions in sub-groups of ST samples: activations live in LDS as [sample][channel block of 8][padded position][8]
//      bf16 with a zero halo ring (so the nine taps of the implicit GEMM are plain offsets); each wavefront keeps the layer's
//      weight fragments in registers (conv3: 144 VGPRs) and walks over 32-position tiles.  conv3 writes its output to the
//      workgroup's scratch slot in HBM/L2 as X[sample][position][64]  (= the K order the repacked FC1 weights expect).
//   B. FC1 over the whole tile (K = 64 R^2): weight fragments stream from L2, X fragments from the scratch slot (the slot was
//      written by this workgroup and is re-read after a workgroup barrier + ONE agent-scope acquire that drops stale L1 lines);
//      ReLU -> H[sample][256] in LDS.   C. FC2 from H, result back into H.   D. logits (M = actions padded to 32), the two
//      lane halves exchange their rows, argmax or Gumbel-max sampling, int8 store into actions[b][slot].
// The sample list is never materialised: a one-wavefront plan kernel writes exclusive prefix sums of the per-env row counts and
// each tile resolves sample -> (handle, env, row) by bisection.
#pragma once

#include <hip/hip_runtime.h>

#include <math.h>

#include <vector>

namespace ppgpol {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE = 128;         // samples per workgroup tile
constexpr int HSTRIDE = 264;      // H[sample][256 + 8] bf16: rows 528 B apart -> conflict-free 16-byte fragment reads
constexpr int MAX_HANDLES = PPG_PACK_MAX_HANDLES;

struct PolParams {
    // geometry
    int32_t R, P, Wp, Wp2;        // window side, R*R, R+2, (R+2)^2
    int32_t K1;                   // 64 * P
    int32_t ST;                   // samples per convolution sub-group
    int32_t n_actions;
    int32_t species;              // 0 predators, 1 prey
    int32_t obs_f32;
    int32_t sample;               // PPG_POLICY_SAMPLE
    uint32_t seed_lo, seed_hi;
    // weights in fragment order (device, bf16) and biases (float)
    const bf16x8 *wc1, *wc2, *wc3, *w1, *w2, *w3;
    const float *bc1, *bc2, *bc3, *b1, *b2, *b3;
    // envs
    int32_t n_handles, n_envs;
    int32_t env_base[MAX_HANDLES + 1];
    const int32_t *env_state[MAX_HANDLES];
    const unsigned char *obs[MAX_HANDLES];   // obs_pred or obs_prey of handle k
    int8_t *actions[MAX_HANDLES];
    int32_t S, cap, slot0;        // rows per env of the action tensor; row capacity of this species; its first slot
    // scratch
    const uint32_t *plan;         // [0] = total rows of this species, [1 + e] = exclusive prefix sum of env e
    __bf16 *xg;                   // [gridDim.x][TILE][K1]
    float *logits;                // optional [rows][n_actions]
};

struct PlanParams {
    int32_t n_handles, n_envs, word;   // word: PPG_ENV_N_PRED_ROWS / PPG_ENV_N_PREY_ROWS
    int32_t env_base[MAX_HANDLES + 1];
    const int32_t *env_state[MAX_HANDLES];
    uint32_t *plan;
};

template <class P>
__device__ __forceinline__ int handle_of(P env_base, int n_handles, int e) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < MAX_HANDLES; ++q) k += (q < n_handles && e >= env_base[q]) ? 1 : 0;
    return k;
}

// exclusive prefix sums of one env_state word over the concatenated envs of all handles (one wavefront)
extern "C" __global__ void __launch_bounds__(64) ppg_policy_plan(const PlanParams K) {
    __shared__ uint32_t tot[64];
    const int ln = (int)threadIdx.x;
    const int per = (K.n_envs + 63) / 64;
    const int lo = ln * per, hi = (lo + per) < K.n_envs ? (lo + per) : K.n_envs;
    uint32_t s = 0;
    for (int e = lo; e < hi; ++e) {
        const int k = handle_of(K.env_base, K.n_handles, e);
        s += (uint32_t)K.env_state[k][(size_t)(e - K.env_base[k]) * PPG_ENV_WORDS + K.word];
    }
    tot[ln] = s;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int l = 0; l < 64; ++l) { const uint32_t a = tot[l]; if (l < ln) before += a; all += a; }
    for (int e = lo; e < hi; ++e) {
        const int k = handle_of(K.env_base, K.n_handles, e);
        K.plan[1 + e] = before;
        before += (uint32_t)K.env_state[k][(size_t)(e - K.env_base[k]) * PPG_ENV_WORDS + K.word];
    }
    if (ln == 0) K.plan[0] = all;
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
    return v;
}

__device__ __forceinline__ bf16x4 relu_pack(float a, float b, float c, float d) {
    bf16x4 v;
    v[0] = (__bf16)(a > 0.f ? a : 0.f); v[1] = (__bf16)(b > 0.f ? b : 0.f);
    v[2] = (__bf16)(c > 0.f ? c : 0.f); v[3] = (__bf16)(d > 0.f ? d : 0.f);
    return v;
}

// One convolution layer over the `ns` samples of a sub-group, this wavefront's share of the 32-position tiles.
//   CBIN   channel blocks (of 8) of the input image: 1 (4 real channels), 2, 4        K = 9 taps x CBIN blocks
//   MT     32-row tiles of output channels: 1 (16 real for conv1 / 32 for conv2), 2 (conv3)
//   COUT_BLOCKS  channel blocks written: 2 (conv1), 4 (conv2), 8 (conv3)
//   TO_GLOBAL    conv3: the result goes to the scratch slot X[sample][position][64] instead of an LDS image
template <int CBIN, int MT, int COUT_BLOCKS, bool TO_GLOBAL, class KP>
__device__ __forceinline__ void conv_layer(const KP &K, const bf16x8 *wfrag, const float *bias, const __bf16 *in,
                                           int in_sample_stride, __bf16 *out, int out_sample_stride, __bf16 *xg_tile,
                                           int s_local0, int ns, int wave, int lane, int mt_base = 0) {
    constexpr int Q = 9 * CBIN, KS = (Q + 1) / 2;
    const int h = lane >> 5, col = lane & 31;
    // the layer's weight fragments and the per-k-step offsets of this lane half stay in registers for all tiles
    bf16x8 a[MT][KS];
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt][ks] = wfrag[((mt_base + mt) * KS + ks) * 64 + lane];
        const int q = 2 * ks + h;
        const int qq = q < Q ? q : 0;                 // (a padding block: its weights are zero, any in-bounds address will do)
        const int tap = qq / CBIN, cb = qq % CBIN;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        koff[ks] = (cb * K.Wp2 + dy * K.Wp + dx) * 8;
    }
    const int n_pos = ns * K.P;
    const int n_tiles = (n_pos + 31) / 32;
    for (int nt = wave; nt < n_tiles; nt += 4) {
        const int n = 32 * nt + col;
        const bool valid = n < n_pos;
        const int nn = valid ? n : 0;
        const int s = nn / K.P, p = nn - s * K.P;
        const int y = p / K.R, x = p - y * K.R;
        const int pidx = (y + 1) * K.Wp + (x + 1);
        const __bf16 *base = in + (size_t)s * in_sample_stride + pidx * 8;
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // the bias of this lane's rows (all lanes of a half read the same 16 bytes)
                const float4 bb = *(const float4 *)(bias + 32 * (mt_base + mt) + 8 * g + 4 * h);
                acc[mt][4 * g] = bb.x; acc[mt][4 * g + 1] = bb.y; acc[mt][4 * g + 2] = bb.z; acc[mt][4 * g + 3] = bb.w;
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 b = *(const bf16x8 *)(base + koff[ks]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], b, acc[mt], 0, 0, 0);
        }
        if (!valid) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (4 * mt + g >= COUT_BLOCKS) continue;   // conv1: 16 real output channels = blocks 0, 1
                const bf16x4 v = relu_pack(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
                if (TO_GLOBAL) *(bf16x4 *)(xg_tile + ((size_t)(s_local0 + s) * K.P + p) * 64 + 32 * (mt_base + mt) + 8 * g + 4 * h) = v;
                else *(bf16x4 *)(out + (size_t)s * out_sample_stride + ((4 * (mt_base + mt) + g) * K.Wp2 + pidx) * 8 + 4 * h) = v;
            }
    }
}

// a fully connected layer over the tile: M = 256 output features (8 row tiles), N = 128 samples (4 column tiles).
// Wavefront w: row tiles 4*(w&1) .. +3, column tiles 2*(w>>1), +1.  B fragments: 16 bytes at bsrc + sample*bstride + 16ks + 8h.
template <int KSTEPS_KNOWN>
__device__ __forceinline__ void fc_256(const bf16x8 *wfrag, int ksteps, const __bf16 *bsrc, size_t bstride, f32x16 (&acc)[4][2],
                                       int wave, int lane) {
    const int h = lane >> 5, col = lane & 31;
    const int mt0 = 4 * (wave & 1), nt0 = 2 * (wave >> 1);
    const __bf16 *b0 = bsrc + (size_t)(32 * nt0 + col) * bstride + 8 * h;
    const __bf16 *b1 = b0 + 32 * bstride;
    const bf16x8 *wa = wfrag + (size_t)mt0 * 64 + lane;
#pragma unroll 2
    for (int ks = 0; ks < ksteps; ++ks) {
        const bf16x8 x0 = *(const bf16x8 *)(b0 + 16 * ks), x1 = *(const bf16x8 *)(b1 + 16 * ks);
        bf16x8 w[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) w[m] = wa[((size_t)ks * 8 + m) * 64];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m], x0, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m], x1, acc[m][1], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void fc_init(const float *bias, f32x16 (&acc)[4][2], int wave, int lane) {
    const int h = lane >> 5, mt0 = 4 * (wave & 1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float b = bias[32 * (mt0 + m) + 8 * g + 4 * h + i];
                acc[m][0][4 * g + i] = b;
                acc[m][1][4 * g + i] = b;
            }
}

// ReLU(acc) -> H[sample][feature] (bf16, LDS)
__device__ __forceinline__ void fc_store(const f32x16 (&acc)[4][2], __bf16 *H, int wave, int lane) {
    const int h = lane >> 5, col = lane & 31;
    const int mt0 = 4 * (wave & 1), nt0 = 2 * (wave >> 1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(bf16x4 *)(H + (size_t)(32 * (nt0 + t) + col) * HSTRIDE + 32 * (mt0 + m) + 8 * g + 4 * h) =
                    relu_pack(acc[m][t][4 * g], acc[m][t][4 * g + 1], acc[m][t][4 * g + 2], acc[m][t][4 * g + 3]);
}

__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// KP: the parameter block read in place from the kernarg segment (scalar loads at the use sites) -- by value it would sit in
// ~100 SGPRs for the whole kernel and spill
template <bool OBS_F32, class KP>
__device__ __forceinline__ void policy_main(const KP &K, unsigned char *lds) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = (int)K.plan[0];
    // LDS: [tile table: 128 x {obs row address (8 B), action address (8 B)}] [activation images X, Y | H]
    unsigned long long *tab = (unsigned long long *)lds;
    __bf16 *img = (__bf16 *)(lds + TILE * 16);
    const int x_stride = 4 * K.Wp2 * 8, y_stride = 2 * K.Wp2 * 8;    // elements per sample
    __bf16 *X = img, *Y = img + (size_t)K.ST * x_stride, *H = img;
    __bf16 *xg_tile = K.xg + (size_t)blockIdx.x * TILE * K.K1;
    const int img_elems = K.ST * (x_stride + y_stride);

    for (int tile = (int)blockIdx.x; tile * TILE < N; tile += (int)gridDim.x) {
        const int n0 = tile * TILE;
        const int nt_samples = (N - n0) < TILE ? (N - n0) : TILE;
        __syncthreads();   // the previous tile's readers of H / tab are done
        // sample -> (handle, env, row): the last env whose prefix sum is <= n
        if (tid < TILE) {
            unsigned long long src = 0, dst = 0;
            if (tid < nt_samples) {
                const uint32_t n = (uint32_t)(n0 + tid);
                int lo = 0, hi = K.n_envs - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (K.plan[1 + mid] <= n) lo = mid; else hi = mid - 1;
                }
                const int e = lo, row = (int)(n - K.plan[1 + e]);
                const int k = handle_of(K.env_base, K.n_handles, e);
                const int b = e - K.env_base[k];
                src = (unsigned long long)(uintptr_t)(K.obs[k] + ((size_t)b * K.cap + row) * (size_t)(4 * K.P) * (OBS_F32 ? 4 : 8));
                dst = (unsigned long long)(uintptr_t)(K.actions[k] + (size_t)b * K.S + K.slot0 + row);
            }
            tab[2 * tid] = src;
            tab[2 * tid + 1] = dst;
        }
        // halo rings (and the unused channels of the input image) are zero and stay zero: only interiors are ever written
        for (int i = tid; i < img_elems / 8; i += 256) ((bf16x8 *)img)[i] = zero8();
        __syncthreads();

        // ---- A: convolutions, ST samples at a time ----
        for (int s0 = 0; s0 < nt_samples; s0 += K.ST) {
            const int ns = (nt_samples - s0) < K.ST ? (nt_samples - s0) : K.ST;
            for (int idx = tid; idx < ns * K.P; idx += 256) {
                const int s = idx / K.P, p = idx - s * K.P;
                const int y = p / K.R, x = p - y * K.R;
                bf16x8 v = zero8();
                if (OBS_F32) {
                    const float *src = (const float *)(uintptr_t)tab[2 * (s0 + s)];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = (__bf16)src[c * K.P + p];
                } else {
                    const double *src = (const double *)(uintptr_t)tab[2 * (s0 + s)];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = (__bf16)(float)src[c * K.P + p];
                }
                *(bf16x8 *)(X + (size_t)s * x_stride + ((y + 1) * K.Wp + (x + 1)) * 8) = v;
            }
            __syncthreads();
            conv_layer<1, 1, 2, false>(K, K.wc1, K.bc1, X, x_stride, Y, y_stride, nullptr, 0, ns, wave, lane);
            __syncthreads();
            conv_layer<2, 1, 4, false>(K, K.wc2, K.bc2, Y, y_stride, X, x_stride, nullptr, 0, ns, wave, lane);
            __syncthreads();
            // (two passes of 32 output channels each: 72 instead of 144 registers of weight fragments)
            conv_layer<4, 1, 8, true>(K, K.wc3, K.bc3, X, x_stride, nullptr, 0, xg_tile, s0, ns, wave, lane, 0);
            conv_layer<4, 1, 8, true>(K, K.wc3, K.bc3, X, x_stride, nullptr, 0, xg_tile, s0, ns, wave, lane, 1);
            __syncthreads();
        }
        // rows of the scratch slot behind the last sample of a partial tile hold older data: finite bf16 values whose columns
        // are never stored.  (The slot is zero-filled at creation, so they are never NaN patterns.)

        // ---- B: FC1.  This workgroup's stores to its scratch slot have to be re-read: drain them, meet, drop stale L1 lines ----
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 acc[4][2];
        fc_init(K.b1, acc, wave, lane);
        fc_256<0>(K.w1, K.K1 / 16, xg_tile, (size_t)K.K1, acc, wave, lane);
        fc_store(acc, H, wave, lane);      // (H overlays the activation images: the convolutions of this tile are finished)
        __syncthreads();
        // ---- C: FC2 from H, back into H ----
        fc_init(K.b2, acc, wave, lane);
        fc_256<0>(K.w2, 16, H, (size_t)HSTRIDE, acc, wave, lane);
        __syncthreads();
        fc_store(acc, H, wave, lane);
        __syncthreads();
        // ---- D: logits = W3 (actions padded to 32 rows) x H^T, one 32-sample column tile per wavefront ----
        {
            const int h = lane >> 5, col = lane & 31;
            f32x16 lg;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) lg[4 * g + i] = K.b3[8 * g + 4 * h + i];
            const __bf16 *hb = H + (size_t)(32 * wave + col) * HSTRIDE + 8 * h;
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks)
                lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(K.w3[ks * 64 + lane], *(const bf16x8 *)(hb + 16 * ks), lg, 0, 0, 0);
            // lane (h = 0) of a column holds actions 0-3, 8-11, 16-19, 24-27; its partner lane + 32 holds 4-7, 12-15, ...
            float mine[16], other[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { mine[r] = lg[r]; other[r] = __shfl_xor(mine[r], 32, 64); }
            const int s_local = 32 * wave + col;
            if (h == 0 && s_local < nt_samples) {
                float logit[32];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { logit[8 * g + i] = mine[4 * g + i]; logit[8 * g + 4 + i] = other[4 * g + i]; }
                if (K.logits) {
                    float *lo = K.logits + (size_t)(n0 + s_local) * K.n_actions;
#pragma unroll
                    for (int a = 0; a < 32; ++a) if (a < K.n_actions) lo[a] = logit[a];
                }
                int8_t *dst = (int8_t *)(uintptr_t)tab[2 * s_local + 1];
                uint32_t rnd[4] = {0, 0, 0, 0};
                int best = 0;
                float bestv = -INFINITY;
#pragma unroll
                for (int a = 0; a < 32; ++a) {
                    if (a >= K.n_actions) continue;
                    float v = logit[a];
                    if (K.sample) {   // Gumbel-max: argmax(logit - log(-log u)) ~ softmax(logits)
                        if ((a & 3) == 0) philox((uint32_t)(uintptr_t)dst, (uint32_t)((uintptr_t)dst >> 32), (uint32_t)(a >> 2), 0x504F4C31u,
                                                 K.seed_lo, K.seed_hi, rnd);
                        const float u = ((float)(rnd[a & 3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                        v -= __logf(-__logf(u));
                    }
                    if (v > bestv) { bestv = v; best = a; }
                }
                *dst = (int8_t)best;
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(256, 2) ppg_policy_forward_f64(const PolParams K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    policy_main<false>(*(const __attribute__((address_space(4))) PolParams *)__builtin_amdgcn_kernarg_segment_ptr(), lds);
}
extern "C" __global__ void __launch_bounds__(256, 2) ppg_policy_forward_f32(const PolParams K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    policy_main<true>(*(const __attribute__((address_space(4))) PolParams *)__builtin_amdgcn_kernarg_segment_ptr(), lds);
}

}  // namespace ppgpol

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------

struct ppg_policy {
    int32_t device, R, n_actions;
    ppgpol::PolParams base;
    void *dev_weights;     // one allocation: fragments + biases
    __bf16 *xg;            // scratch slots
    uint32_t *plan;        // [1 + plan_envs]
    int32_t plan_envs;
    int32_t grid;
    int32_t lds_bytes;
    char err[256];
};

static char g_ppg_policy_error[256] = "";

static int ppg_policy_fail(ppg_policy *p, int code, const char *fmt, ...) {
    char *dst = p ? p->err : g_ppg_policy_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}

#define PPG_POL_TRY(p, call)                                                                              \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return ppg_policy_fail(p, PPG_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static uint16_t ppg_bf16_bits(float f) {   // round to nearest even (finite weights)
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// conv weights [cout][cin][3][3] -> fragments [mt][ks][lane][8]: lane (r = lane & 31, h = lane >> 5) holds, for K block
// q = 2 ks + h = tap * CBIN + cb, the eight input channels 8 cb .. 8 cb + 7 of output channel 32 mt + r at tap (ky, kx)
static void ppg_pack_conv(const float *w, int cout, int cin, int cbin, int mt_n, std::vector<uint16_t> &out) {
    const int Q = 9 * cbin, KS = (Q + 1) / 2;
    out.assign((size_t)mt_n * KS * 64 * 8, 0);
    for (int mt = 0; mt < mt_n; ++mt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, h = lane >> 5, q = 2 * ks + h, co = 32 * mt + r;
                if (q >= Q || co >= cout) continue;
                const int tap = q / cbin, cb = q % cbin;
                for (int j = 0; j < 8; ++j) {
                    const int ci = 8 * cb + j;
                    if (ci < cin) out[(((size_t)mt * KS + ks) * 64 + lane) * 8 + j] = ppg_bf16_bits(w[((size_t)co * cin + ci) * 9 + tap]);
                }
            }
}

// Linear.weight [n_out][K] -> fragments [ks][mt][lane][8]: lane holds output feature 32 mt + r, inputs kmap(16 ks + 8 h + j)
template <class KMap>
static void ppg_pack_fc(const float *w, int n_out, int K, int mt_n, KMap kmap, std::vector<uint16_t> &out) {
    const int KS = K / 16;
    out.assign((size_t)KS * mt_n * 64 * 8, 0);
    for (int ks = 0; ks < KS; ++ks)
        for (int mt = 0; mt < mt_n; ++mt)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, h = lane >> 5, o = 32 * mt + r;
                if (o >= n_out) continue;
                for (int j = 0; j < 8; ++j)
                    out[(((size_t)ks * mt_n + mt) * 64 + lane) * 8 + j] = ppg_bf16_bits(w[(size_t)o * K + kmap(16 * ks + 8 * h + j)]);
            }
}

extern "C" {

int ppg_policy_create(int32_t device, int32_t obs_range, int32_t n_actions, const ppg_policy_weights *w, ppg_policy **out) {
    if (!w || !out) return ppg_policy_fail(nullptr, PPG_EINVAL, "null argument");
    if (obs_range < 1 || obs_range > 15) return ppg_policy_fail(nullptr, PPG_EINVAL, "obs_range %d outside 1..15", obs_range);
    if (n_actions < 1 || n_actions > 32) return ppg_policy_fail(nullptr, PPG_EINVAL, "n_actions %d outside 1..32", n_actions);
    for (int l = 0; l < 3; ++l)
        if (!w->conv_w[l] || !w->conv_b[l] || !w->fc_w[l] || !w->fc_b[l]) return ppg_policy_fail(nullptr, PPG_EINVAL, "a weight pointer is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return ppg_policy_fail(nullptr, PPG_ENODEV, "device %d not available", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ppg_policy_fail(nullptr, PPG_ENODEV, "ppg_policy needs a gfx950 device (MI355X)");
    ppg_policy *p = new (std::nothrow) ppg_policy();
    if (!p) return ppg_policy_fail(nullptr, PPG_ENOMEM, "out of host memory");
    memset(p, 0, sizeof *p);
    p->device = device; p->R = obs_range; p->n_actions = n_actions;
    const int R = obs_range, P = R * R, K1 = 64 * P;
    std::vector<uint16_t> f[6];
    ppg_pack_conv(w->conv_w[0], 16, 4, 1, 1, f[0]);
    ppg_pack_conv(w->conv_w[1], 32, 16, 2, 1, f[1]);
    ppg_pack_conv(w->conv_w[2], 64, 32, 4, 2, f[2]);
    // FC1: our K order is position-major (k = p * 64 + c, the layout conv3 writes); PyTorch flattens channel-major (c * P + p)
    ppg_pack_fc(w->fc_w[0], 256, K1, 8, [P](int k) { return (k % 64) * P + k / 64; }, f[3]);
    ppg_pack_fc(w->fc_w[1], 256, 256, 8, [](int k) { return k; }, f[4]);
    ppg_pack_fc(w->fc_w[2], n_actions, 256, 1, [](int k) { return k; }, f[5]);
    std::vector<float> bias(32 + 32 + 64 + 256 + 256 + 32, 0.0f);
    const int boff[6] = {0, 32, 64, 128, 384, 640};
    const int bn[6] = {16, 32, 64, 256, 256, n_actions};
    for (int l = 0; l < 3; ++l) for (int i = 0; i < bn[l]; ++i) bias[boff[l] + i] = w->conv_b[l][i];
    for (int l = 0; l < 3; ++l) for (int i = 0; i < bn[3 + l]; ++i) bias[boff[3 + l] + i] = w->fc_b[l][i];
    size_t off[7], total = 0;
    for (int l = 0; l < 6; ++l) { off[l] = total; total += (f[l].size() * 2 + 255) / 256 * 256; }
    off[6] = total; total += bias.size() * 4;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p->dev_weights, total) != hipSuccess) {
        delete p;
        return ppg_policy_fail(nullptr, PPG_EHIP, "hipMalloc of %zu bytes of weights failed", total);
    }
    std::vector<unsigned char> stage(total, 0);
    for (int l = 0; l < 6; ++l) memcpy(stage.data() + off[l], f[l].data(), f[l].size() * 2);
    memcpy(stage.data() + off[6], bias.data(), bias.size() * 4);
    if (hipMemcpy(p->dev_weights, stage.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->dev_weights);
        delete p;
        return ppg_policy_fail(nullptr, PPG_EHIP, "upload of the weights failed");
    }
    ppgpol::PolParams &K = p->base;
    K.R = R; K.P = P; K.Wp = R + 2; K.Wp2 = (R + 2) * (R + 2); K.K1 = K1; K.n_actions = n_actions;
    const unsigned char *dw = (const unsigned char *)p->dev_weights;
    K.wc1 = (const ppgpol::bf16x8 *)(dw + off[0]); K.wc2 = (const ppgpol::bf16x8 *)(dw + off[1]);
    K.wc3 = (const ppgpol::bf16x8 *)(dw + off[2]); K.w1 = (const ppgpol::bf16x8 *)(dw + off[3]);
    K.w2 = (const ppgpol::bf16x8 *)(dw + off[4]); K.w3 = (const ppgpol::bf16x8 *)(dw + off[5]);
    const float *db = (const float *)(dw + off[6]);
    K.bc1 = db + boff[0]; K.bc2 = db + boff[1]; K.bc3 = db + boff[2]; K.b1 = db + boff[3]; K.b2 = db + boff[4]; K.b3 = db + boff[5];
    // LDS: tile table + max(activation images of ST samples, H); ST as large as 2 workgroups per CU (80 KB each) allow
    const int per_sample = 6 * K.Wp2 * 8 * 2;                       // X (4 blocks) + Y (2 blocks), bytes
    const int h_bytes = ppgpol::TILE * ppgpol::HSTRIDE * 2;
    int st = (78 * 1024 - ppgpol::TILE * 16) / per_sample;
    if (st < 1) st = 1;
    if (st > 16) st = 16;
    K.ST = st;
    const int img = st * per_sample;
    p->lds_bytes = ppgpol::TILE * 16 + (img > h_bytes ? img : h_bytes);
    p->grid = 2 * prop.multiProcessorCount;
    const size_t xg_bytes = (size_t)p->grid * ppgpol::TILE * K1 * 2;
    if (hipMalloc((void **)&p->xg, xg_bytes) != hipSuccess || hipMemset(p->xg, 0, xg_bytes) != hipSuccess) {
        (void)hipFree(p->dev_weights);
        delete p;
        return ppg_policy_fail(nullptr, PPG_EHIP, "hipMalloc of %zu bytes of scratch failed", xg_bytes);
    }
    K.xg = p->xg;
    (void)hipFuncSetAttribute((const void *)ppgpol::ppg_policy_forward_f64, hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
    (void)hipFuncSetAttribute((const void *)ppgpol::ppg_policy_forward_f32, hipFuncAttributeMaxDynamicSharedMemorySize, p->lds_bytes);
    *out = p;
    return PPG_OK;
}

int ppg_policy_destroy(ppg_policy *p) {
    if (!p) return PPG_OK;
    if (p->dev_weights) (void)hipFree(p->dev_weights);
    if (p->xg) (void)hipFree(p->xg);
    if (p->plan) (void)hipFree(p->plan);
    delete p;
    return PPG_OK;
}

uint64_t ppg_policy_macs_per_observation(const ppg_policy *p) {
    if (!p) return 0;
    const uint64_t P = (uint64_t)p->R * p->R;
    return P * (16 * 36 + 32 * 144 + 64 * 288) + 64 * P * 256 + 256 * 256 + 256 * (uint64_t)p->n_actions;
}

const char *ppg_policy_last_error(const ppg_policy *p) { return p ? p->err : g_ppg_policy_error; }

static int ppg_policy_run(ppg_policy *p, int species, ppg_handle *const *handles, int32_t n, int8_t *const *actions, uint32_t flags,
                          uint64_t seed, float *logits, void *stream) {
    ppg_handle *h0 = handles[0];
    const int R = species ? h0->base.Rq : h0->base.Rp;
    if (R != p->R) return ppg_policy_fail(p, PPG_EINVAL, "the policy was created for %dx%d observations, the %s observe %dx%d", p->R, p->R,
                                          species ? "prey" : "predators", R, R);
    if (p->n_actions != 9 && !h0->gen2) return ppg_policy_fail(p, PPG_EINVAL, "the base env has 9 actions, the policy %d", p->n_actions);
    ppgpol::PolParams K = p->base;
    ppgpol::PlanParams L;
    memset(&L, 0, sizeof L);
    K.species = species; K.obs_f32 = h0->base.obs_f32; K.sample = (flags & PPG_POLICY_SAMPLE) ? 1 : 0;
    K.seed_lo = (uint32_t)seed ^ (species ? 0x9E3779B9u : 0u); K.seed_hi = (uint32_t)(seed >> 32);
    K.S = h0->base.S; K.cap = species ? h0->base.cap_prey : h0->base.cap_pred; K.slot0 = species ? h0->base.cap_pred : 0;
    K.n_handles = n; L.n_handles = n;
    L.word = species ? PPG_ENV_N_PREY_ROWS : PPG_ENV_N_PRED_ROWS;
    int total = 0;
    for (int k = 0; k < n; ++k) {
        const ppg_handle *h = handles[k];
        if (h->drive || (h->gen2 && h->cfg2.walls && h->cfg2.include_visibility_channel))
            return ppg_policy_fail(p, PPG_EINVAL, "ppg_policy_act expects 4-channel observations");
        if ((species ? h->base.Rq : h->base.Rp) != R || h->base.S != K.S || h->base.obs_f32 != K.obs_f32 || h->device != p->device)
            return ppg_policy_fail(p, PPG_EINVAL, "handle %d has another geometry / dtype / device than handle 0", k);
        if (!actions[k]) return ppg_policy_fail(p, PPG_EINVAL, "actions[%d] is NULL", k);
        K.env_base[k] = L.env_base[k] = total;
        total += h->batch;
        K.env_state[k] = L.env_state[k] = h->bufs.env_state;
        K.obs[k] = (const unsigned char *)(species ? h->bufs.obs_prey : h->bufs.obs_pred);
        K.actions[k] = actions[k];
    }
    for (int k = n; k <= ppgpol::MAX_HANDLES; ++k) K.env_base[k] = L.env_base[k] = total;
    K.n_envs = L.n_envs = total;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != p->device) PPG_POL_TRY(p, hipSetDevice(p->device));
    if (p->plan_envs < total) {
        if (p->plan) (void)hipFree(p->plan);
        p->plan = nullptr;
        PPG_POL_TRY(p, hipMalloc((void **)&p->plan, (size_t)(1 + total) * 4));
        p->plan_envs = total;
    }
    K.plan = L.plan = p->plan;
    K.logits = logits;
    hipLaunchKernelGGL(ppgpol::ppg_policy_plan, dim3(1), dim3(64), 0, (hipStream_t)stream, L);
    if (K.obs_f32) hipLaunchKernelGGL(ppgpol::ppg_policy_forward_f32, dim3((unsigned)p->grid), dim3(256), (size_t)p->lds_bytes, (hipStream_t)stream, K);
    else hipLaunchKernelGGL(ppgpol::ppg_policy_forward_f64, dim3((unsigned)p->grid), dim3(256), (size_t)p->lds_bytes, (hipStream_t)stream, K);
    PPG_POL_TRY(p, hipGetLastError());
    return PPG_OK;
}

int ppg_policy_act(ppg_policy *pred, ppg_policy *prey, ppg_handle *const *handles, int32_t n, int8_t *const *actions,
                   uint32_t flags, uint64_t seed, float *logits_pred, float *logits_prey, void *stream) {
    ppg_policy *any = pred ? pred : prey;
    if (!any) return PPG_EINVAL;
    if (!handles || !actions || n < 1 || n > ppgpol::MAX_HANDLES || !handles[0]) return ppg_policy_fail(any, PPG_EINVAL, "bad handle list");
    if (flags & ~PPG_POLICY_SAMPLE) return ppg_policy_fail(any, PPG_EINVAL, "unknown policy flags 0x%x", flags);
    // (a failure is also reported through ppg_policy_last_error(NULL), whichever of the two policies it came from)
    if (pred) {
        const int rc = ppg_policy_run(pred, 0, handles, n, actions, flags, seed, logits_pred, stream);
        if (rc != PPG_OK) { memcpy(g_ppg_policy_error, pred->err, sizeof g_ppg_policy_error); return rc; }
    }
    if (prey) {
        const int rc = ppg_policy_run(prey, 1, handles, n, actions, flags, seed, logits_prey, stream);
        if (rc != PPG_OK) { memcpy(g_ppg_policy_error, prey->err, sizeof g_ppg_policy_error); return rc; }
    }
    return PPG_OK;
}

}  // extern "C"
