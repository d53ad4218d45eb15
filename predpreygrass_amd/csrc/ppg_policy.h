// ppg_policy.h -- policy inference next to the env (include/ppg.h: ppg_policy_*).
//
// The network the reference trains for each species (base_environment/tune_ppo_base_environment.py:106-141):
//     (4,R,R) observation -> conv3x3(C->16) ReLU -> conv3x3(16->32) ReLU -> conv3x3(32->64) ReLU   ("same" padding, stride 1)
//                         -> flatten -> Linear(64 P -> 256) ReLU -> Linear(256 -> 256) ReLU -> Linear(256 -> n_actions)
// in either of the two ways a (4,R,R) Box can be read as an image (include/ppg.h: PPG_POLICY_LAYOUT_*): channel-first -- an R x R
// image with C = 4 channels, P = R^2 positions -- or channels-last, which is how RLlib's CNN encoder reads a 3-D Box ([H, W, C]:
// a 4 x R image with C = R channels, P = 4 R positions).  The kernels work on an IH x IW image with CIN channels;
// evaluated for every agent row in use, reading the observation rows where ppg_step wrote them and writing one int8 action per
// row.  gfx950 only: every layer is a GEMM on the matrix cores, v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate).
//
// ORIENTATION (all six layers): M = output features / channels (the A operand = weights), N = samples or positions (the B
// operand = activations).  In the 32x32 result a lane holds ONE column (sample / position, = lane & 31) and 16 of the 32 rows:
// register 4g+i of lane half h = lane >> 5 is MFMA row i + 8g + 4h.  Which output feature an MFMA row computes is free -- it is
// decided by the order in which the host packs the weight rows -- so row i + 8g + 4h is given feature 16h + 4g + i of its
// 32-feature tile: a lane's 16 registers are then 16 CONSECUTIVE features (32 bytes of bf16), written by two 16-byte stores
// into an activation image whose inner dimension is the feature index -- exactly the 16-byte-per-lane B fragment the next
// layer reads (k = 8h + j: eight consecutive features).
//
// ONE persistent launch per species.  A workgroup (4 wavefronts) takes tiles of 128 samples:
//   A. convolutions in sub-groups of ST samples: activations live in LDS as [sample][channel block of 8][padded position][8]
//      bf16 with a zero halo ring (so the nine taps of the implicit GEMM are plain offsets); each wavefront keeps the layer's
//      weight fragments in registers (conv3: 144 VGPRs) and walks over 32-position tiles.  conv3 writes its output to the
//      workgroup's scratch slot in HBM/L2 as X[sample][row tile][position][32]  (= the K order the repacked FC1 weights expect).
//   B. FC1 over the whole tile (K = 64 R^2): weight fragments stream from L2, X fragments from the scratch slot (the slot was
//      written by this workgroup and is re-read after a workgroup barrier + ONE agent-scope acquire that drops stale L1 lines);
//      ReLU -> H[sample][256] in LDS.   C. FC2 from H, result back into H.   D. logits (M = actions padded to 32), the two
//      lane halves exchange their rows, argmax or Gumbel-max sampling, int8 store into actions[b][slot].
// The sample list is never materialised: a one-wavefront plan kernel writes exclusive prefix sums of the per-env row counts and
// each tile resolves sample -> (handle, env, row) by bisection.
#pragma once

#include <hip/hip_runtime.h>

#include <math.h>

#include <type_traits>
#include <vector>

namespace ppgpol {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))   // pointers known to be global memory: global_* instead of flat_* instructions

constexpr int TILE = 128;         // samples per workgroup tile
constexpr int HSTRIDE = 264;      // H[sample][256 + 8] bf16: rows 528 B apart -> conflict-free 16-byte fragment reads
constexpr int MAX_HANDLES = PPG_PACK_MAX_HANDLES;
constexpr int PLAN_HDR = 3;       // words in front of the prefix sums of PolParams::plan

struct PolParams {
    // geometry
    int32_t R, P, Wp, Wp2;        // window side of the observation; image positions IH*IW; padded row pitch IW+2; (IH+2)*(IW+2)
    int32_t IH, IW, cin;          // the image the convolutions run on and its real input channels
    int32_t c_stride, p_stride;   // observation element of (channel c, position p) = c * c_stride + p * p_stride
    int32_t obs_elems;            // elements per observation row: 4 R^2
    int32_t K1;                   // 64 * P
    int32_t ST;                   // samples per convolution sub-group
    int32_t n_actions;
    int32_t species;              // 0 predators, 1 prey
    int32_t obs_f32;
    int32_t sample;               // PPG_POLICY_SAMPLE
    int32_t debug_skip;           // ablation builds only (-DPPG_EXPERIMENTS + env PPG_POLICY_SKIP; always 0 in the product): 1 no convolutions,
                                  // 2 no FC1, 4 no observation staging, 8 no conv3, 16 no conv1/conv2, 32 no conv3 stores
    uint32_t seed_lo, seed_hi;
    const uint64_t *seed_dev;     // PPG_POLICY_SEED_ON_DEVICE: the key is read from here by the kernel (seed_lo / seed_hi unused)
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
    const uint32_t *plan;         // [0] = total rows of this species, [1] = number of FULL tiles (128 samples, whole rounds of the
                                  // resident workgroups), [2] = samples per tile of the last round (32 / 64 / 96 / 128),
                                  // [PLAN_HDR + e] = exclusive prefix sum of env e
    const uint32_t *tile_env;     // [tile] = env of the tile's first sample
    uint32_t magic_P, magic_R;    // ceil(2^18 / P), ceil(2^18 / IW): div_small()
    __bf16 *xg;                   // [gridDim.x][TILE][K1]
    float *logits;                // optional [rows][n_actions]
    // hidden layers of the head (phase_head): 2 = FC1, FC2, logits; 1 = FC1, logits
    int32_t n_hidden;
    // ---- direct-head networks (ppg_policy_direct.h) ----
    int32_t n_conv;               // convolution layers
    int32_t cout_blocks[PPG_POLICY_MAX_CONV];   // real output channel blocks (of 8) per layer
    int32_t off_x, off_y, off_f, off_d0, off_d1;   // element offsets of the areas in a sample's LDS region
    int32_t sample_stride;        // elements per sample region
    int32_t flat_c;               // channels per position of area F (8 * cout_blocks[n_conv - 1])
    int32_t kflat_steps;          // k-steps of 32 features of the head: ceil(P * flat_c / 32)
    int32_t head_mt;              // 16-row tiles of actions: 1 or 2
    const bf16x8 *wcd[PPG_POLICY_MAX_CONV - 3];   // fragments of the convolutions behind the third (64 -> 64 channels)
    const bf16x8 *wh;             // head fragments [action tile][k-step][lane]
    const bf16x8 *whw;            // the same of action tile 0 per wavefront: [wavefront][18][lane], zeros behind a wavefront's share of k-steps
    const float *bh;              // head bias [32]
    float *lgs;                   // library-owned [gridDim.x][TILE][16 head_mt]: a workgroup's logits of its current tile
    int32_t range_tile;           // samples per tile inside a workgroup's share (PlanParams::range_tile)
    // the two-role pipeline (ppg_policy_pipe.h): element offsets of the second X / F area in a sample's region; byte offsets of the
    // partial-sum buffers (+ role B's barrier counter) and of the images in LDS
    int32_t pipe_x1, pipe_f1, pipe_red, pipe_img;
    int32_t pipe_raw, pipe_ni;    // bfloat16 rows of a multiple of 8 bytes: byte offset of the LDS area that takes a sub-group's rows as they
    uint32_t pipe_magic;          // lie in HBM, 8-byte loads per role-B thread and sub-group (0: one load per channel), ceil(2^32 / chunks per row),
    int32_t pipe_slots;           // samples whose chunks the 256 role-B threads cover at once: floor(256 / chunks per row)
    const uint16_t *slot_tab;     // the pipeline's slot table [32 slot_tiles]: sample << 8 | position of a sub-group's slot (0xFFFF: none);
    int32_t slot_tiles;           // ppg_slot_table
    const bf16x8 *wc1x;           // conv1 of the pipeline as three 16x16x32 fragments [ky][lane][8] (ppg_policy_pipe.h: Conv1X); its bias [16]
    const float *bc1x;
#ifdef PPG_EXPERIMENTS
    unsigned long long *timeline; // diagnostic builds: [tile][64] = workgroup, hardware id, samples, 4 wall-clock stamps (10 ns units); [8 + 12 wave + i] cycles of wave in step i of the convolutions, [56 + 2 wave + i] FC1 wait / barrier cycles
#endif
};

struct PlanParams {
    int32_t n_handles, n_envs, word;   // word: PPG_ENV_N_PRED_ROWS / PPG_ENV_N_PREY_ROWS
    int32_t slots;                     // workgroups the forward launch keeps resident (its grid)
    int32_t force_ts;                  // experiments (env PPG_POLICY_TILE_PREY / _PRED): every tile 32 / 64 / 96 / 128 samples; 0 = choose
    int32_t range_tile, range_st;      // direct-head kernels: > 0 = RANGE MODE -- every workgroup gets one contiguous share of the samples (a
                                       // multiple of range_st = its sub-group size), cut into tiles of range_tile samples (a multiple too)
    int32_t env_base[MAX_HANDLES + 1];
    const int32_t *env_state[MAX_HANDLES];
    uint32_t *plan;
    uint32_t *tile_env;
};

template <class P>
__device__ __forceinline__ int handle_of(P env_base, int n_handles, int e) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < MAX_HANDLES; ++q) k += (q < n_handles && e >= env_base[q]) ? 1 : 0;
    return k;
}

// Exclusive prefix sums of one env_state word over the concatenated envs of all handles, and for every 128-sample tile the env
// its first sample belongs to.  One workgroup of 1024 threads: thread t sums a contiguous run of envs, the 1024 partial sums are
// scanned in LDS, then every thread bisects for its tiles.
// LDS_SUMS = false (ppg_policy_plan_small, the FC-chain policies): the per-env sums go through memory only and the kernel needs 4 KB of
// LDS -- those policies' forward kernels fill every CU to within 4 KB (two workgroups of 78 KB), and the OTHER species' plan runs on
// a side stream beside them: with 36 KB it found no CU to run on until a forward workgroup finished (fc256: 0.80 -> 0.89 ms per step).
template <bool LDS_SUMS>
__device__ __forceinline__ void plan_body(const PlanParams &K) {
    __shared__ uint32_t part[1024];
    // Up to PLAN_LDS_ENVS envs the per-env prefix sums stay in LDS as well and a thread keeps its run's counts in registers: ONE trip to
    // memory per launch (the counts) instead of four dependent ones (counts, counts again, two bisection steps over the sums just
    // written) -- the plan is a 1024-thread chain in front of every policy step (14 -> 9 us at 4096 envs).
    constexpr int PLAN_LDS_ENVS = 8192, PLAN_RUN = PLAN_LDS_ENVS / 1024;
    __shared__ uint32_t env_pre[LDS_SUMS ? PLAN_LDS_ENVS : 1];
    const int t = (int)threadIdx.x;
    const int per = (K.n_envs + 1023) / 1024;
    const bool in_lds = LDS_SUMS && K.n_envs <= PLAN_LDS_ENVS;
    const int lo = t * per < K.n_envs ? t * per : K.n_envs, hi = (lo + per) < K.n_envs ? (lo + per) : K.n_envs;
    // (the handle's env_state pointer by scalar comparisons: indexed with a per-lane handle number it is a vector load from the
    //  parameter block, one more dependent trip)
    auto count_of = [&](int e) -> uint32_t {
        const int32_t *base = K.env_state[0];
        int eb = 0;
#pragma unroll
        for (int q = 1; q < MAX_HANDLES; ++q) {
            const bool in = q < K.n_handles && e >= K.env_base[q];
            base = in ? K.env_state[q] : base;
            eb = in ? K.env_base[q] : eb;
        }
        return (uint32_t)base[(size_t)(e - eb) * PPG_ENV_WORDS + K.word];
    };
    uint32_t cnt[PLAN_RUN];
    uint32_t s = 0;
    if (in_lds) {
#pragma unroll
        for (int i = 0; i < PLAN_RUN; ++i) cnt[i] = (i < per && lo + i < hi) ? count_of(lo + i) : 0u;
#pragma unroll
        for (int i = 0; i < PLAN_RUN; ++i) s += cnt[i];
    } else {
        for (int e = lo; e < hi; ++e) s += count_of(e);
    }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // inclusive Hillis-Steele scan
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t before = part[t] - s;
    const uint32_t all = part[1023];
    if (in_lds) {
#pragma unroll
        for (int i = 0; i < PLAN_RUN; ++i)
            if (i < per && lo + i < hi) {
                K.plan[PLAN_HDR + lo + i] = before;
                env_pre[lo + i] = before;
                before += cnt[i];
            }
    } else {
        for (int e = lo; e < hi; ++e) {
            K.plan[PLAN_HDR + e] = before;
            before += count_of(e);
        }
    }
    // exclusive prefix sum of env `e` as the bisections below read it
    auto prefix_of = [&](int e) -> uint32_t { return in_lds ? env_pre[e] : __builtin_nontemporal_load(&K.plan[PLAN_HDR + e]); };
    if (K.range_tile > 0) {
        // RANGE MODE: workgroup w owns samples [w S, (w + 1) S), S = the per-workgroup share rounded up to whole sub-groups, as tiles of
        // range_tile samples -- every workgroup the same number of FULL sub-groups (whole rounds of 128-sample tiles + a short last round
        // leave most of the chip idle in the last round and cut a two-sample sub-group off every tile: 7 % at the benchmark's 132 k prey)
        const uint32_t st = (uint32_t)K.range_st, slots = (uint32_t)K.slots, tsz = (uint32_t)K.range_tile;
        const uint32_t S = st * ((all + slots * st - 1u) / (slots * st));
        const uint32_t tpw = S ? (S + tsz - 1u) / tsz : 1u;
        if (t == 0) { K.plan[0] = all; K.plan[1] = S; K.plan[2] = tpw; }
        __threadfence_block();
        __syncthreads();
        const int n_slots = (int)(slots * tpw);
        for (int tile = t; tile < n_slots; tile += 1024) {
            uint32_t n = ((uint32_t)tile / tpw) * S + ((uint32_t)tile % tpw) * tsz;
            if (all == 0u) { K.tile_env[tile] = 0u; continue; }
            if (n > all - 1u) n = all - 1u;    // (an empty tile slot: behind every sample of the tiles in front of it -- keeps the list monotone)
            int ua = 0, ub = 1023;
            while (ua < ub) {
                const int mid = (ua + ub + 1) >> 1;
                if (part[mid - 1] <= n) ua = mid; else ub = mid - 1;
            }
            int a = ua * per < K.n_envs ? ua * per : K.n_envs - 1, b = (a + per - 1) < (K.n_envs - 1) ? (a + per - 1) : (K.n_envs - 1);
            while (a < b) {
                const int mid = (a + b + 1) >> 1;
                if (prefix_of(mid) <= n) a = mid; else b = mid - 1;
            }
            K.tile_env[tile] = (uint32_t)a;
        }
        return;
    }
    // Tiles: as many whole ROUNDS of 128-sample tiles as the resident workgroups (slots) can be given (the FC1 weights are re-read
    // once per tile, so tiles are as large as the accumulators allow), then ONE last round in which the remaining samples are
    // spread over the slots in tiles of 32 / 64 / 96 / 128.  A last round of a few full tiles would take as long as any other
    // round while most of the chip idles: at the benchmark's 134 k prey rows, 1047 tiles of 128 on 512 slots are 2.04 rounds and
    // cost three; 1024 + 92 tiles of 32 cost two and a short one.
    const uint32_t per_round = (uint32_t)TILE * (uint32_t)K.slots;
    uint32_t n_full = (all / per_round) * (uint32_t)K.slots;
    const uint32_t rest = all - n_full * (uint32_t)TILE;
    uint32_t ts = 32u * ((rest + 32u * (uint32_t)K.slots - 1u) / (32u * (uint32_t)K.slots));   // 32 * ceil(rest / (32 * slots))
    if (ts < 32u) ts = 32u;
    if (K.force_ts) { ts = (uint32_t)K.force_ts; n_full = 0; }
    if (t == 0) { K.plan[0] = all; K.plan[1] = n_full; K.plan[2] = ts; }
    __threadfence_block();
    __syncthreads();   // (the prefix sums are re-read below by other threads of this workgroup: same CU, written through L1)
    const uint32_t tail = all - n_full * (uint32_t)TILE;
    const int n_tiles = (int)(n_full + (tail + ts - 1) / ts);
    // exclusive prefix of thread u's run = part[u] - (its own sum) = part[u - 1]: the bisection first finds the RUN in LDS (ten steps, no
    // memory round trip), then the env inside the run (`per` <= a handful of prefix sums from memory)
    for (int tile = t; tile < n_tiles; tile += 1024) {
        const uint32_t n = (uint32_t)tile < n_full ? (uint32_t)tile * (uint32_t)TILE : n_full * (uint32_t)TILE + ((uint32_t)tile - n_full) * ts;
        int ua = 0, ub = 1023;
        while (ua < ub) {   // the last run whose first env's prefix sum (= the inclusive sum of the runs before it) is <= n
            const int mid = (ua + ub + 1) >> 1;
            if (part[mid - 1] <= n) ua = mid; else ub = mid - 1;
        }
        int a = ua * per < K.n_envs ? ua * per : K.n_envs - 1, b = (a + per - 1) < (K.n_envs - 1) ? (a + per - 1) : (K.n_envs - 1);
        while (a < b) {   // the last env of that run whose prefix sum is <= n
            const int mid = (a + b + 1) >> 1;
            if (prefix_of(mid) <= n) a = mid; else b = mid - 1;
        }
        K.tile_env[tile] = (uint32_t)a;
    }
}
extern "C" __global__ void __launch_bounds__(1024) ppg_policy_plan(const PlanParams K) { plan_body<true>(K); }
extern "C" __global__ void __launch_bounds__(1024) ppg_policy_plan_small(const PlanParams K) { plan_body<false>(K); }
// both species' plans in one launch (workgroup 0: A, workgroup 1: B): one launch and one dependent-load chain less per step
extern "C" __global__ void __launch_bounds__(1024) ppg_policy_plan2(const PlanParams A, const PlanParams B) {
    if (blockIdx.x == 0) plan_body<true>(A); else plan_body<true>(B);
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
    return v;
}

// ReLU + round 8 accumulator registers to bf16: round first (v_cvt_pk_bf16_f32, two values per instruction), then ReLU on the PACKED
// pair -- a bf16 with the sign bit set is a negative 16-bit integer, so one v_pk_max_i16 with 0 clears both halves.  (Same values
// as ReLU before rounding: a negative number rounds to a negative number or -0, a positive one is untouched.)  12 instructions per
// 8 values instead of 8 v_max + 4 conversions... and half the v_max: the epilogues are a fifth of the convolutions' time.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 relu_pack8(const f32x16 &a, int r0) {
    u32x4_t w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 f = {a[r0 + 2 * j], a[r0 + 2 * j + 1]};
        const s16x2 zero = {0, 0};
        const s16x2 q = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2)), zero);
        w[j] = __builtin_bit_cast(uint32_t, q);
    }
    return __builtin_bit_cast(bf16x8, w);
}

// n / d for 0 <= n < 1024, 1 <= d <= 255 with ONE full-rate instruction pair: (n * ceil(2^18 / d)) >> 18 (v_mul_u32_u24; the 32-bit
// multiplies and v_mul_hi the position arithmetic used before are quarter rate).  Exact: the error term n * (d ceil(2^18 / d) - 2^18)
// stays below 2^18.  The other products of the position arithmetic are 24-bit multiplies too.
constexpr int DIV_SHIFT = 18;
__device__ __forceinline__ int div_small(int n, uint32_t magic) { return (int)(__umul24((uint32_t)n, magic) >> DIV_SHIFT); }

// The weight fragments of one convolution layer and, per k-step, the offset of this lane half's K block: in registers.
//   CBIN   channel blocks (of 8) of the input image: 1 (4 real channels), 2, 4        K = 9 taps x CBIN blocks
//   MT     32-row tiles of output channels: 1 (16 real for conv1 / 32 for conv2), 2 (conv3)
// The bias rides in the GEMM: K block Q (the first one behind the 9 taps x CBIN channel blocks) holds the bias, split into a bf16
// head and a bf16 remainder, against a constant B fragment {1, 1, 0, ...} -- no bias loads per position tile and no registers to
// keep it in.  For CBIN = 1 that block is the second half of the last tap's k-step (free); for CBIN = 2 / 4 it costs one more MFMA
// per tile (10 instead of 9 / 19 instead of 18).
template <int CBIN, int MT>
struct ConvW {
    static constexpr int Q = 9 * CBIN, KS = (Q + 2) / 2;       // k-steps incl. the bias block
    static constexpr int KS_BIAS = Q / 2, H_BIAS = Q & 1;      // the bias block is block Q = 2 * KS_BIAS + H_BIAS
    bf16x8 a[MT][KS];
    int koff1[CBIN == 1 ? KS : 1];   // CBIN == 1 only: the tap of a k-step depends on the lane half -> per-lane offsets
    template <class KP>
    __device__ __forceinline__ void load(const KP &K, const bf16x8 *wfrag, int lane, int mt_base = 0) {
        const int h = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt][ks] = wfrag[((mt_base + mt) * KS + ks) * 64 + lane];
            if (CBIN == 1) {
                const int q = 2 * ks + h;
                const int tap = q < Q ? q : 0;        // (the bias block: its B fragment is a constant, any in-bounds address will do)
                koff1[ks] = ((tap / 3 - 1) * K.Wp + tap % 3 - 1) * 8;
            }
        }
    }
    // element offset of k-step ks relative to (image of the sample, padded position, channel block in_blk + h * [CBIN > 1]):
    // K block q = 2 ks + h = tap * CBIN + cb.  For CBIN = 2: tap = ks, cb = h; for CBIN = 4: tap = ks >> 1, cb = 2 (ks & 1) + h --
    // the lane half only selects the channel block, which the caller folds into the base address, the rest is wave-uniform.
    // `d23` (CBIN = 4) = element distance from the image block that holds channels 0-7 to the one that holds channels 16-23.
    template <class KP>
    __device__ __forceinline__ int offset(const KP &K, int ks, int d23) const {
        if (CBIN == 1) return koff1[ks];
        // (CBIN = 8, the 64 -> 64 layers of ppg_policy_direct.h: four pairs of channel blocks d23 apart, tap = ks >> 2)
        const int tap = CBIN == 2 ? ks : CBIN == 4 ? ks >> 1 : ks >> 2;
        const int pr = CBIN == 2 ? 0 : CBIN == 4 ? (ks & 1) : (ks & 3);
        return pr * d23 + ((tap / 3 - 1) * K.Wp + tap % 3 - 1) * 8;
    }
    // The fragments were requested by load(): wait for them HERE, once, and hide their origin from the compiler.  Its wait-count
    // bookkeeping cannot follow a load across a loop's back edge: left alone it puts s_waitcnt vmcnt(0) in front of the first MFMA
    // of every position-tile loop -- which on this hardware also waits for every STORE the wavefront has issued since (conv3's, to
    // the scratch slot: one memory round trip per sub-group).
    __device__ __forceinline__ void landed() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4_t v = __builtin_bit_cast(u32x4_t, a[mt][ks]);
                __asm__ volatile("" : "+v"(v));
                a[mt][ks] = __builtin_bit_cast(bf16x8, v);
            }
    }
};

#ifdef PPG_EXPERIMENTS   // ablation builds (-DPPG_ABLATE=bits, timing only): 64 no ReLU / pack / stores, 128 the convolutions' MFMAs become one
// v_add per fragment, 256 no LDS fragment reads, 512 no position arithmetic
#ifndef PPG_ABLATE
#define PPG_ABLATE 0
#endif
#define PPG_MFMA(a, b, c) ((PPG_ABLATE & 128) ? ppg_fake_mfma(a, b, c) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0))
__device__ __forceinline__ f32x16 ppg_fake_mfma(bf16x8 a, bf16x8 b, f32x16 c) { c[0] += (float)a[0] + (float)b[0]; return c; }
#else
#define PPG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// One convolution layer over the `ns` samples of a sub-group, this wavefront's share of the 32-position tiles.
//   COUT_BLOCKS  channel blocks written: 2 (conv1), 4 (conv2), 8 (conv3)
//   TO_GLOBAL    conv3: the result goes to the scratch slot X[sample][position][64] instead of an LDS image
// The LDS image of a sample is six channel blocks of [padded position][8]; which of them a layer reads and writes rotates with
// the sub-group (phase_conv): `in_blk` = the block of input channels 0-7 (8-15 in the next one; conv3's 16-31 always in blocks 2, 3),
// `out_blk0` / `out_blk1` = where the lane halves h = 0 / 1 put output channels 16 h .. 16 h + 15 (two blocks each).
template <int CBIN, int MT, int COUT_BLOCKS, bool TO_GLOBAL, int BATCH = 0, class KP>
__device__ __forceinline__ void conv_layer(const KP &K, const ConvW<CBIN, MT> &W, __bf16 *img, int sample_stride, int in_blk,
                                           int out_blk0, int out_blk1, GLOBAL_AS __bf16 *xg_tile,
                                           int s_local0, int ns, int nt_first, int nt_step, int lane, int mt_base = 0) {
    constexpr int KS = ConvW<CBIN, MT>::KS;
    const int h = lane >> 5, col = lane & 31;
    const int n_pos = ns * K.P;
    const int n_tiles = (n_pos + 31) / 32;
    const __bf16 *in = img + (in_blk + (CBIN > 1 ? h : 0)) * K.Wp2 * 8;
    __bf16 *out = img + (h ? out_blk1 : out_blk0) * K.Wp2 * 8;
    const int d23 = (2 - in_blk) * K.Wp2 * 8;
    for (int nt = nt_first; nt < n_tiles; nt += nt_step) {
        const int n = 32 * nt + col;
        const bool valid = n < n_pos;
        const int nn = valid ? n : 0;
#ifdef PPG_EXPERIMENTS
        int s, p, pidx;
        if (PPG_ABLATE & 512) { s = 0; p = col; pidx = K.Wp + 1 + col; }   // ablation: no position arithmetic
        else {
            s = div_small(nn, K.magic_P); p = nn - __mul24(s, K.P);
            const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
            pidx = __mul24(y + 1, K.Wp) + (x + 1);
        }
#else
        const int s = div_small(nn, K.magic_P), p = nn - __mul24(s, K.P);
        const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
        const int pidx = __mul24(y + 1, K.Wp) + (x + 1);
#endif
        const __bf16 *base = in + __mul24(s, sample_stride) + pidx * 8;
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
        // the B fragment of k-step ks (the bias block's: {1, 1, 0 ...} in the lane half that holds it, zeros in the other)
        auto fragment = [&](int ks) -> bf16x8 {
            constexpr int KSB = ConvW<CBIN, MT>::KS_BIAS, HB = ConvW<CBIN, MT>::H_BIAS;
            bf16x8 v;
            if (ks == KSB && CBIN > 1) {
                v = zero8();
                if (h == HB) { v[0] = (__bf16)1.0f; v[1] = (__bf16)1.0f; }
            } else {
#ifdef PPG_EXPERIMENTS
                if (PPG_ABLATE & 256) v = zero8(); else   // ablation: no LDS fragment reads
#endif
                v = *(const bf16x8 *)(base + W.offset(K, ks, d23));
                if (ks == KSB && h == HB) { v = zero8(); v[0] = (__bf16)1.0f; v[1] = (__bf16)1.0f; }
            }
            return v;
        };
        if constexpr (BATCH == 0) {
            // all of the tile's B fragments are requested before the first MFMA: the LDS latency is paid once per tile, and while this
            // wavefront's MFMA chain runs the SIMD's other wavefront has the LDS to itself.  (A software-pipelined version -- fragment k
            // of tile t + 1 re-read right behind the MFMA of tile t that consumed it, MFMA / DS-read alternation pinned with
            // sched_group_barrier, no lgkmcnt stall left in the loop -- measured the same: 1.39 vs 1.40 ms under load, 4.84 vs 4.92 ms
            // with 64 workgroups alone on their CUs.  What a tile waits for is its epilogue and the barriers, not LDS.)
            bf16x8 b[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = fragment(ks);
            __builtin_amdgcn_sched_barrier(0);   // (keep the reads together: left alone, the scheduler re-pairs each with its MFMA)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = PPG_MFMA(W.a[mt][ks], b[ks], acc[mt]);
        } else {
            // conv3 (19 k-steps): the fragments come in batches of BATCH, batch n + 1 requested before the MFMAs of batch n -- two
            // batches of registers instead of 76, which is what lets the weights of all three layers stay in registers
            constexpr int NB = (KS + BATCH - 1) / BATCH;
            bf16x8 b[2][BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) b[0][i] = fragment(i);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (nb + 1 < NB) {
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) if ((nb + 1) * BATCH + i < KS) b[(nb + 1) & 1][i] = fragment((nb + 1) * BATCH + i);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < BATCH; ++i)
                    if (nb * BATCH + i < KS) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = PPG_MFMA(W.a[mt][nb * BATCH + i], b[nb & 1][i], acc[mt]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!valid || (TO_GLOBAL && (K.debug_skip & 32))) continue;
#ifdef PPG_EXPERIMENTS
        if (PPG_ABLATE & 64) {   // ablation: no ReLU / pack / stores (one value stored so that the MFMAs stay)
            if (acc[0][0] == 123.0f) *(float *)(img) = acc[0][0];
            continue;
        }
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {   // this lane's channels 32 mt + 16 h + 8 j .. + 7 = channel block 4 mt + 2 h + j
                const bf16x8 v = relu_pack8(acc[mt], 8 * j);
                if (TO_GLOBAL) {
                    // scratch slot X[sample][row tile][position][32]: a wavefront's 32 positions x 32 channels are 2 KB contiguous
                    *(GLOBAL_AS bf16x8 *)(xg_tile + (((size_t)(s_local0 + s) * 2 + (mt_base + mt)) * K.P + p) * 32 + 16 * h + 8 * j) = v;
                } else if (4 * (mt_base + mt) + 2 * h + j < COUT_BLOCKS) {   // (conv1: 16 real output channels: the h = 0 half only)
                    *(bf16x8 *)(out + __mul24(s, sample_stride) + (j * K.Wp2 + pidx) * 8) = v;
                }
            }
    }
}

// FC1 over the tile: M = 256 output features (A = weight fragments), N = up to 128 samples (B = the scratch slot X[sample][K1]),
// K in chunks of 32.  Wavefront w owns the feature row tiles 2w, 2w + 1 and ALL column tiles (2 x NT accumulators), so the same
// code serves tiles of 64, 96 and 128 samples.
//  - X: every wavefront needs every column, so the chunk (128 x 64 B) goes through LDS, brought by LDS-DMA loads
//    (global_load_lds_dwordx4: no registers, 1 KB per wave instruction) into one of THREE buffers, two chunks ahead of the one
//    being computed.  An LDS-DMA writes lane l's 16 bytes at base + 16 l, so the image cannot be padded; instead the 16-byte pieces
//    of row n sit at position piece ^ ((n >> 2) & 3) -- the swizzle is applied to the SOURCE address of the copy -- which makes the
//    16-lane groups of the fragment reads conflict-free.
//  - W: a wavefront's four fragments of a chunk are its own (nobody else reads them), so they go from L2 straight into registers,
//    one chunk ahead (two register sets, the chunk loop is unrolled by two; K1 / 32 = 2 R^2 is even).  Through LDS they cost a
//    third of the LDS bandwidth of the phase, which was its bound: with two workgroups per CU the fragment reads + DMA writes of a
//    chunk took longer than its 16 MFMAs per wavefront.
// Loads complete in order, so "chunk c is here" is a count: a step requests W(c + 1) [4 loads] and then X(c + 2) [2 loads]; at the
// top of step c everything but the two X(c + 1) loads must have landed: s_waitcnt vmcnt(2).
constexpr int FC1_BUF = TILE * 64;                   // bytes per X buffer

// (Cache hints were tried for the scratch slot -- nt stores in conv3, nt on these loads, nt on the observation reads, so that the
// streaming data would not push FC1's weights out of L2: every combination was 5-10 % SLOWER.)
__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)g, (void __attribute__((address_space(3))) *)l, 16, 0, 0);
}

__device__ __forceinline__ void fc1_issue_x(int K1, const __bf16 *xg_tile, unsigned char *stage, int c, int buf, int wave, int lane) {
    unsigned char *xb = stage + (size_t)buf * FC1_BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i0 = (wave * 2 + j) * 64, i = i0 + lane;
        const int row = i >> 2, q = (i & 3) ^ ((row >> 2) & 3);
        glds16(xg_tile + (size_t)row * K1 + 32 * c + 8 * q, xb + (size_t)i0 * 16);
    }
}

// The weight loads are written as inline assembly ON PURPOSE: hipcc's wait-count bookkeeping cannot follow loads whose results
// are consumed one loop iteration later in a rotating register set -- it puts s_waitcnt vmcnt(0) in front of the first MFMA of
// every chunk, i.e. it waits for the loads it has just issued.  Loads it does not know about it does not wait for; the explicit
// s_waitcnt vmcnt(2) at the top of each step is what guarantees their arrival, and fc1_landed() (an empty asm that "redefines" the
// registers behind that wait) keeps every use of them below it.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fc1_issue_w(const GLOBAL_AS bf16x8 *wlane, int c, u32x4 (&w)[2][2]) {
    const GLOBAL_AS bf16x8 *p0 = wlane + (size_t)c * 1024, *p1 = p0 + 512;   // k2 = 0 / 1: 8 KB apart; m = 1: + 1 KB
    __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[0][0]) : "v"(p0) : "memory");
    __asm__ volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(w[0][1]) : "v"(p0) : "memory");
    __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[1][0]) : "v"(p1) : "memory");
    __asm__ volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(w[1][1]) : "v"(p1) : "memory");
}
__device__ __forceinline__ void fc1_landed(u32x4 (&w)[2][2]) {
    __asm__ volatile("" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
}

template <int NT>
__device__ __forceinline__ void fc1_compute(const unsigned char *xb, const u32x4 (&w)[2][2], f32x16 (&acc)[2][NT], int lane) {
    const int h = lane >> 5, col = lane & 31;
    // all 2 x NT fragments of the chunk are requested before its first MFMA (the LDS latency is paid once per chunk)
    bf16x8 x[2][NT];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const int q = 2 * k2 + h;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 32 * t + col;
            x[k2][t] = *(const bf16x8 *)(xb + n * 64 + 16 * (q ^ ((n >> 2) & 3)));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[k2][m]), x[k2][t], acc[m][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);   // (or the next step's wait + barrier are scheduled in front of these MFMAs: nothing hidden)
}

#ifdef PPG_EXPERIMENTS
struct FcTimes {   // diagnostic builds: cycles a wave spends in the chunk loop's wait / barrier / the rest
    long long t[3] = {0, 0, 0}, prev = 0;
    __device__ __forceinline__ void start() { prev = (long long)clock64(); }
    __device__ __forceinline__ void add(int i) { const long long now = (long long)clock64(); t[i] += now - prev; prev = now; }
};
#else
struct FcTimes {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void add(int) {}
};
#endif

// one chunk: wait for it, meet, request W(c + 1) and X(c + 2) (compile-time switches: with a run-time condition around the loads
// the compiler's own wait-count bookkeeping gives up at the join and drains ALL loads before the MFMAs), compute.
template <int NT, bool ISSUE_W, bool ISSUE_X>
__device__ __forceinline__ void fc1_step(const GLOBAL_AS bf16x8 *wlane, int K1, const __bf16 *xg_tile, unsigned char *stage, int c, int buf,
                                         u32x4 (&w_cur)[2][2], u32x4 (&w_next)[2][2], f32x16 (&acc)[2][NT], int wave, int lane, FcTimes &T) {
    if (ISSUE_W) __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fc1_landed(w_cur);
    T.add(0);
    __builtin_amdgcn_s_barrier();   // everybody's X copies of chunk c are in LDS, and everybody is done with chunk c - 1's buffer
    T.add(1);
    if (ISSUE_W) fc1_issue_w(wlane, c + 1, w_next);
    if (ISSUE_X) fc1_issue_x(K1, xg_tile, stage, c + 2, buf >= 1 ? buf - 1 : 2, wave, lane);
    fc1_compute<NT>(stage + (size_t)buf * FC1_BUF, w_cur, acc, lane);
    T.add(2);
}

template <int NT>
__device__ __forceinline__ void fc1_staged(const bf16x8 *w1, int K1, const __bf16 *xg_tile, unsigned char *stage, f32x16 (&acc)[2][NT],
                                           int wave, int lane, FcTimes &T) {
    const int n_chunks = K1 / 32;    // = 2 R^2: even, >= 18
    const GLOBAL_AS bf16x8 *wlane = (const GLOBAL_AS bf16x8 *)w1 + (2 * wave) * 64 + lane;
    u32x4 w[2][2][2];
    fc1_issue_x(K1, xg_tile, stage, 0, 0, wave, lane);
    fc1_issue_w(wlane, 0, w[0]);
    fc1_issue_x(K1, xg_tile, stage, 1, 1, wave, lane);
    int buf = 0;
    T.start();
    for (int c = 0; c + 2 < n_chunks; c += 2) {   // (chunk c uses LDS buffer c % 3 and register set c % 2)
        fc1_step<NT, true, true>(wlane, K1, xg_tile, stage, c, buf, w[0], w[1], acc, wave, lane, T);
        buf = buf == 2 ? 0 : buf + 1;
        fc1_step<NT, true, true>(wlane, K1, xg_tile, stage, c + 1, buf, w[1], w[0], acc, wave, lane, T);
        buf = buf == 2 ? 0 : buf + 1;
    }
    fc1_step<NT, true, false>(wlane, K1, xg_tile, stage, n_chunks - 2, buf, w[0], w[1], acc, wave, lane, T);
    buf = buf == 2 ? 0 : buf + 1;
    fc1_step<NT, false, false>(wlane, K1, xg_tile, stage, n_chunks - 1, buf, w[1], w[0], acc, wave, lane, T);
    __syncthreads();
}

// M = 256 output features from K = 16 * KSTEPS inputs held in LDS as bsrc[sample][...] (B fragments: 16 bytes at
// bsrc + sample * bstride + 16 ks + 8 h); the weight fragments come straight from L2, fetched two k-steps ahead of their use.
template <int KSTEPS, int NT>
__device__ __forceinline__ void fc_256(const bf16x8 *wfrag, const __bf16 *bsrc, size_t bstride, f32x16 (&acc)[2][NT], int wave, int lane) {
    const int h = lane >> 5, col = lane & 31;
    const __bf16 *b0 = bsrc + (size_t)col * bstride + 8 * h;
    const bf16x8 *wa = wfrag + (size_t)(2 * wave) * 64 + lane;
    bf16x8 w[3][2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m) w[d][m] = wa[((size_t)d * 8 + m) * 64];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int cur = ks % 3, nxt = (ks + 2) % 3;
        if (ks + 2 < KSTEPS) {
#pragma unroll
            for (int m = 0; m < 2; ++m) w[nxt][m] = wa[((size_t)(ks + 2) * 8 + m) * 64];
        }
        bf16x8 x[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] = *(const bf16x8 *)(b0 + (size_t)(32 * t) * bstride + 16 * ks);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cur][m], x[t], acc[m][t], 0, 0, 0);
    }
}

template <int NT>
__device__ __forceinline__ void fc_init(const float *bias, f32x16 (&acc)[2][NT], int wave, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *(const GLOBAL_AS f32x4 *)(bias + 32 * (2 * wave + m) + 16 * h + 4 * g);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[m][t][4 * g + i] = b[i];
        }
}

// ReLU(acc) -> H[sample][feature] (bf16, LDS): 32 bytes per lane and tile
template <int NT>
__device__ __forceinline__ void fc_store(const f32x16 (&acc)[2][NT], __bf16 *H, int wave, int lane) {
    const int h = lane >> 5, col = lane & 31;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(bf16x8 *)(H + (size_t)(32 * t + col) * HSTRIDE + 32 * (2 * wave + m) + 16 * h + 8 * j) = relu_pack8(acc[m][t], 8 * j);
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

// The parameter block is read in place from the kernarg segment (scalar loads at the use sites): by value it would sit in ~100
// SGPRs for the whole kernel.  The three phases of a tile are separate NON-INLINED functions: each gets its own register
// allocation (inlined into one body, hipcc hoists address arithmetic of every phase to the top of the tile loop and spills
// hundreds of registers -- scratch reloads then sit between the LDS-DMA loads of FC1 and force vmcnt(0) waits).
typedef const __attribute__((address_space(4))) PolParams *KPtr;
// Arguments of a non-inlined device function arrive in VECTOR registers: the compiler has to treat them (and everything loaded
// through them) as possibly different per lane -- vector loads of the parameters, loop bounds in VGPRs, exec-masked "divergent"
// loops, conservative s_waitcnt vmcnt(0) at every join.  So the phases make their arguments uniform again with v_readfirstlane,
// the kernarg pointer included (__builtin_amdgcn_kernarg_segment_ptr() is a null pointer inside a callee).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uintptr_t uniform(uintptr_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uintptr_t)hi << 32) | lo;
}
template <class T>
__device__ __forceinline__ T *uniform(T *p) { return (T *)uniform((uintptr_t)p); }
__device__ __forceinline__ KPtr uniform(KPtr p) { return (KPtr)uniform((uintptr_t)p); }

// How phase A keeps the observation values it has requested for the next sub-group: in the rows' own element type (the
// conversion to bf16 -- via float, round to nearest even both times, as ppg_step's bfloat16 rows are made -- happens when they are
// written to LDS, one sub-group later); the 16-channel float64 variant converts to float at once (64 registers otherwise).
template <int OBS, int NCH>
struct ObsRaw {
    typedef typename std::conditional<OBS == 2, uint16_t, typename std::conditional<OBS == 1, float, double>::type>::type elem;
    // (bfloat16 rows travel as the zero-extended 16 bits in a register of their own: two 16-bit values in one register would be
    //  packed as soon as they are loaded, i.e. waited for)
    typedef typename std::conditional<OBS == 2, uint32_t, typename std::conditional<OBS == 0 && (NCH > 8), float, elem>::type>::type type;
    // (the empty asm pins the conversion -- and with it the wait for the load -- to the place where the value is staged: a pure
    //  function of a loaded value is otherwise scheduled right behind its load)
    static __device__ __forceinline__ __bf16 to_bf16(type v) {
        __asm__ volatile("" : "+v"(v));
        if constexpr (OBS == 2) return __builtin_bit_cast(__bf16, (uint16_t)v);
        else return (__bf16)(float)v;
    }
};


// ---- phase A: the tile's sample table, then the convolutions, ST samples at a time -> scratch slot X ----
// NCH = input channel slots a thread stages per position: 4 (channel-first: the four observation channels), 8 or 16 (channels-last:
// R <= 8 / R <= 15 channels); conv1 reads CB1 = 1 or 2 channel blocks of 8.
// OBS = element type of the observation rows: 0 float64, 1 float32, 2 bfloat16 (ppg_config.obs_dtype)
template <int OBS, int NCH>
__device__ __noinline__ void phase_conv(KPtr Kp, unsigned char *lds, int tile_, int n0_, int nt_samples_, __bf16 *xg_tile_) {
    constexpr int CB1 = NCH > 8 ? 2 : 1;
    const auto &K = *uniform(Kp);
    const int tile = uniform(tile_), n0 = uniform(n0_), nt_samples = uniform(nt_samples_);
    __bf16 *xg_tile = uniform(xg_tile_);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long *tab = (unsigned long long *)lds;
    __bf16 *img = (__bf16 *)(lds + TILE * 16);
    const int blk = K.Wp2 * 8, sample_stride = 6 * blk;    // elements per channel block and per sample: six blocks of [padded position][8]
    const int img_elems = K.ST * sample_stride;
#ifdef PPG_EXPERIMENTS
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = (long long)clock64();
#define PPG_TA(i) do { const long long now_ = (long long)clock64(); tacc[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define PPG_TA(i) do { } while (0)
#endif
    // conv3: wavefronts 0, 1 compute output channels 0-31, wavefronts 2, 3 channels 32-63, each for every other position tile --
    // so a wavefront needs ONE row tile's weight fragments (76 registers).  With those of conv1 / conv2 (20 - 40 / 40 registers)
    // they stay in registers for the whole sample tile (requested here, awaited behind the table and the zero fill).
    ConvW<4, 1> w3c;
    ConvW<CB1, 1> w1c;
    ConvW<2, 1> w2c;
    w3c.load(K, K.wc3, lane, wave >> 1);
    w1c.load(K, K.wc1, lane);
    w2c.load(K, K.wc2, lane);
    const int dbg = K.debug_skip;
    // sample -> (handle, env, row): walk forward from the tile's first env (a tile spans a handful of envs)
    if (tid < TILE) {
        unsigned long long src = 0, dst = 0;
        if (tid < nt_samples) {
            const uint32_t n = (uint32_t)(n0 + tid);
            int lo = (int)K.tile_env[tile];
            while (lo + 1 < K.n_envs && K.plan[PLAN_HDR + 1 + lo] <= n) ++lo;
            const int e = lo, row = (int)(n - K.plan[PLAN_HDR + e]);
            const int k = handle_of(K.env_base, K.n_handles, e);
            const int b = e - K.env_base[k];
            src = (unsigned long long)(uintptr_t)(K.obs[k] + ((size_t)b * K.cap + row) * (size_t)K.obs_elems * (OBS == 2 ? 2 : OBS == 1 ? 4 : 8));
            dst = (unsigned long long)(uintptr_t)(K.actions[k] + (size_t)b * K.S + K.slot0 + row);
        }
        tab[2 * tid] = src;
        tab[2 * tid + 1] = dst;
    }
    // halo rings (and the unused channels of the input image) are zero and stay zero: only interiors are ever written
    for (int i = tid; i < img_elems / 8; i += 256) ((bf16x8 *)img)[i] = zero8();
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    w3c.landed();
    w1c.landed();
    w2c.landed();
    __syncthreads();
    GLOBAL_AS __bf16 *xg = (GLOBAL_AS __bf16 *)xg_tile;
    // The observation values of a sub-group are REQUESTED two sub-groups ahead and WRITTEN to LDS one sub-group ahead (`stage`, into
    // the image block conv1 will read), so neither wait meets anything young: on this hardware a wavefront's loads and stores
    // complete in issue order and share one counter, so waiting for a load also waits for every store issued before... and, as the
    // compiler cannot count the stores of a loop, for every one issued AFTER it too.  Here the youngest stores in front of a wait are
    // conv3's of the previous sub-group, two layers old.  (The values stay in the rows' element type until they are staged: a
    // conversion at request time would wait for the loads it has just issued.)  A thread stages at most two positions (ST * P <= 512).
    typedef typename ObsRaw<OBS, NCH>::type raw_t;
    raw_t pre[2][NCH];
    auto request = [&](int s0) {
        const int left = nt_samples - s0, ns = left < K.ST ? left : K.ST;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
#pragma unroll
            for (int c = 0; c < NCH; ++c) pre[j][c] = (raw_t)0;
            if (idx < ns * K.P && !(dbg & 4)) {
                const int s = div_small(idx, K.magic_P), p = idx - __mul24(s, K.P);
                typedef typename ObsRaw<OBS, NCH>::elem elem_t;   // bfloat16 rows: what ppg_step rounded is what conv1 gets, bit for bit
                const GLOBAL_AS elem_t *src = (const GLOBAL_AS elem_t *)(uintptr_t)tab[2 * (s0 + s)] + p * K.p_stride;
#pragma unroll
                for (int c = 0; c < NCH; ++c) if (NCH == 4 || c < K.cin) pre[j][c] = (raw_t)src[c * K.c_stride];
            }
        }
    };
    auto stage = [&](int s0, int in_blk) {
        const int left = nt_samples - s0, ns = left < K.ST ? left : K.ST;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            if (idx < ns * K.P) {
                const int s = div_small(idx, K.magic_P), p = idx - __mul24(s, K.P);
                const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
#pragma unroll
                for (int cb = 0; cb < CB1; ++cb) {
                    bf16x8 v = zero8();
#pragma unroll
                    for (int c = 0; c < (NCH < 8 ? NCH : 8); ++c) v[c] = ObsRaw<OBS, NCH>::to_bf16(pre[j][8 * cb + c]);
                    *(bf16x8 *)(img + __mul24(s, sample_stride) + ((in_blk + cb) * K.Wp2 + __mul24(y + 1, K.Wp) + (x + 1)) * 8) = v;
                }
            }
        }
    };
    // Which image blocks a layer uses alternates with the sub-group k, so that the NEXT sub-group's input can be staged while conv3
    // still reads its own:        conv1: in -> c1, c1 + 1     conv2: c1, c1 + 1 -> in, in + 1, 2, 3     conv3: in, in + 1, 2, 3 -> scratch slot
    // with (in, c1) = (4, 0) for even k and (0, 4) for odd k; the input of sub-group k + 1 goes to block c1 (+ 1) once conv2 is done.
    if (!(dbg & 1)) {
        request(0);
        stage(0, 4);
        if (K.ST < nt_samples) request(K.ST);
    }
    __syncthreads();
    PPG_TA(0);
    int in_blk = 4, c1_blk = 0;
    for (int s0 = 0; s0 < nt_samples && !(dbg & 1); s0 += K.ST) {
        const int ns = (nt_samples - s0) < K.ST ? (nt_samples - s0) : K.ST;
        if (!(dbg & 16)) conv_layer<CB1, 1, 2, false>(K, w1c, img, sample_stride, in_blk, c1_blk, c1_blk, nullptr, 0, ns, wave, 4, lane);
        PPG_TA(3);
        __syncthreads();
        PPG_TA(4);
        if (!(dbg & 16)) conv_layer<2, 1, 4, false>(K, w2c, img, sample_stride, c1_blk, in_blk, 2, nullptr, 0, ns, wave, 4, lane);
        PPG_TA(5);
        __syncthreads();
        PPG_TA(6);
        if (s0 + K.ST < nt_samples) {
            stage(s0 + K.ST, c1_blk);
            PPG_TA(1);
            if (s0 + 2 * K.ST < nt_samples) request(s0 + 2 * K.ST);
        }
        PPG_TA(7);
        if (!(dbg & 8)) conv_layer<4, 1, 8, true, 5>(K, w3c, img, sample_stride, in_blk, 0, 0, xg, s0, ns, wave & 1, 2, lane, wave >> 1);
        PPG_TA(8);
        __syncthreads();
        PPG_TA(9);
        const int t_ = in_blk; in_blk = c1_blk; c1_blk = t_;
    }
    // rows of the scratch slot behind the last sample of a partial tile hold older data: finite bf16 values whose columns are
    // never stored.  (The slot is zero-filled at creation, so they are never NaN patterns.)
    // This workgroup's stores to its scratch slot are re-read by FC1: drain them, meet, drop stale L1 lines.
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PPG_EXPERIMENTS
    PPG_TA(0);   // (the drain + acquire at the end goes with the start-up work)
    if (K.timeline && lane == 0)
        for (int i = 0; i < 10; ++i) K.timeline[(size_t)tile * 64 + 8 + 12 * wave + i] = (unsigned long long)tacc[i];
#endif
}

// ---- phase B: FC1 from the scratch slot, ReLU -> H ----
template <int NT>
__device__ __noinline__ void phase_fc1(KPtr Kp, unsigned char *lds, const __bf16 *xg_tile_, int tile_ = 0) {
    const auto &K = *uniform(Kp);
    FcTimes T;
    const __bf16 *xg_tile = uniform(xg_tile_);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char *overlay = lds + TILE * 16;
    f32x16 acc[2][NT];
    fc_init<NT>(K.b1, acc, wave, lane);
    if (!(K.debug_skip & 2)) fc1_staged<NT>(K.w1, K.K1, xg_tile, overlay, acc, wave, lane, T);   // (staging buffers overlay the images)
#ifdef PPG_EXPERIMENTS
    if (K.timeline && lane == 0) {   // [8 + 12 wave + 10 / 11] = wait / barrier cycles of the chunk loop, [56 + wave] = the rest of it
        unsigned long long *t = K.timeline + (size_t)uniform(tile_) * 64;
        t[8 + 12 * wave + 10] = (unsigned long long)T.t[0];
        t[8 + 12 * wave + 11] = (unsigned long long)T.t[1];
        t[56 + wave] = (unsigned long long)T.t[2];
    }
#endif
    __syncthreads();
    fc_store<NT>(acc, (__bf16 *)overlay, wave, lane);      // (so does H)
    __syncthreads();
}

// ---- phases C, D: FC2 from H back into H; logits; actions ----
template <int NT>
__device__ __noinline__ void phase_head(KPtr Kp, unsigned char *lds, int n0_, int nt_samples_) {
    const auto &K = *uniform(Kp);
    const int n0 = uniform(n0_), nt_samples = uniform(nt_samples_);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long *tab = (const unsigned long long *)lds;
    __bf16 *H = (__bf16 *)(lds + TILE * 16);
    if (K.n_hidden >= 2) {   // (one hidden layer: the logits come straight from FC1's output)
        f32x16 acc[2][NT];
        fc_init<NT>(K.b2, acc, wave, lane);
        fc_256<16, NT>(K.w2, H, (size_t)HSTRIDE, acc, wave, lane);
        __syncthreads();
        fc_store<NT>(acc, H, wave, lane);
        __syncthreads();
    }
    if (wave >= NT) return;   // (one 32-sample column tile per wavefront)
    // logits = W3 (actions padded to 32 rows) x H^T, one 32-sample column tile per wavefront
    const int h = lane >> 5, col = lane & 31;
    f32x16 lg;
#pragma unroll
    for (int r = 0; r < 16; ++r) lg[r] = K.b3[16 * h + r];
    const __bf16 *hb = H + (size_t)(32 * wave + col) * HSTRIDE + 8 * h;
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks)
        lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(K.w3[ks * 64 + lane], *(const bf16x8 *)(hb + 16 * ks), lg, 0, 0, 0);
    // lane (h = 0) of a column holds the logits of actions 0-15, its partner lane + 32 those of actions 16-31
    float mine[16], other[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { mine[r] = lg[r]; other[r] = __shfl_xor(mine[r], 32, 64); }
    const int s_local = 32 * wave + col;
    if (h == 0 && s_local < nt_samples) {
        float logit[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) { logit[r] = mine[r]; logit[16 + r] = other[r]; }
        if (K.logits) {
            float *lo = K.logits + (size_t)(n0 + s_local) * K.n_actions;
#pragma unroll
            for (int a = 0; a < 32; ++a) if (a < K.n_actions) lo[a] = logit[a];
        }
        int8_t *dst = (int8_t *)(uintptr_t)tab[2 * s_local + 1];
        // Philox counter of this agent = (global env index, row slot): recovered from where its action goes.  (Never the address
        // itself: the same seed must give the same actions whatever tensor they are written to.)
        uint32_t c_env = 0, c_slot = 0;
        if (K.sample) {
            int k = 0;
#pragma unroll
            for (int q = 1; q < MAX_HANDLES; ++q)
                if (q < K.n_handles && (uintptr_t)dst >= (uintptr_t)K.actions[q] &&
                    (uintptr_t)dst < (uintptr_t)K.actions[q] + (size_t)(K.env_base[q + 1] - K.env_base[q]) * (size_t)K.S) k = q;
            const uint32_t off = (uint32_t)((uintptr_t)dst - (uintptr_t)K.actions[k]);
            const uint32_t b = off / (uint32_t)K.S;
            c_env = (uint32_t)K.env_base[k] + b;
            c_slot = off - b * (uint32_t)K.S;
        }
        uint32_t rnd[4] = {0, 0, 0, 0};
        int best = 0;
        float bestv = -INFINITY;
#pragma unroll
        for (int a = 0; a < 32; ++a) {
            if (a >= K.n_actions) continue;
            float v = logit[a];
            if (K.sample) {   // Gumbel-max: argmax(logit - log(-log u)) ~ softmax(logits)
                if ((a & 3) == 0) philox(c_env, c_slot, (uint32_t)(a >> 2), 0x504F4C31u, K.seed_lo, K.seed_hi, rnd);
                const float u = (float)(rnd[a & 3] >> 9) * (1.0f / 8388608.0f) + (1.0f / 16777216.0f);   // 23 bits: 2^-24 <= u < 1, exactly
                v -= __logf(-__logf(u));
            }
            if (v > bestv) { bestv = v; best = a; }
        }
        *dst = (int8_t)best;
    }
}

template <int OBS, int NCH>
__device__ __forceinline__ void policy_main(KPtr Kp, unsigned char *lds) {
    const int N = (int)Kp->plan[0], n_full = (int)Kp->plan[1], ts = (int)Kp->plan[2];
    const int n_tiles = n_full + (N - n_full * TILE + ts - 1) / ts;
    __bf16 *xg_tile = Kp->xg + (size_t)blockIdx.x * TILE * Kp->K1;
    for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
        const int size = tile < n_full ? TILE : ts;
        const int n0 = tile < n_full ? tile * TILE : n_full * TILE + (tile - n_full) * ts;
        const int nt_samples = (N - n0) < size ? (N - n0) : size;
        __syncthreads();   // the previous tile's readers of H / the table are done
#ifdef PPG_EXPERIMENTS
        unsigned long long *tl = Kp->timeline ? Kp->timeline + (size_t)tile * 64 : nullptr;
        const bool stamp = tl && threadIdx.x == 0;
        if (stamp) { tl[0] = blockIdx.x; tl[1] = (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); tl[2] = (unsigned)nt_samples; tl[3] = wall_clock64(); }
#define PPG_POL_STAMP(i) do { if (stamp) tl[i] = wall_clock64(); } while (0)
#else
#define PPG_POL_STAMP(i) do { } while (0)
#endif
        phase_conv<OBS, NCH>(Kp, lds, tile, n0, nt_samples, xg_tile);
        PPG_POL_STAMP(4);
        if (size <= 32) { phase_fc1<1>(Kp, lds, xg_tile, tile); PPG_POL_STAMP(5); phase_head<1>(Kp, lds, n0, nt_samples); }
        else if (size <= 64) { phase_fc1<2>(Kp, lds, xg_tile, tile); PPG_POL_STAMP(5); phase_head<2>(Kp, lds, n0, nt_samples); }
        else if (size <= 96) { phase_fc1<3>(Kp, lds, xg_tile, tile); PPG_POL_STAMP(5); phase_head<3>(Kp, lds, n0, nt_samples); }
        else { phase_fc1<4>(Kp, lds, xg_tile, tile); PPG_POL_STAMP(5); phase_head<4>(Kp, lds, n0, nt_samples); }
        PPG_POL_STAMP(6);
    }
}

#define PPG_POLICY_KERNEL(name, OBS, NCH)                                                        \
    extern "C" __global__ void __launch_bounds__(256, 2) name(const PolParams K) {               \
        extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                      \
        policy_main<OBS, NCH>((KPtr)__builtin_amdgcn_kernarg_segment_ptr(), lds);                \
    }
PPG_POLICY_KERNEL(ppg_policy_forward_f64, 0, 4)          // channel-first: R x R image, 4 channels
PPG_POLICY_KERNEL(ppg_policy_forward_f32, 1, 4)
PPG_POLICY_KERNEL(ppg_policy_forward_bf16, 2, 4)
PPG_POLICY_KERNEL(ppg_policy_forward_hwc8_f64, 0, 8)     // channels-last: 4 x R image, R <= 8 channels
PPG_POLICY_KERNEL(ppg_policy_forward_hwc8_f32, 1, 8)
PPG_POLICY_KERNEL(ppg_policy_forward_hwc8_bf16, 2, 8)
PPG_POLICY_KERNEL(ppg_policy_forward_hwc16_f64, 0, 16)   // channels-last, 9 <= R <= 15 channels (two channel blocks into conv1)
PPG_POLICY_KERNEL(ppg_policy_forward_hwc16_f32, 1, 16)
PPG_POLICY_KERNEL(ppg_policy_forward_hwc16_bf16, 2, 16)

}  // namespace ppgpol

#include "ppg_policy_direct.h"
#include "ppg_policy_pipe.h"

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------

struct ppg_policy {
    int32_t device, R, n_actions, layout, cin;
    int32_t obs_channels;  // C of the (C,R,R) rows this network reads
    int32_t direct;        // 1: no hidden head layer, ppg_policy_direct.h (2: with more than three convolutions); 0: the FC chain
    int32_t nch16;         // more than 8 input channels into conv1
    int32_t pipe;          // direct == 1 only: the two-role pipeline kernels (ppg_policy_pipe.h), 512 threads per workgroup
    uint64_t macs;         // real multiply-accumulates per observation
    ppgpol::PolParams base;
    void *dev_weights;     // one allocation: fragments + biases
    __bf16 *xg;            // scratch slots (FC chain only)
    float *lgs;            // logits rows of the tiles in flight (direct path only)
    bool xg_is_spread;     // (from ppg_alloc_spread)
    uint32_t *plan;        // [PLAN_HDR + plan_envs] header + prefix sums, then the first env of every tile
    int32_t plan_envs;
    size_t plan_words;     // words the buffer was allocated with (it depends on the species' row capacity too)
    uint32_t *pre_g;       // the fused launch's prefix sums in memory (PolParams2::pre_g), [2][FUSED_MAX_ENVS]; owned by the prey's policy
    int32_t grid;
    int32_t lds_bytes;
    hipStream_t side;      // ppg_policy_act with both species: the predators' launch runs beside the prey's (fork / join by events)
    hipEvent_t fork, join;
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

// MFMA row r of a 32-row tile computes this feature of the tile (see ORIENTATION above)
static int ppg_row_feature(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }

static uint16_t ppg_bf16_bits(float f) {   // round to nearest even (finite weights)
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// conv weights [cout][cin][3][3] -> fragments [mt][ks][lane][8]: lane (r = lane & 31, h = lane >> 5) holds, for K block
// q = 2 ks + h = tap * CBIN + cb, the eight input channels 8 cb .. 8 cb + 7 of output channel 32 mt + r at tap (ky, kx)
static void ppg_pack_conv(const float *w, const float *bias, int cout, int cin, int cbin, int mt_n, std::vector<uint16_t> &out) {
    const int Q = 9 * cbin, KS = (Q + 2) / 2;   // (block Q = the bias, ConvW)
    out.assign((size_t)mt_n * KS * 64 * 8, 0);
    for (int mt = 0; mt < mt_n; ++mt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, h = lane >> 5, q = 2 * ks + h, co = 32 * mt + ppg_row_feature(r);
                if (q > Q || co >= cout) continue;
                uint16_t *dst = &out[(((size_t)mt * KS + ks) * 64 + lane) * 8];
                if (q == Q) {   // bias = bf16 head + bf16 remainder
                    const uint16_t hi = ppg_bf16_bits(bias[co]);
                    uint32_t hb = (uint32_t)hi << 16;
                    float hf;
                    memcpy(&hf, &hb, 4);
                    dst[0] = hi;
                    dst[1] = ppg_bf16_bits(bias[co] - hf);
                    continue;
                }
                const int tap = q / cbin, cb = q % cbin;
                for (int j = 0; j < 8; ++j) {
                    const int ci = 8 * cb + j;
                    if (ci < cin) dst[j] = ppg_bf16_bits(w[((size_t)co * cin + ci) * 9 + tap]);
                }
            }
}

// conv1 of the pipeline kernels as fragments of v_mfma_f32_16x16x32_bf16 (ppg_policy_pipe.h: Conv1X), [ky][lane][8]: lane (r = lane &
// 15 = output channel, kq = lane >> 4) holds for kernel row ky -- kq < 3: the eight input channels 0-7 at tap (ky, kx = kq); kq = 3: the
// ninth input channel at the taps kx = 0, 1, 2 (elements 0-2; the rest zero).  cin <= 9.
static void ppg_pack_conv1x(const float *w, int cout, int cin, std::vector<uint16_t> &out) {
    out.assign((size_t)3 * 64 * 8, 0);
    for (int ky = 0; ky < 3; ++ky)
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 15, kq = lane >> 4;
            if (co >= cout) continue;
            uint16_t *dst = &out[((size_t)ky * 64 + lane) * 8];
            for (int j = 0; j < 8; ++j) {
                const int ci = kq < 3 ? j : 8, kx = kq < 3 ? kq : j;
                if (ci < cin && kx < 3) dst[j] = ppg_bf16_bits(w[((size_t)co * cin + ci) * 9 + ky * 3 + kx]);
            }
        }
}

// A Linear layer -> fragments [ks][mt][lane][8] of the 32x32x16 MFMA: lane holds output feature o = 32 mt + row feature(lane & 31) and
// inputs k = 16 ks + 8 h + j; `value(o, k)` returns the weight (0 for padding)
template <class Value>
static void ppg_pack_fc(int K, int mt_n, Value value, std::vector<uint16_t> &out) {
    const int KS = K / 16;
    out.assign((size_t)KS * mt_n * 64 * 8, 0);
    for (int ks = 0; ks < KS; ++ks)
        for (int mt = 0; mt < mt_n; ++mt)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, h = lane >> 5, o = 32 * mt + ppg_row_feature(r);
                for (int j = 0; j < 8; ++j)
                    out[(((size_t)ks * mt_n + mt) * 64 + lane) * 8 + j] = ppg_bf16_bits(value(o, 16 * ks + 8 * h + j));
            }
}

// the single Linear head of a network without hidden head layers -> fragments of v_mfma_f32_16x16x32_bf16, [action tile][k-step][lane][8]:
// lane holds action 16 tile + (lane & 15) and the features of area F's elements 32 ks + 8 (lane >> 4) + j; F is [position][flat_c
// channels] (flat_c = the last convolution's channels padded to blocks of 8); `nhwc`: which feature (channel c, position q) is
static void ppg_pack_head(const float *hw, int n_actions, int P, int cout_last, bool nhwc, std::vector<uint16_t> &out) {
    const int flat_c = (cout_last + 7) / 8 * 8, ksteps = (P * flat_c + 31) / 32, head_mt = (n_actions + 15) / 16, flat = P * cout_last;
    out.assign((size_t)head_mt * ksteps * 64 * 8, 0);
    for (int mt = 0; mt < head_mt; ++mt)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int a = 16 * mt + (lane & 15);
                if (a >= n_actions) continue;
                for (int j = 0; j < 8; ++j) {
                    const int e = 32 * ks + 8 * (lane >> 4) + j, q = e / flat_c, c = e % flat_c;
                    const int ft = (q < P && c < cout_last) ? (nhwc ? q * cout_last + c : c * P + q) : -1;
                    if (ft >= 0) out[(((size_t)mt * ksteps + ks) * 64 + lane) * 8 + j] = ppg_bf16_bits(hw[(size_t)a * flat + ft]);
                }
            }
}

extern "C" {

int ppg_policy_create_spec(int32_t device, const ppg_policy_spec *spec, ppg_policy **out);

int ppg_policy_create_layout(int32_t device, int32_t obs_range, int32_t n_actions, int32_t layout, const ppg_policy_weights *w,
                             ppg_policy **out) {
    if (!w || !out) return ppg_policy_fail(nullptr, PPG_EINVAL, "null argument");
    ppg_policy_spec sp;
    memset(&sp, 0, sizeof sp);
    sp.obs_channels = 4; sp.obs_range = obs_range; sp.n_actions = n_actions; sp.layout = layout; sp.flatten = PPG_POLICY_FLATTEN_NCHW;
    sp.n_conv = 3; sp.conv_out[0] = 16; sp.conv_out[1] = 32; sp.conv_out[2] = 64;
    sp.n_fc = 3; sp.fc_out[0] = 256; sp.fc_out[1] = 256; sp.fc_out[2] = n_actions;
    for (int l = 0; l < 3; ++l) { sp.conv_w[l] = w->conv_w[l]; sp.conv_b[l] = w->conv_b[l]; sp.fc_w[l] = w->fc_w[l]; sp.fc_b[l] = w->fc_b[l]; }
    return ppg_policy_create_spec(device, &sp, out);
}

int ppg_policy_create(int32_t device, int32_t obs_range, int32_t n_actions, const ppg_policy_weights *w, ppg_policy **out) {
    return ppg_policy_create_layout(device, obs_range, n_actions, PPG_POLICY_LAYOUT_CHW, w, out);
}

int ppg_policy_describe(const ppg_policy_spec *spec, int32_t *out, int32_t n);
int ppg_policy_destroy(ppg_policy *p);
// ppg_policy_describe restates the kernel choice and the layout without a device: a policy is only handed out if the two agree
static int ppg_policy_matches_description(ppg_policy *p, const ppg_policy_spec *spec) {
    int32_t d[4] = {-1, -1, -1, -1};
    const int family = p->pipe ? 3 : p->direct;
    if (ppg_policy_describe(spec, d, 4) != PPG_OK || d[0] != family || d[1] != p->base.ST || d[2] != p->lds_bytes || d[3] != (p->pipe ? 512 : 256))
        return ppg_policy_fail(nullptr, PPG_EHIP, "internal: ppg_policy_describe says family %d, %d samples per sub-group, %d bytes of LDS; built %d, %d, %d",
                               d[0], d[1], d[2], family, p->base.ST, p->lds_bytes);
    return PPG_OK;
}

#ifndef PPG_POLICY_PIPE
#define PPG_POLICY_PIPE 1   // (0: A/B builds without the two-role pipeline)
#endif
// diagnostic switch: environment variable PPG_POLICY_PIPE=0 at creation time sends a network the pipeline would take to the one-role
// direct-head kernels instead (tests compare the two bit for bit)
static bool ppg_pipe_enabled() {
    const char *e = getenv("PPG_POLICY_PIPE");
    return PPG_POLICY_PIPE && !(e && e[0] == '0' && e[1] == 0);
}
// The pipeline's SLOT TABLE (ppg_policy_direct.h: dconv_cells): the ST * P positions of a sub-group in an order in which slot n's cell of
// the padded image -- 16-byte cell number sample * stride_cells + (y + 1) * Wp + x + 1 -- is congruent to n modulo 16.  A 32-slot MFMA
// tile then is, bank-wise, a run of 32 consecutive cells: ds_read_b128's sixteen-lane groups and the stores' eight-lane groups touch every
// bank once.  Possible iff no residue class holds more than two cells per tile; which classes the cells fall into depends on the
// sample stride (ppg_pipe_layout tries the eight strides that keep the head's reads conflict-free).  Returns false if some class is
// too full (the table is then the plain order: slot n = position n).
static bool ppg_slot_table(int ST, int IH, int IW, int stride_elems, std::vector<uint16_t> &tab) {
    const int P = IH * IW, Wp = IW + 1, tiles = (ST * P + 31) / 32, per_class = 2 * tiles, stride_cells = stride_elems / 8;
    std::vector<std::vector<uint16_t>> cls(16);
    for (int s = 0; s < ST; ++s)
        for (int p = 0; p < P; ++p) {
            const int y = p / IW, x = p % IW;
            cls[(s * stride_cells + (y + 1) * Wp + x + 1) & 15].push_back((uint16_t)(s << 8 | p));
        }
    bool ok = true;
    for (int r = 0; r < 16; ++r) ok = ok && (int)cls[r].size() <= per_class;
    tab.assign((size_t)32 * tiles, 0xFFFFu);
    if (!ok) {
        for (int n = 0; n < ST * P; ++n) tab[n] = (uint16_t)((n / P) << 8 | (n % P));
        return false;
    }
    // slot 16 k + r <- the k-th cell of class r, the classes' cells in (sample, position) order: a tile's two halves then hold
    // neighbouring positions, and the first sub-group-of-a-share's short last sub-group leaves whole slots empty rather than scattered
    for (int r = 0; r < 16; ++r)
        for (size_t k = 0; k < cls[r].size(); ++k) tab[16 * k + r] = cls[r][k];
    return true;
}

// LDS layout of the two-role pipeline kernels (ppg_policy_pipe.h) for rows of C x R x R elements read as P positions; blk = elements of
// one channel block of a sample's padded image, f_elems = elements of an area F.  Device-free (ppg_policy_describe, CPU tests).
static bool ppg_pipe_layout_at(int C, int R, int P, int blk, int f_elems, int tail_slack, ppgpol::PolParams &K, int *lds_bytes, int stride);
static bool ppg_pipe_layout(int C, int R, int P, int blk, int f_elems, int tail_slack, ppgpol::PolParams &K, int *lds_bytes,
                            std::vector<uint16_t> *slot_tab = nullptr) {
    int stride = 10 * blk + 2 * f_elems;
    while (((stride / 2) % 64) % 8 != 4) stride += 8;   // consecutive samples 16 bytes x an odd number apart in the banks: the head's
                                                        // sixteen sample columns read conflict-free
    // ... and of the eight such strides modulo 256 bytes the first for which the slot table exists (ppg_slot_table: whether a tile's
    // cells can be spread over all bank groups depends on where consecutive samples' images start)
    std::vector<uint16_t> tab;
    const char *sw = getenv("PPG_POLICY_SLOTS");   // diagnostic switch: PPG_POLICY_SLOTS=0 at creation time keeps the plain order (A/B runs)
    for (int k = 0; k < 8 && !(sw && sw[0] == '0' && sw[1] == 0); ++k) {
        if (!ppg_pipe_layout_at(C, R, P, blk, f_elems, tail_slack, K, lds_bytes, stride + 16 * k)) break;
        if (ppg_slot_table(K.ST, P / R, R, K.sample_stride, tab)) {
            if (slot_tab) *slot_tab = tab;
            K.slot_tiles = (int)tab.size() / 32;
            return true;
        }
    }
    if (!ppg_pipe_layout_at(C, R, P, blk, f_elems, tail_slack, K, lds_bytes, stride)) return false;
    if (ppg_slot_table(K.ST, P / R, R, K.sample_stride, tab)) {   // (PPG_POLICY_SLOTS=0 and this stride happens to admit a table: not wanted)
        tab.assign(tab.size(), 0xFFFFu);
        for (int n = 0; n < K.ST * P; ++n) tab[n] = (uint16_t)((n / P) << 8 | (n % P));
    }
    if (slot_tab) *slot_tab = tab;
    K.slot_tiles = (int)tab.size() / 32;
    return true;
}
static bool ppg_pipe_layout_at(int C, int R, int P, int blk, int f_elems, int tail_slack, ppgpol::PolParams &K, int *lds_bytes, int stride) {
    const int fixed_p = 2 * 4096 + 64 + 2048 + 1024 + tail_slack;   // partial sums x 2, role B's counter, Gumbel noise x 2, dconv's dummy slots (64 lanes)
    const int row_bf16 = C * R * R * 2;   // bytes of a bfloat16 row; the area `raw` takes ST of them when they are whole 8-byte chunks
    const bool chunks = row_bf16 % 8 == 0;
    int st_cap = (160 * 1024 - fixed_p - 128 * 16) / (stride * 2 + (chunks ? row_bf16 : 0));   // (at least 128 samples of table)
    if (st_cap > 16) st_cap = 16;
    while (st_cap > 1 && st_cap * P > 256) --st_cap;   // a role-B thread stages one position
    if (st_cap < 1 || P > 256) return false;
    {
        int st = st_cap;
        double best = 0.0;
        for (int c = st_cap; c >= 1; --c) {   // the most samples per round of position tiles (four wavefronts a tile each)
            const int tiles = (c * P + 31) / 32, rounds = (tiles + 3) / 4;
            const double score = (double)c / rounds;
            if (score > best * 1.0001) { best = score; st = c; }
        }
        int raw_bytes = chunks ? (st * row_bf16 + 15) / 16 * 16 : 0;
        const int cpr = row_bf16 / 8;   // 8-byte chunks per row; the role-B threads cover floor(256 / cpr) samples per load
        K.pipe_slots = chunks && cpr <= 256 ? 256 / cpr : 0;
        K.pipe_ni = K.pipe_slots ? (st + K.pipe_slots - 1) / K.pipe_slots : 0;
        if (K.pipe_ni > ppgpol::PIPE_CHUNKS) K.pipe_ni = 0;   // (that many chunk registers per thread)
        if (!K.pipe_ni) raw_bytes = 0;
        K.pipe_magic = K.pipe_ni ? (uint32_t)((0x100000000ull + cpr - 1) / cpr) : 0u;
        int cap_tab = (160 * 1024 - fixed_p - raw_bytes - st * stride * 2) / 16;
        if (cap_tab > 2048) cap_tab = 2048;
        K.ST = st;
        K.range_tile = st * (cap_tab / st);
        K.off_x = 0; K.pipe_x1 = 4 * blk; K.off_y = 8 * blk; K.off_f = 10 * blk; K.pipe_f1 = K.off_f + f_elems;
        K.sample_stride = stride;
        K.pipe_red = K.range_tile * 16;
        K.pipe_raw = K.pipe_red + 2 * 4096 + 64 + 2048;
        K.pipe_img = K.pipe_raw + raw_bytes + 1024;
        *lds_bytes = K.pipe_img + st * stride * 2 + tail_slack;
    }
    return true;
}

// The shape rules of ppg_policy_create_spec, device-free: shared with ppg_policy_describe so that the description of a network exists
// exactly when the network can be created.  need_weights: also require the weight pointers.
static int ppg_policy_check_spec(const ppg_policy_spec &sp, bool need_weights) {
    const int layout = sp.layout, R = sp.obs_range, n_actions = sp.n_actions, C = sp.obs_channels;
    if (layout != PPG_POLICY_LAYOUT_CHW && layout != PPG_POLICY_LAYOUT_HWC) return ppg_policy_fail(nullptr, PPG_EINVAL, "unknown layout %d", layout);
    if (sp.flatten != PPG_POLICY_FLATTEN_NCHW && sp.flatten != PPG_POLICY_FLATTEN_NHWC) return ppg_policy_fail(nullptr, PPG_EINVAL, "unknown flatten order %d", sp.flatten);
    if (R < 1 || R > 15) return ppg_policy_fail(nullptr, PPG_EINVAL, "obs_range %d outside 1..15", R);
    if (C < 1 || C > 8) return ppg_policy_fail(nullptr, PPG_EINVAL, "obs_channels %d outside 1..8", C);
    if (n_actions < 1 || n_actions > 32) return ppg_policy_fail(nullptr, PPG_EINVAL, "n_actions %d outside 1..32", n_actions);
    if (sp.n_conv < 1 || sp.n_conv > PPG_POLICY_MAX_CONV) return ppg_policy_fail(nullptr, PPG_EINVAL, "n_conv %d outside 1..%d", sp.n_conv, PPG_POLICY_MAX_CONV);
    if (sp.n_fc < 1 || sp.n_fc > PPG_POLICY_MAX_FC) return ppg_policy_fail(nullptr, PPG_EINVAL, "n_fc %d outside 1..%d", sp.n_fc, PPG_POLICY_MAX_FC);
    for (int l = 0; l < sp.n_conv; ++l) {
        const int lim = l == 0 ? 16 : l == 1 ? 32 : 64;
        if (sp.conv_out[l] < 1 || sp.conv_out[l] > lim)
            return ppg_policy_fail(nullptr, PPG_EINVAL, "convolution %d has %d output channels: the kernels take up to 16 / 32 / 64 / 64 ...", l + 1, sp.conv_out[l]);
        if (need_weights && (!sp.conv_w[l] || !sp.conv_b[l])) return ppg_policy_fail(nullptr, PPG_EINVAL, "a convolution weight pointer is NULL");
    }
    for (int l = 0; l < sp.n_fc; ++l) {
        if (need_weights && (!sp.fc_w[l] || !sp.fc_b[l])) return ppg_policy_fail(nullptr, PPG_EINVAL, "a linear weight pointer is NULL");
        if (l + 1 < sp.n_fc && (sp.fc_out[l] < 1 || sp.fc_out[l] > 256))
            return ppg_policy_fail(nullptr, PPG_EINVAL, "hidden head layer %d has %d features: the kernels take up to 256", l + 1, sp.fc_out[l]);
    }
    if (sp.fc_out[sp.n_fc - 1] != n_actions) return ppg_policy_fail(nullptr, PPG_EINVAL, "the last linear layer has %d outputs, n_actions is %d", sp.fc_out[sp.n_fc - 1], n_actions);
    if (sp.n_fc > 1 && sp.n_conv != 3) return ppg_policy_fail(nullptr, PPG_EINVAL, "a head with hidden layers needs exactly three convolutions (found %d)", sp.n_conv);
    if (sp.n_fc > 1 && layout != PPG_POLICY_LAYOUT_HWC && C != 4)
        return ppg_policy_fail(nullptr, PPG_EINVAL, "channel-first networks with hidden head layers read 4-channel rows (found %d)", C);
    return PPG_OK;
}

int ppg_policy_create_spec(int32_t device, const ppg_policy_spec *spec, ppg_policy **out) {
    if (!spec || !out) return ppg_policy_fail(nullptr, PPG_EINVAL, "null argument");
    const ppg_policy_spec &sp = *spec;
    const int layout = sp.layout, R = sp.obs_range, n_actions = sp.n_actions, C = sp.obs_channels;
    {
        const int rc = ppg_policy_check_spec(sp, true);
        if (rc != PPG_OK) return rc;
    }
    // the image the convolutions run on: R x R with C channels, or (channels-last) C x R with R channels
    const int hwc = layout == PPG_POLICY_LAYOUT_HWC;
    const int IH = hwc ? C : R, IW = R, CIN = hwc ? R : C, CB1 = CIN > 8 ? 2 : 1;
    const int P = IH * IW;
    const bool direct = sp.n_fc == 1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return ppg_policy_fail(nullptr, PPG_ENODEV, "device %d not available", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ppg_policy_fail(nullptr, PPG_ENODEV, "ppg_policy needs a gfx950 device (MI355X)");
    ppg_policy *p = new (std::nothrow) ppg_policy();
    if (!p) return ppg_policy_fail(nullptr, PPG_ENOMEM, "out of host memory");
    memset(p, 0, sizeof *p);
    p->device = device; p->R = R; p->n_actions = n_actions; p->layout = layout; p->cin = CIN; p->obs_channels = C;
    p->direct = direct ? (sp.n_conv > 3 ? 2 : 1) : 0;
    p->nch16 = CIN > 8;
    const int cout_last = sp.conv_out[sp.n_conv - 1];
    const int flat = P * cout_last;   // features behind the flatten
    {   // real multiply-accumulates per observation
        uint64_t m = 0;
        int ci = CIN;
        for (int l = 0; l < sp.n_conv; ++l) { m += (uint64_t)P * sp.conv_out[l] * 9 * ci; ci = sp.conv_out[l]; }
        int fi = flat;
        for (int l = 0; l < sp.n_fc; ++l) { m += (uint64_t)fi * sp.fc_out[l]; fi = sp.fc_out[l]; }
        p->macs = m;
    }
    ppgpol::PolParams &K = p->base;
    // feature index of (channel c, position q) behind the flatten, or -1 for a padding channel
    const int flatten = sp.flatten;
    auto feature = [=](int c, int q) { return c >= cout_last ? -1 : flatten == PPG_POLICY_FLATTEN_NHWC ? q * cout_last + c : c * P + q; };
    std::vector<uint16_t> f[PPG_POLICY_MAX_CONV + 5];   // conv layers, then up to three linear layers, conv1 in the pipeline's form, its slot table
    std::vector<float> bias(256 + 256 + 32 + 16, 0.0f); // FC chain: b1, b2, b3; direct: head bias at [512]; the pipeline's conv1 bias at [544]
    int ci = CIN;
    for (int l = 0; l < sp.n_conv; ++l) {
        ppg_pack_conv(sp.conv_w[l], sp.conv_b[l], sp.conv_out[l], ci, l == 0 ? CB1 : l == 1 ? 2 : l == 2 ? 4 : 8, l < 2 ? 1 : 2, f[l]);
        K.cout_blocks[l] = (sp.conv_out[l] + 7) / 8;
        ci = sp.conv_out[l];
    }
    const int FC0 = PPG_POLICY_MAX_CONV;
    const int K1 = 64 * P;
    if (direct) {
        // head fragments of v_mfma_f32_16x16x32_bf16, [action tile][k-step][lane][8]: lane holds action 16 tile + (lane & 15) and the
        // features of area F's elements 32 ks + 8 (lane >> 4) + j; F is [position][flat_c channels]
        const int flat_c = 8 * K.cout_blocks[sp.n_conv - 1], ksteps = (P * flat_c + 31) / 32, head_mt = (n_actions + 15) / 16;
        K.flat_c = flat_c; K.kflat_steps = ksteps; K.head_mt = head_mt;
        ppg_pack_head(sp.fc_w[0], n_actions, P, cout_last, flatten == PPG_POLICY_FLATTEN_NHWC, f[FC0]);
        for (int a = 0; a < n_actions; ++a) bias[512 + a] = sp.fc_b[0][a];
        {   // K.whw: wavefront w owns k-steps [w per, (w + 1) per), its first 18 ride in registers (ppg_policy_direct.h)
            const int HFR = 18, per = (ksteps + 3) / 4;
            f[FC0 + 1].assign((size_t)4 * HFR * 64 * 8, 0);
            for (int w = 0; w < 4; ++w)
                for (int i = 0; i < HFR && i < per; ++i) {
                    const int ks = w * per + i;
                    if (ks >= ksteps) break;
                    memcpy(&f[FC0 + 1][((size_t)w * HFR + i) * 64 * 8], &f[FC0][(size_t)ks * 64 * 8], 64 * 8 * 2);
                }
        }
    } else {
        // FC1: the K order is the scratch slot's [row tile of conv3][position][32 channels]; hidden widths are zero-padded to 256
        const int h1 = sp.fc_out[0], h2 = sp.n_fc == 3 ? sp.fc_out[1] : 0;
        const float *w1 = sp.fc_w[0];
        ppg_pack_fc(K1, 8, [&](int o, int k) {
            const int c = 32 * (k / (32 * P)) + k % 32, q = (k % (32 * P)) / 32, ft = feature(c, q);
            return (o < h1 && ft >= 0) ? w1[(size_t)o * flat + ft] : 0.0f; }, f[FC0]);
        for (int i = 0; i < h1; ++i) bias[i] = sp.fc_b[0][i];
        if (sp.n_fc == 3) {
            const float *w2 = sp.fc_w[1];
            ppg_pack_fc(256, 8, [&](int o, int k) { return (o < h2 && k < h1) ? w2[(size_t)o * h1 + k] : 0.0f; }, f[FC0 + 1]);
            for (int i = 0; i < h2; ++i) bias[256 + i] = sp.fc_b[1][i];
        }
        const float *w3 = sp.fc_w[sp.n_fc - 1];
        const int hl = sp.n_fc == 3 ? h2 : h1;
        ppg_pack_fc(256, 1, [&](int o, int k) { return (o < n_actions && k < hl) ? w3[(size_t)o * hl + k] : 0.0f; }, f[FC0 + 2]);
        for (int i = 0; i < n_actions; ++i) bias[512 + i] = sp.fc_b[sp.n_fc - 1][i];
        K.n_hidden = sp.n_fc - 1;
    }
    if (CIN <= 9) {
        ppg_pack_conv1x(sp.conv_w[0], sp.conv_out[0], CIN, f[PPG_POLICY_MAX_CONV + 3]);
        for (int c = 0; c < sp.conv_out[0] && c < 16; ++c) bias[544 + c] = sp.conv_b[0][c];
    }
    // (the pipeline's LDS layout decides its slot table: computed here, before the upload, once more below for the parameter block)
    int pipe_lds_early = 0;
    const bool want_pipe = direct && ppg_pipe_enabled() && sp.n_conv == 3 && (n_actions + 15) / 16 == 1 && CIN <= 9;
    if (want_pipe) {
        ppgpol::PolParams Kt = K;
        Kt.Wp = IW + 1; Kt.Wp2 = (IH + 2) * (IW + 1) + 1;
        const int kfs = (P * 8 * ((cout_last + 7) / 8) + 31) / 32;
        if ((kfs + 3) / 4 <= 18)
            (void)ppg_pipe_layout(C, R, P, Kt.Wp2 * 8, kfs * 32 + 8, 18 * 32 * 2, Kt, &pipe_lds_early, &f[PPG_POLICY_MAX_CONV + 4]);
    }
    const int NF = PPG_POLICY_MAX_CONV + 5;
    size_t off[NF + 1], total = 0;
    for (int l = 0; l < NF; ++l) { off[l] = total; total += (f[l].size() * 2 + 255) / 256 * 256; }
    off[NF] = total; total += bias.size() * 4;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p->dev_weights, total) != hipSuccess) {
        delete p;
        return ppg_policy_fail(nullptr, PPG_EHIP, "hipMalloc of %zu bytes of weights failed", total);
    }
    std::vector<unsigned char> stage(total, 0);
    for (int l = 0; l < NF; ++l) if (!f[l].empty()) memcpy(stage.data() + off[l], f[l].data(), f[l].size() * 2);
    memcpy(stage.data() + off[NF], bias.data(), bias.size() * 4);
    if (hipMemcpy(p->dev_weights, stage.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->dev_weights);
        delete p;
        return ppg_policy_fail(nullptr, PPG_EHIP, "upload of the weights failed");
    }
#ifdef PPG_EXPERIMENTS   // ablation builds only (hipcc -DPPG_EXPERIMENTS): skip phases -- the results are then meaningless
    if (const char *dbg = getenv("PPG_POLICY_SKIP")) K.debug_skip = atoi(dbg);
#endif
    K.magic_P = (uint32_t)(((1u << ppgpol::DIV_SHIFT) + P - 1) / P);
    K.magic_R = (uint32_t)(((1u << ppgpol::DIV_SHIFT) + IW - 1) / IW);
    K.R = R; K.P = P; K.K1 = K1; K.n_actions = n_actions;
    K.IH = IH; K.IW = IW; K.cin = CIN; K.obs_elems = C * R * R;
    K.n_conv = sp.n_conv;
    // observation element [a][b][c] of the (C,R,R) row: channel-first = (channel a, position b * R + c); channels-last = (position
    // a * R + b, channel c)
    K.c_stride = hwc ? 1 : R * R; K.p_stride = hwc ? R : 1;
    const unsigned char *dw = (const unsigned char *)p->dev_weights;
    K.wc1 = (const ppgpol::bf16x8 *)(dw + off[0]); K.wc2 = (const ppgpol::bf16x8 *)(dw + off[1]);
    K.wc3 = (const ppgpol::bf16x8 *)(dw + off[2]);
    for (int l = 3; l < PPG_POLICY_MAX_CONV; ++l) K.wcd[l - 3] = (const ppgpol::bf16x8 *)(dw + off[l]);
    K.w1 = (const ppgpol::bf16x8 *)(dw + off[FC0]); K.w2 = (const ppgpol::bf16x8 *)(dw + off[FC0 + 1]); K.w3 = (const ppgpol::bf16x8 *)(dw + off[FC0 + 2]);
    K.wh = K.w1; K.whw = K.w2;
    const float *db = (const float *)(dw + off[NF]);
    K.bc1 = K.bc2 = K.bc3 = nullptr;   // (the convolutions' biases ride in their fragments)
    K.b1 = db; K.b2 = db + 256; K.b3 = db + 512; K.bh = db + 512;
    K.wc1x = (const ppgpol::bf16x8 *)(dw + off[PPG_POLICY_MAX_CONV + 3]); K.bc1x = db + 544;
    K.slot_tab = (const uint16_t *)(dw + off[PPG_POLICY_MAX_CONV + 4]);
    p->grid = 2 * prop.multiProcessorCount;
#ifdef PPG_EXPERIMENTS
    if (const char *g = getenv("PPG_POLICY_GRID")) p->grid = atoi(g) > 0 ? atoi(g) : p->grid;   // resident workgroups
#endif
    if (direct) {
        // LDS region of a sample: X (4 blocks) | Y (2 blocks) | F | [D0 | D1] -- padded pitch W + 1 (ppg_policy_direct.h)
        K.Wp = IW + 1; K.Wp2 = (IH + 2) * (IW + 1) + 1;
        const int blk = K.Wp2 * 8, f_elems = K.kflat_steps * 32 + 8;   // (+ 8: consecutive samples' fragments fall into different banks)
        K.off_x = 0; K.off_y = 4 * blk; K.off_f = 6 * blk; K.off_d0 = K.off_f + f_elems; K.off_d1 = K.off_d0 + 8 * blk;
        K.sample_stride = K.off_f + f_elems + (sp.n_conv > 3 ? 8 * blk : 0) + (sp.n_conv > 4 ? 8 * blk : 0);
        const int tail_slack = 18 * 32 * 2;   // bytes behind the last sample's region: the head's unconditional fragment reads end there
        // the two-role pipeline (ppg_policy_pipe.h): three convolutions, up to 16 actions, a wavefront's quarter of the head's k-steps in 18
        // fragments; region of a sample: X0 | X1 (4 blocks each) | Y (2 blocks) | F0 | F1
        if (ppg_pipe_enabled() && sp.n_conv == 3 && K.head_mt == 1 && (K.kflat_steps + 3) / 4 <= 18 && CIN <= 9) {   // (CIN: Conv1X)
            int pipe_lds = 0;
            if (ppg_pipe_layout(C, R, P, blk, f_elems, tail_slack, K, &pipe_lds)) {
                p->pipe = 1;
                p->grid = prop.multiProcessorCount;
                p->lds_bytes = pipe_lds;
                if (ppg_policy_matches_description(p, spec) != PPG_OK) { (void)ppg_policy_destroy(p); return PPG_EHIP; }
                *out = p;
                return PPG_OK;
            }
        }
        const int fixed = ppgpol::TILE * 16 + K.head_mt * 4096 + 4096 + tail_slack;   // table, partial logits, dconv's dummy slots
        const bool w1 = PPG_DIRECT_W1;   // one workgroup per CU with all of its LDS (ppg_policy_direct.h)
        int st_max = ((w1 ? 160 : 80) * 1024 - fixed) / (K.sample_stride * 2);
        if (st_max > 16) st_max = 16;                // (the head's 16 sample columns)
        while (st_max > 1 && st_max * P > (w1 ? 512 : 256)) --st_max;   // a thread stages at most two / one positions
        if (w1) p->grid = prop.multiProcessorCount;
        if (st_max < 1) {
            (void)hipFree(p->dev_weights);
            delete p;
            return ppg_policy_fail(nullptr, PPG_EINVAL, "a %d x %d image with %d convolutions needs %d bytes of LDS per sample: too large", IH, IW, sp.n_conv, K.sample_stride * 2);
        }
        // samples per sub-group: the most samples per round of position tiles (four wavefronts take a tile each)
        int st = st_max;
        double best = 0.0;
        for (int c = st_max; c >= 1; --c) {
            const int tiles = (c * P + 31) / 32, rounds = (tiles + 3) / 4;
            const double score = (double)c / rounds;
            if (score > best * 1.0001) { best = score; st = c; }
        }
        K.ST = st;
        K.range_tile = st * (ppgpol::TILE / st);   // whole sub-groups per tile
        p->lds_bytes = fixed + st * K.sample_stride * 2;
        const size_t lgs_bytes = (size_t)p->grid * ppgpol::TILE * 16 * K.head_mt * 4;
        if (hipMalloc((void **)&p->lgs, lgs_bytes) != hipSuccess || hipMemset(p->lgs, 0, lgs_bytes) != hipSuccess) {
            (void)hipFree(p->dev_weights);
            delete p;
            return ppg_policy_fail(nullptr, PPG_EHIP, "hipMalloc of %zu bytes of scratch failed", lgs_bytes);
        }
        K.lgs = p->lgs;
        if (ppg_policy_matches_description(p, spec) != PPG_OK) { (void)ppg_policy_destroy(p); return PPG_EHIP; }
        *out = p;
        return PPG_OK;
    }
    K.Wp = IW + 2; K.Wp2 = (IH + 2) * (IW + 2);
    // LDS: tile table + max(activation images of ST samples, H); ST as large as 2 workgroups per CU (80 KB each) allow
    const int per_sample = 6 * K.Wp2 * 8 * 2;                       // X (4 blocks) + Y (2 blocks), bytes
    const int h_bytes = ppgpol::TILE * ppgpol::HSTRIDE * 2;
    int st = (78 * 1024 - ppgpol::TILE * 16) / per_sample;
    if (st < 1) st = 1;
    if (st > 16) st = 16;
    while (st > 1 && st * P > 512) --st;   // the staging of a sub-group gives every thread at most two positions
    K.ST = st;
    const int img = st * per_sample, fc1_stage = 3 * ppgpol::FC1_BUF;
    int overlay = img > h_bytes ? img : h_bytes;
    if (fc1_stage > overlay) overlay = fc1_stage;
    p->lds_bytes = ppgpol::TILE * 16 + overlay;
    const size_t xg_bytes = (size_t)p->grid * ppgpol::TILE * K1 * 2;
    // the scratch slots: 512 concurrent sequential streams, written by conv3 and read back by FC1 -- on spread physical pages like the
    // env's observation tensors where the virtual-memory calls work (-DPPG_POLICY_XG_SPREAD=0: A/B builds)
#ifndef PPG_POLICY_XG_SPREAD
#define PPG_POLICY_XG_SPREAD 16
#endif
    const int xg_spread = PPG_POLICY_XG_SPREAD;
    void *xg_ptr = nullptr;
    p->xg_is_spread = xg_spread > 1 && ppg_alloc_spread(device, (uint64_t)xg_bytes, xg_spread, 0x5850u + (uint64_t)R, &xg_ptr) == PPG_OK;
    if (p->xg_is_spread) p->xg = (__bf16 *)xg_ptr;
    if ((!p->xg_is_spread && hipMalloc((void **)&p->xg, xg_bytes) != hipSuccess) || hipMemset(p->xg, 0, xg_bytes) != hipSuccess) {
        (void)hipFree(p->dev_weights);
        delete p;
        return ppg_policy_fail(nullptr, PPG_EHIP, "hipMalloc of %zu bytes of scratch failed", xg_bytes);
    }
    K.xg = p->xg;
    if (ppg_policy_matches_description(p, spec) != PPG_OK) { (void)ppg_policy_destroy(p); return PPG_EHIP; }
    *out = p;
    return PPG_OK;
}

int ppg_policy_pack(const ppg_policy_spec *spec, int32_t what, uint16_t *out, uint64_t capacity, uint64_t *n_words) {
    if (!spec || !n_words) return ppg_policy_fail(nullptr, PPG_EINVAL, "null argument");
    const ppg_policy_spec &sp = *spec;
    const int rc = ppg_policy_check_spec(sp, true);
    if (rc != PPG_OK) return rc;
    const int hwc = sp.layout == PPG_POLICY_LAYOUT_HWC;
    const int IH = hwc ? sp.obs_channels : sp.obs_range, IW = sp.obs_range, CIN = hwc ? sp.obs_range : sp.obs_channels, P = IH * IW;
    std::vector<uint16_t> f;
    if (what >= 0 && what < sp.n_conv) {
        int ci = CIN;
        for (int l = 0; l < what; ++l) ci = sp.conv_out[l];
        ppg_pack_conv(sp.conv_w[what], sp.conv_b[what], sp.conv_out[what], ci, what == 0 ? (CIN > 8 ? 2 : 1) : what == 1 ? 2 : what == 2 ? 4 : 8,
                      what < 2 ? 1 : 2, f);
    } else if (what == PPG_POLICY_PACK_CONV1X) {
        if (CIN > 9) return ppg_policy_fail(nullptr, PPG_EINVAL, "the pipeline's first convolution takes up to nine input channels (found %d)", CIN);
        ppg_pack_conv1x(sp.conv_w[0], sp.conv_out[0], CIN, f);
    } else if (what == PPG_POLICY_PACK_SLOTS) {
        ppgpol::PolParams K;
        memset(&K, 0, sizeof K);
        const int cb = (sp.conv_out[sp.n_conv - 1] + 7) / 8, kfs = (P * 8 * cb + 31) / 32;
        K.Wp = IW + 1; K.Wp2 = (IH + 2) * (IW + 1) + 1;
        int lds = 0;
        if (!(sp.n_fc == 1 && sp.n_conv == 3 && (sp.n_actions + 15) / 16 == 1 && (kfs + 3) / 4 <= 18 && CIN <= 9 &&
              ppg_pipe_layout(sp.obs_channels, sp.obs_range, P, K.Wp2 * 8, kfs * 32 + 8, 18 * 32 * 2, K, &lds, &f)))
            return ppg_policy_fail(nullptr, PPG_EINVAL, "PPG_POLICY_PACK_SLOTS: not a network of the two-role pipeline");
    } else if (what == PPG_POLICY_PACK_HEAD) {
        if (sp.n_fc != 1) return ppg_policy_fail(nullptr, PPG_EINVAL, "PPG_POLICY_PACK_HEAD: the network has hidden head layers");
        ppg_pack_head(sp.fc_w[0], sp.n_actions, P, sp.conv_out[sp.n_conv - 1], sp.flatten == PPG_POLICY_FLATTEN_NHWC, f);
    } else {
        return ppg_policy_fail(nullptr, PPG_EINVAL, "ppg_policy_pack: unknown layer %d", what);
    }
    *n_words = f.size();
    if (out && capacity >= f.size()) memcpy(out, f.data(), f.size() * 2);
    return PPG_OK;
}

int ppg_policy_describe(const ppg_policy_spec *spec, int32_t *out, int32_t n) {
    if (!spec || !out || n < 1) return PPG_EINVAL;
    const ppg_policy_spec &sp = *spec;
    const int R = sp.obs_range, C = sp.obs_channels;
    if (ppg_policy_check_spec(sp, false) != PPG_OK) return PPG_EINVAL;
    const int hwc = sp.layout == PPG_POLICY_LAYOUT_HWC;
    const int IH = hwc ? C : R, IW = R, P = IH * IW;
    int32_t v[12] = {0, 0, 0, 256, 0, 0, 0, 0, 0, 0, 0, 0};
    if (sp.n_fc > 1) {   // FC chain: tile table + max(images of ST samples, H, FC1 staging); two workgroups per CU
        const int per_sample = 6 * (IH + 2) * (IW + 2) * 8 * 2;
        int st = (78 * 1024 - ppgpol::TILE * 16) / per_sample;
        st = st < 1 ? 1 : st > 16 ? 16 : st;
        while (st > 1 && st * P > 512) --st;
        int overlay = st * per_sample;
        if (ppgpol::TILE * ppgpol::HSTRIDE * 2 > overlay) overlay = ppgpol::TILE * ppgpol::HSTRIDE * 2;
        if (3 * ppgpol::FC1_BUF > overlay) overlay = 3 * ppgpol::FC1_BUF;
        v[1] = st; v[2] = ppgpol::TILE * 16 + overlay;
    } else {
        ppgpol::PolParams K;
        memset(&K, 0, sizeof K);
        const int cout_last_blocks = (sp.conv_out[sp.n_conv - 1] + 7) / 8;
        K.flat_c = 8 * cout_last_blocks; K.kflat_steps = (P * K.flat_c + 31) / 32; K.head_mt = (sp.n_actions + 15) / 16;
        K.Wp = IW + 1; K.Wp2 = (IH + 2) * (IW + 1) + 1;
        const int blk = K.Wp2 * 8, f_elems = K.kflat_steps * 32 + 8, tail_slack = 18 * 32 * 2;
        int lds = 0;
        if (ppg_pipe_enabled() && sp.n_conv == 3 && K.head_mt == 1 && (K.kflat_steps + 3) / 4 <= 18 && (hwc ? R : C) <= 9 &&
            ppg_pipe_layout(C, R, P, blk, f_elems, tail_slack, K, &lds)) {
            v[0] = 3; v[1] = K.ST; v[2] = lds; v[3] = 512; v[4] = K.range_tile; v[5] = K.sample_stride * 2; v[6] = K.pipe_ni; v[7] = K.pipe_slots;
            v[8] = K.pipe_red; v[9] = K.pipe_raw; v[10] = K.pipe_img; v[11] = K.pipe_img - 1024 - K.pipe_raw;
        } else {
            const int stride = 6 * blk + f_elems + (sp.n_conv > 3 ? 8 * blk : 0) + (sp.n_conv > 4 ? 8 * blk : 0);
            const int fixed = ppgpol::TILE * 16 + K.head_mt * 4096 + 4096 + tail_slack;
            const bool w1 = PPG_DIRECT_W1;
            int st_max = ((w1 ? 160 : 80) * 1024 - fixed) / (stride * 2);
            if (st_max > 16) st_max = 16;
            while (st_max > 1 && st_max * P > (w1 ? 512 : 256)) --st_max;
            if (st_max < 1) return PPG_EINVAL;
            int st = st_max;
            double best = 0.0;
            for (int c = st_max; c >= 1; --c) {
                const int tiles = (c * P + 31) / 32, rounds = (tiles + 3) / 4;
                const double score = (double)c / rounds;
                if (score > best * 1.0001) { best = score; st = c; }
            }
            v[0] = sp.n_conv > 3 ? 2 : 1; v[1] = st; v[2] = fixed + st * stride * 2;
        }
    }
    for (int i = 0; i < n && i < 12; ++i) out[i] = v[i];
    return PPG_OK;
}

int ppg_policy_destroy(ppg_policy *p) {
    if (!p) return PPG_OK;
    if (p->dev_weights) (void)hipFree(p->dev_weights);
    if (p->xg) { if (p->xg_is_spread) (void)ppg_free_spread(p->xg); else (void)hipFree(p->xg); }
    if (p->plan) (void)hipFree(p->plan);
    if (p->pre_g) (void)hipFree(p->pre_g);
    if (p->lgs) (void)hipFree(p->lgs);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->fork) (void)hipEventDestroy(p->fork);
    if (p->join) (void)hipEventDestroy(p->join);
    delete p;
    return PPG_OK;
}

uint64_t ppg_policy_macs_per_observation(const ppg_policy *p) { return p ? p->macs : 0; }

const char *ppg_policy_last_error(const ppg_policy *p) { return p ? p->err : g_ppg_policy_error; }

// the plan buffer of a policy: [PLAN_HDR + envs] header + prefix sums, then the first env of every tile
static int ppg_policy_ensure_plan(ppg_policy *p, int total, int cap) {
    size_t max_tiles = ((size_t)total * cap + 31) / 32 + 1;   // (tiles of 32 samples are the smallest the plan picks)
    if (p->direct) {   // range mode: grid x tiles-per-workgroup slots, some of them empty
        const size_t share = ((size_t)total * cap + p->grid - 1) / p->grid + (size_t)p->base.ST;
        const size_t slots = (size_t)p->grid * ((share + p->base.range_tile - 1) / p->base.range_tile + 1);
        if (slots > max_tiles) max_tiles = slots;
    }
    // the tile list starts right behind the prefix sums of `total` envs: a buffer is reused only for the same env count and at most
    // the tile slots it was sized for (handles with the same env count but a larger row capacity need more slots: ADVICE r4)
    const size_t words = (size_t)(ppgpol::PLAN_HDR + total) + max_tiles;
    if (p->plan && p->plan_envs == total && p->plan_words >= words) return PPG_OK;
    if (p->plan) (void)hipFree(p->plan);
    p->plan = nullptr;
    p->plan_envs = 0; p->plan_words = 0;
    PPG_POL_TRY(p, hipMalloc((void **)&p->plan, words * 4));
    p->plan_envs = total;
    p->plan_words = words;
    return PPG_OK;
}

#ifdef PPG_PIPE_DEBUG
// diagnostic build: status words the pipeline kernels' bounded waits report into; checked (with a device synchronisation) after every
// launch.  PPG_PIPE_DEBUG_INJECT=1 (tests): pretend a report, to exercise the path that turns it into an error.
static uint32_t *g_pipe_status_host = nullptr;
static int ppg_pipe_debug_arm(ppg_policy *p) {
    if (!g_pipe_status_host) {
        PPG_POL_TRY(p, hipMalloc((void **)&g_pipe_status_host, 32));
        PPG_POL_TRY(p, hipMemcpyToSymbol(HIP_SYMBOL(ppgpol::g_pipe_status), &g_pipe_status_host, sizeof(uint32_t *)));
    }
    PPG_POL_TRY(p, hipMemset(g_pipe_status_host, 0, 32));
    return PPG_OK;
}
static int ppg_pipe_debug_check(ppg_policy *p) {
    uint32_t st[8];
    PPG_POL_TRY(p, hipDeviceSynchronize());
    PPG_POL_TRY(p, hipMemcpy(st, g_pipe_status_host, 32, hipMemcpyDeviceToHost));
    if (getenv("PPG_PIPE_DEBUG_INJECT")) { st[0] = 1; st[1] = 7; st[2] = 5; st[3] = 11; st[4] = 12; }
    if (st[0])
        return ppg_policy_fail(p, PPG_EHIP, "pipeline barrier: %u wavefront(s) gave up waiting; first: workgroup %u wavefront %u saw counter %u, wanted %u",
                               st[0], st[1], st[2], st[3], st[4]);
    return PPG_OK;
}
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the process-global kernel symbol, not of a policy: two live policies that
// share a kernel (say two pipe8 policies with different windows) need different amounts.  A running maximum per kernel is kept and raised
// right before the launch that needs more; the return code is checked.
static int ppg_policy_reserve_lds(ppg_policy *p, const void *fn, int bytes) {
    static const void *fns[256];   // (kernel, device): the attribute belongs to the code object loaded on a device
    static int devs[256], have[256];
    static int n_fns = 0;
    int i = 0;
    while (i < n_fns && !(fns[i] == fn && devs[i] == p->device)) ++i;
    if (i == n_fns) {
        if (n_fns == 256) return ppg_policy_fail(p, PPG_EHIP, "internal: more than 256 (policy kernel, device) pairs");
        fns[n_fns] = fn; devs[n_fns] = p->device; have[n_fns] = 0; ++n_fns;
    }
    if (have[i] >= bytes) return PPG_OK;
    PPG_POL_TRY(p, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have[i] = bytes;
    return PPG_OK;
}

// The per-call part of a species' parameter block: what the handles, the action tensors and the flags decide (no plan, no launch).
static int ppg_policy_fill(ppg_policy *p, int species, ppg_handle *const *handles, int32_t n, int8_t *const *actions, uint32_t flags,
                           uint64_t seed, float *logits, ppgpol::PolParams &K, ppgpol::PlanParams &L, int &total) {
    ppg_handle *h0 = handles[0];
    const int R = species ? h0->base.Rq : h0->base.Rp;
    if (R != p->R) return ppg_policy_fail(p, PPG_EINVAL, "the policy was created for %dx%d observations, the %s observe %dx%d", p->R, p->R,
                                          species ? "prey" : "predators", R, R);
    if (p->n_actions != 9 && !h0->gen2) return ppg_policy_fail(p, PPG_EINVAL, "the base env has 9 actions, the policy %d", p->n_actions);
    K = p->base;
    memset(&L, 0, sizeof L);
    K.species = species; K.obs_f32 = h0->base.obs_f32; K.sample = (flags & PPG_POLICY_SAMPLE) ? 1 : 0;
    K.seed_lo = (uint32_t)seed ^ (species ? 0x9E3779B9u : 0u); K.seed_hi = (uint32_t)(seed >> 32);
    K.seed_dev = (flags & PPG_POLICY_SEED_ON_DEVICE) ? (const uint64_t *)(uintptr_t)seed : nullptr;
    K.S = h0->base.S; K.cap = species ? h0->base.cap_prey : h0->base.cap_pred; K.slot0 = species ? h0->base.cap_pred : 0;
    K.n_handles = n; L.n_handles = n;
    L.word = species ? PPG_ENV_N_PREY_ROWS : PPG_ENV_N_PRED_ROWS;
    total = 0;
    for (int k = 0; k < n; ++k) {
        const ppg_handle *h = handles[k];
        if (!h) return ppg_policy_fail(p, PPG_EINVAL, "handle %d is NULL", k);
        const int channels = h->drive ? 4 + h->cfg.n_drive[species] : (h->gen2 && h->cfg2.walls && h->cfg2.include_visibility_channel) ? 5 : 4;
        if (channels != p->obs_channels)
            return ppg_policy_fail(p, PPG_EINVAL, "the policy reads %d-channel observations, handle %d writes %d channels", p->obs_channels, k, channels);
        if ((species ? h->base.Rq : h->base.Rp) != R || h->base.S != K.S || h->base.obs_f32 != h0->base.obs_f32 || h->device != p->device)
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
    K.logits = logits;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != p->device) PPG_POL_TRY(p, hipSetDevice(p->device));
    return PPG_OK;
}

// also_plan_for: the OTHER species' policy whose plan this call's plan launch computes as well (ppg_policy_plan2); skip_plan: this
// species' plan was computed by the other call
static int ppg_policy_run(ppg_policy *p, int species, ppg_handle *const *handles, int32_t n, int8_t *const *actions, uint32_t flags,
                          uint64_t seed, float *logits, void *stream, ppg_policy *also_plan_for = nullptr, bool skip_plan = false) {
    ppg_handle *h0 = handles[0];
    ppgpol::PolParams K;
    ppgpol::PlanParams L;
    int total = 0;
    {
        const int rc = ppg_policy_fill(p, species, handles, n, actions, flags, seed, logits, K, L, total);
        if (rc != PPG_OK) return rc;
    }
    if (K.obs_f32 == 3)
        return ppg_policy_fail(p, PPG_EINVAL, "observation rows in the cell layout (obs_dtype 3) need the one-launch form: both species' pipeline policies in one ppg_policy_act call");
    {
        const int rc = ppg_policy_ensure_plan(p, total, K.cap);
        if (rc != PPG_OK) return rc;
    }
    K.plan = L.plan = p->plan;
    K.tile_env = L.tile_env = p->plan + ppgpol::PLAN_HDR + total;
    L.slots = p->grid;
    L.range_tile = p->direct ? K.range_tile : 0;
    L.range_st = K.ST;
#ifdef PPG_EXPERIMENTS
    if (const char *f = getenv(species ? "PPG_POLICY_TILE_PREY" : "PPG_POLICY_TILE_PRED")) {
        const int v = atoi(f);
        if (v == 32 || v == 64 || v == 96 || v == 128) L.force_ts = v;
    }
#endif
#ifdef PPG_EXPERIMENTS   // PPG_POLICY_TIMELINE=<file prefix>: per-tile phase stamps of launch number PPG_POLICY_TIMELINE_RUN (default 300) of each species
    static int tl_runs[2] = {0, 0};
    static unsigned long long *tl_buf[2] = {nullptr, nullptr};
    const char *tl_path = getenv("PPG_POLICY_TIMELINE");
    const int tl_at = getenv("PPG_POLICY_TIMELINE_RUN") ? atoi(getenv("PPG_POLICY_TIMELINE_RUN")) : 300;
    const size_t tl_tiles = ((size_t)total * K.cap + 31) / 32 + 1;
    if (tl_path && !tl_buf[species]) {
        PPG_POL_TRY(p, hipMalloc((void **)&tl_buf[species], tl_tiles * 512));
        PPG_POL_TRY(p, hipMemset(tl_buf[species], 0, tl_tiles * 512));
        PPG_POL_TRY(p, hipDeviceSynchronize());
    }
    const bool tl_now = tl_path && ++tl_runs[species] == tl_at;
    K.timeline = tl_now ? tl_buf[species] : nullptr;
#endif
    if (also_plan_for) {   // the other species' plan in the same launch: same envs, its row-count word, its buffers, its grid
        ppg_policy *o = also_plan_for;
        const int rc = ppg_policy_ensure_plan(o, total, species ? h0->base.cap_pred : h0->base.cap_prey);
        if (rc != PPG_OK) { memcpy(p->err, o->err, sizeof p->err); return rc; }
        ppgpol::PlanParams L2 = L;
        L2.word = species ? PPG_ENV_N_PRED_ROWS : PPG_ENV_N_PREY_ROWS;
        L2.plan = o->plan;
        L2.tile_env = o->plan + ppgpol::PLAN_HDR + total;
        L2.slots = o->grid;
        L2.force_ts = 0;
        L2.range_tile = o->direct ? o->base.range_tile : 0;
        L2.range_st = o->base.ST;
        hipLaunchKernelGGL(ppgpol::ppg_policy_plan2, dim3(2), dim3(1024), 0, (hipStream_t)stream, L, L2);
    } else if (!skip_plan) {
        hipLaunchKernelGGL(p->direct ? ppgpol::ppg_policy_plan : ppgpol::ppg_policy_plan_small, dim3(1), dim3(1024), 0, (hipStream_t)stream, L);
    }
    typedef void (*fwd_fn)(const ppgpol::PolParams);
    const fwd_fn fwd[3][3] = {{ppgpol::ppg_policy_forward_f64, ppgpol::ppg_policy_forward_f32, ppgpol::ppg_policy_forward_bf16},
                              {ppgpol::ppg_policy_forward_hwc8_f64, ppgpol::ppg_policy_forward_hwc8_f32, ppgpol::ppg_policy_forward_hwc8_bf16},
                              {ppgpol::ppg_policy_forward_hwc16_f64, ppgpol::ppg_policy_forward_hwc16_f32, ppgpol::ppg_policy_forward_hwc16_bf16}};
    const int dt = K.obs_f32 == 2 ? 2 : K.obs_f32 ? 1 : 0;
#ifdef PPG_DIRECT_PROFILE   // diagnostic build: per-wavefront phase cycles of launch number PPG_DIRECT_PROFILE_RUN (default 300) -> $PPG_DIRECT_PROFILE_FILE.{pred,prey}
    static int dp_runs[2] = {0, 0};
    static unsigned long long *dp_buf[2] = {nullptr, nullptr};
    const char *dp_path = getenv("PPG_DIRECT_PROFILE_FILE");
    const int dp_at = getenv("PPG_DIRECT_PROFILE_RUN") ? atoi(getenv("PPG_DIRECT_PROFILE_RUN")) : 300;
    const size_t dp_bytes = (size_t)p->grid * 8 * 16 * 8;   // (the pipeline kernels have eight wavefronts, the others use the first half)
    if (p->direct && dp_path && !dp_buf[species]) {
        PPG_POL_TRY(p, hipMalloc((void **)&dp_buf[species], dp_bytes));
        PPG_POL_TRY(p, hipMemset(dp_buf[species], 0, dp_bytes));
        PPG_POL_TRY(p, hipDeviceSynchronize());
    }
    const bool dp_now = p->direct && dp_path && ++dp_runs[species] == dp_at;
    K.xg = dp_now ? (__bf16 *)dp_buf[species] : nullptr;
#endif
    if (p->direct) {
        const fwd_fn dir[4][3] = {{ppgpol::ppg_policy_direct8_f64, ppgpol::ppg_policy_direct8_f32, ppgpol::ppg_policy_direct8_bf16},
                                  {ppgpol::ppg_policy_direct16_f64, ppgpol::ppg_policy_direct16_f32, ppgpol::ppg_policy_direct16_bf16},
                                  {ppgpol::ppg_policy_deep8_f64, ppgpol::ppg_policy_deep8_f32, ppgpol::ppg_policy_deep8_bf16},
                                  {ppgpol::ppg_policy_deep16_f64, ppgpol::ppg_policy_deep16_f32, ppgpol::ppg_policy_deep16_bf16}};
        const fwd_fn pipe[2][3] = {{ppgpol::ppg_policy_pipe8_f64, ppgpol::ppg_policy_pipe8_f32, ppgpol::ppg_policy_pipe8_bf16},
                                   {ppgpol::ppg_policy_pipe16_f64, ppgpol::ppg_policy_pipe16_f32, ppgpol::ppg_policy_pipe16_bf16}};
        const fwd_fn fn = p->pipe ? pipe[p->nch16 ? 1 : 0][dt] : dir[(p->direct == 2 ? 2 : 0) + (p->nch16 ? 1 : 0)][dt];
        int rc = ppg_policy_reserve_lds(p, (const void *)fn, p->lds_bytes);
        if (rc != PPG_OK) return rc;
#ifdef PPG_PIPE_DEBUG
        rc = ppg_pipe_debug_arm(p);
        if (rc != PPG_OK) return rc;
#endif
        hipLaunchKernelGGL(fn, dim3((unsigned)p->grid), dim3(p->pipe ? 512 : 256), (size_t)p->lds_bytes, (hipStream_t)stream, K);
#ifdef PPG_PIPE_DEBUG
        PPG_POL_TRY(p, hipGetLastError());
        rc = ppg_pipe_debug_check(p);
        if (rc != PPG_OK) return rc;
#endif
    } else {
        const int variant = p->layout == PPG_POLICY_LAYOUT_HWC ? (p->cin > 8 ? 2 : 1) : 0;
        const int rc = ppg_policy_reserve_lds(p, (const void *)fwd[variant][dt], p->lds_bytes);
        if (rc != PPG_OK) return rc;
        hipLaunchKernelGGL(fwd[variant][dt], dim3((unsigned)p->grid), dim3(256), (size_t)p->lds_bytes, (hipStream_t)stream, K);
    }
    PPG_POL_TRY(p, hipGetLastError());
#ifdef PPG_DIRECT_PROFILE
    if (dp_now) {
        PPG_POL_TRY(p, hipDeviceSynchronize());
        std::vector<unsigned long long> host(dp_bytes / 8);
        PPG_POL_TRY(p, hipMemcpy(host.data(), dp_buf[species], dp_bytes, hipMemcpyDeviceToHost));
        char name[512];
        snprintf(name, sizeof name, "%s.%s", dp_path, species ? "prey" : "pred");
        if (FILE *f = fopen(name, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    }
#endif
#ifdef PPG_EXPERIMENTS
    if (tl_now && species == 0) {   // (the predators' launch is the second of a step: both species' stamps are complete after a device sync)
        PPG_POL_TRY(p, hipDeviceSynchronize());
        for (int sp = 0; sp < 2; ++sp) {
            if (!tl_buf[sp]) continue;
            std::vector<unsigned long long> host(tl_tiles * 64);
            PPG_POL_TRY(p, hipMemcpy(host.data(), tl_buf[sp], tl_tiles * 512, hipMemcpyDeviceToHost));
            char name[512];
            snprintf(name, sizeof name, "%s.%s", tl_path, sp ? "prey" : "pred");
            if (FILE *f = fopen(name, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
        }
    }
#endif
    return PPG_OK;
}

// diagnostic switch: environment variable PPG_POLICY_FUSED=0 sends a call with both species' pipeline policies through the two
// forward launches + plan launch of rounds 2-4 instead of the one fused launch (tests compare the two bit for bit)
static bool ppg_fused_enabled() {
    const char *e = getenv("PPG_POLICY_FUSED");
    return !(e && e[0] == '0' && e[1] == 0);
}

static int ppg_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && atoi(e) > 0) ? atoi(e) : dflt;
}

// Both species' pipeline policies in ONE launch and no plan launch (ppg_policy_pipe.h: fused_main).  Returns 1 if this call cannot
// take that path (the caller then runs the separate launches), PPG_OK after the launch, or an error.
static int ppg_policy_run_fused(ppg_policy *pred, ppg_policy *prey, ppg_handle *const *handles, int32_t n, int8_t *const *actions,
                                uint32_t flags, uint64_t seed, float *logits_pred, float *logits_prey, void *stream) {
    ppgpol::PolParams2 K2;
    ppgpol::PlanParams L;
    int total_q = 0, total_p = 0;
    int rc = ppg_policy_fill(prey, 1, handles, n, actions, flags, seed, logits_prey, K2.q, L, total_q);
    if (rc != PPG_OK) { memcpy(g_ppg_policy_error, prey->err, sizeof g_ppg_policy_error); return rc; }
    rc = ppg_policy_fill(pred, 0, handles, n, actions, flags, seed, logits_pred, K2.p, L, total_p);
    if (rc != PPG_OK) { memcpy(g_ppg_policy_error, pred->err, sizeof g_ppg_policy_error); return rc; }
    const int lds = prey->lds_bytes > pred->lds_bytes ? prey->lds_bytes : pred->lds_bytes;
    const int img = K2.q.pipe_img > K2.p.pipe_img ? K2.q.pipe_img : K2.p.pipe_img;
    K2.scratch_off = (img + 15) / 16 * 16;
    if (total_q > ppgpol::FUSED_MAX_ENVS || K2.scratch_off + ppgpol::FUSED_PART_WORDS * 4 + 8 * total_q > lds) return 1;
    K2.q.plan = K2.p.plan = nullptr; K2.q.tile_env = K2.p.tile_env = nullptr;
    if (!prey->pre_g) PPG_POL_TRY(prey, hipMalloc((void **)&prey->pre_g, (size_t)2 * ppgpol::FUSED_MAX_ENVS * 4));
    K2.pre_g = prey->pre_g;
    // cycles per pipeline iteration of either network (measured on the reference's shapes, profiles/r04-r05; PPG_POLICY_ITER_Q / _P:
    // experiments): only their RATIO matters -- it decides how many workgroups serve which species
    K2.iter_q = ppg_env_int("PPG_POLICY_ITER_Q", 7500);
    K2.iter_p = ppg_env_int("PPG_POLICY_ITER_P", 8500);
    typedef void (*fused_fn)(const ppgpol::PolParams2);
    const fused_fn fn[2][2][3] = {
        {{ppgpol::ppg_policy_pipe2_8_8_f64, ppgpol::ppg_policy_pipe2_8_8_f32, ppgpol::ppg_policy_pipe2_8_8_bf16},
         {ppgpol::ppg_policy_pipe2_8_16_f64, ppgpol::ppg_policy_pipe2_8_16_f32, ppgpol::ppg_policy_pipe2_8_16_bf16}},
        {{ppgpol::ppg_policy_pipe2_16_8_f64, ppgpol::ppg_policy_pipe2_16_8_f32, ppgpol::ppg_policy_pipe2_16_8_bf16},
         {ppgpol::ppg_policy_pipe2_16_16_f64, ppgpol::ppg_policy_pipe2_16_16_f32, ppgpol::ppg_policy_pipe2_16_16_bf16}}};
    const int dt = K2.q.obs_f32 == 2 ? 2 : K2.q.obs_f32 ? 1 : 0;
    const fused_fn f = fn[prey->nch16 ? 1 : 0][pred->nch16 ? 1 : 0][dt];
#ifdef PPG_DIRECT_PROFILE   // diagnostic build: per-wavefront phase cycles of launch number PPG_DIRECT_PROFILE_RUN (default 300) -> $PPG_DIRECT_PROFILE_FILE.fused
    static int dp_runs = 0;
    static unsigned long long *dp_buf = nullptr;
    const char *dp_path = getenv("PPG_DIRECT_PROFILE_FILE");
    const int dp_at = getenv("PPG_DIRECT_PROFILE_RUN") ? atoi(getenv("PPG_DIRECT_PROFILE_RUN")) : 300;
    const size_t dp_bytes = (size_t)prey->grid * 8 * 16 * 8;   // (eight wavefronts per workgroup, sixteen words each)
    if (dp_path && !dp_buf) {
        PPG_POL_TRY(prey, hipMalloc((void **)&dp_buf, dp_bytes));
        PPG_POL_TRY(prey, hipMemset(dp_buf, 0, dp_bytes));
        PPG_POL_TRY(prey, hipDeviceSynchronize());
    }
    const bool dp_now = dp_path && ++dp_runs == dp_at;
    K2.q.xg = K2.p.xg = dp_now ? (__bf16 *)dp_buf : nullptr;
#endif
    rc = ppg_policy_reserve_lds(prey, (const void *)f, lds);
    if (rc != PPG_OK) { memcpy(g_ppg_policy_error, prey->err, sizeof g_ppg_policy_error); return rc; }
#ifdef PPG_PIPE_DEBUG
    rc = ppg_pipe_debug_arm(prey);
    if (rc != PPG_OK) { memcpy(g_ppg_policy_error, prey->err, sizeof g_ppg_policy_error); return rc; }
#endif
    hipLaunchKernelGGL(f, dim3((unsigned)prey->grid), dim3(512), (size_t)lds, (hipStream_t)stream, K2);
    PPG_POL_TRY(prey, hipGetLastError());
#ifdef PPG_PIPE_DEBUG
    rc = ppg_pipe_debug_check(prey);
    if (rc != PPG_OK) { memcpy(g_ppg_policy_error, prey->err, sizeof g_ppg_policy_error); return rc; }
#endif
#ifdef PPG_DIRECT_PROFILE
    if (dp_now) {
        PPG_POL_TRY(prey, hipDeviceSynchronize());
        std::vector<unsigned long long> host(dp_bytes / 8);
        PPG_POL_TRY(prey, hipMemcpy(host.data(), dp_buf, dp_bytes, hipMemcpyDeviceToHost));
        char name[512];
        snprintf(name, sizeof name, "%s.fused", dp_path);
        if (FILE *fo = fopen(name, "wb")) { fwrite(host.data(), 8, host.size(), fo); fclose(fo); }
    }
#endif
    return PPG_OK;
}

int ppg_policy_act(ppg_policy *pred, ppg_policy *prey, ppg_handle *const *handles, int32_t n, int8_t *const *actions,
                   uint32_t flags, uint64_t seed, float *logits_pred, float *logits_prey, void *stream) {
    ppg_policy *any = pred ? pred : prey;
    if (!any) return PPG_EINVAL;
    if (!handles || !actions || n < 1 || n > ppgpol::MAX_HANDLES || !handles[0]) return ppg_policy_fail(any, PPG_EINVAL, "bad handle list");
    if (flags & ~(PPG_POLICY_SAMPLE | PPG_POLICY_SEED_ON_DEVICE)) return ppg_policy_fail(any, PPG_EINVAL, "unknown policy flags 0x%x", flags);
    if (pred && prey && pred->pipe && prey->pipe && pred->device == prey->device && pred->grid == prey->grid && ppg_fused_enabled()) {
        const int rc = ppg_policy_run_fused(pred, prey, handles, n, actions, flags, seed, logits_pred, logits_prey, stream);
        if (rc != 1) return rc;
    }
    if (flags & PPG_POLICY_SEED_ON_DEVICE)
        return ppg_policy_fail(any, PPG_EINVAL, "PPG_POLICY_SEED_ON_DEVICE is for the one-launch form: both species' pipeline policies on one device");
    // (a failure is also reported through ppg_policy_last_error(NULL), whichever of the two policies it came from)
    // The predators are few (a third of the chip's workgroup slots at 4096 envs): their launch goes to a side stream that is
    // forked from and joined back into `stream` with events, so that it fills the CUs the prey launch leaves idle.
    void *pred_stream = stream;
    // (direct-head networks: every workgroup needs a whole CU's LDS, so the two launches cannot share CUs -- a side stream only adds two
    //  cross-stream event waits; they run back to back on the caller's stream)
    const bool side = pred && prey && !(pred->direct && prey->direct);
    if (side) {
        if (!pred->side) {
            PPG_POL_TRY(pred, hipStreamCreateWithFlags(&pred->side, hipStreamNonBlocking));
            PPG_POL_TRY(pred, hipEventCreateWithFlags(&pred->fork, hipEventDisableTiming));
            PPG_POL_TRY(pred, hipEventCreateWithFlags(&pred->join, hipEventDisableTiming));
        }
        PPG_POL_TRY(pred, hipEventRecord(pred->fork, (hipStream_t)stream));
        PPG_POL_TRY(pred, hipStreamWaitEvent(pred->side, pred->fork, 0));
        pred_stream = pred->side;
    }
    const bool one_plan = pred && prey && !side;   // (same stream: the prey call's plan launch computes the predators' plan too)
    if (prey) {
        const int rc = ppg_policy_run(prey, 1, handles, n, actions, flags, seed, logits_prey, stream, one_plan ? pred : nullptr);
        if (rc != PPG_OK) { memcpy(g_ppg_policy_error, prey->err, sizeof g_ppg_policy_error); return rc; }
    }
    if (pred) {
        const int rc = ppg_policy_run(pred, 0, handles, n, actions, flags, seed, logits_pred, pred_stream, nullptr, one_plan);
        if (rc != PPG_OK) { memcpy(g_ppg_policy_error, pred->err, sizeof g_ppg_policy_error); return rc; }
    }
    if (side) {
        PPG_POL_TRY(pred, hipEventRecord(pred->join, pred->side));
        PPG_POL_TRY(pred, hipStreamWaitEvent((hipStream_t)stream, pred->join, 0));
    }
    return PPG_OK;
}

}  // extern "C"
