// ppg_policy_pipe.h -- the three-convolution direct-head network (ppg_policy_direct.h) as a TWO-ROLE pipeline: eight wavefronts per
// workgroup, one workgroup per CU, two wavefronts per SIMD that run DIFFERENT programs.
//
// ppg_policy_direct.h runs one wavefront per SIMD (512 registers: every weight resident) and nothing hides a stall but the code itself:
// its MFMA pipe is busy a third of the time (profiles/r04/b_policy_direct_*: conv3's MFMAs, epilogues and LDS reads ADD UP).  Two copies
// of that program do not fit a SIMD's registers -- but the network splits into two halves of about the same duration whose weights do:
//   wavefronts 0-3 (role A)   conv3 only: X -> F                                    152 registers of weights
//   wavefronts 4-7 (role B)   staging, conv1, conv2, the head's partial sums, logits  40 + 40 + 72 registers of weights
// and a SIMD that holds one of each interleaves A's dense MFMA stream with B's latency-bound chain in hardware.  Sub-group g (ST
// samples) moves through four iterations:
//   iteration g - 1   B: rows -> X[g & 1] (input blocks), conv1 -> Y, conv2 -> X[g & 1]
//   iteration g       A: conv3  X[g & 1] -> F[g & 1]
//   iteration g + 1   B: head partial sums  F[g & 1] -> red[g & 1]
//   iteration g + 2   B: bias + the four partial sums -> the workgroup's logits rows
// ONE workgroup barrier per iteration; X, F and `red` are double-buffered, Y is private to role B.  Inside an iteration the B wavefronts
// meet twice more (rows staged -> conv1 -> conv2 read across wavefronts): s_barrier would stop role A in the middle of its tiles, so
// those two are a counter in LDS that only the four B wavefronts touch (all eight are resident: no wavefront waits for one that is not
// running).  A workgroup's whole share of the samples is ONE tile (table in LDS, up to K.range_tile samples): the pipeline fills and
// drains once per launch.  Same arithmetic in the same order as ppg_policy_direct.h: the logits are bit-identical.
#pragma once

// Fragment reads per batch in role A's conv3.  LDS serves requests in arrival order, whatever the priority of the wavefront: five
// 1 KB reads per role-A wavefront queued at once stand in front of every LDS access of role B's chain (its table reads alone took
// 1000 cycles per sub-group); batches of 5 / 4 / 3 / 2: 0.362 / 0.356 / 0.350 / 0.347 ms per 4096-env step on one GPU; with area F swizzled and the LDS less crowded,
// batches of 3 / 2 / 1: 0.340 / 0.321 / 0.314 (profiles/r04)
#ifndef PPG_PIPE_B3
#define PPG_PIPE_B3 1
#endif

#ifndef PPG_PIPE_B12
#define PPG_PIPE_B12 2      // fragment reads per batch in role B's conv1 / conv2 (0 = a tile's ten at once): 0 / 5 / 4 / 3 / 2 / 1:
                            // 0.3118 / 0.3129 / 0.3077 / 0.3027 / 0.3007 / 0.3058 ms per step on one GPU (profiles/r04)
#endif
#ifndef PPG_PIPE_D3
#define PPG_PIPE_D3 1       // batches of fragment reads in flight ahead of the MFMAs in role A's conv3 (dconv's DEPTH)
#endif
#ifndef PPG_PIPE_D2
#define PPG_PIPE_D2 1       // the same in role B's conv2
#endif
#ifndef PPG_PIPE_CONV2_PAIR
#define PPG_PIPE_CONV2_PAIR 0   // role B's conv2: 1 = the wavefront's two position tiles as interleaved MFMA chains; 0 = tile after tile (dconv).
                                // Measured (profiles/r05/e_*): the pair is 1 % SLOWER end to end -- conv2's own phase stays at 1270 cycles
                                // and role A's conv3 on the same SIMD grows by what role B's denser MFMA stream takes from it
#endif
#ifndef PPG_PIPE_HEAD_CHAINS
#define PPG_PIPE_HEAD_CHAINS 1  // independent accumulator chains of the head's eighteen MFMAs: 2 measured 1 % slower than 1 (profiles/r05/e_*)
#endif
#ifndef PPG_PIPE_FETCH_EARLY
#define PPG_PIPE_FETCH_EARLY 0   // 1: the chunk loads of sub-group it + 2 go out right behind the staging of sub-group it + 1 (experiment, profiles/r05/r_*)
#endif
#ifndef PPG_PIPE_ABLATE
#define PPG_PIPE_ABLATE 0   // timing-only ablation builds (rounds 4-5, profiles/r05/h_*; never the product -- the results are then meaningless):
                            // 1 no head, 2 no staging, 4 no conv1, 8 no conv2, 16 no conv3, 32 no logits / actions, 64 no Gumbel noise,
                            // 128 no row fetch / park
#endif
#ifndef PPG_PIPE_HOFF
#define PPG_PIPE_HOFF 0   // 1: the head's eighteen operand offsets in registers instead of recomputed per sub-group -- measured 2 % SLOWER
                          // (profiles/r05/e_*: eighteen more live registers in role B cost more than the address arithmetic)
#endif
#ifndef PPG_PIPE_CONV1X_LOOP
#define PPG_PIPE_CONV1X_LOOP 0  // conv1x tile after tile (1) instead of the four tiles as one straight line (0)
#endif
#ifndef PPG_PIPE_SWP_B
#define PPG_PIPE_SWP_B true   // role B's convolution loops software-pipelined (the epilogue of tile t behind the reads of tile t + 1)
#endif
#ifndef PPG_PIPE_SLEEP
#define PPG_PIPE_SLEEP 1    // s_sleep argument between two polls of role B's private barrier
#endif
#ifndef PPG_PIPE_PRIO_B
#define PPG_PIPE_PRIO_B 3   // s_setprio of role B's wavefronts (role A: 0)
#endif

namespace ppgpol {

// Role B's private barrier, in two halves: a wavefront ARRIVES (its LDS writes have landed, the counter goes up) and WAITS for the
// `target`-th arrival later -- with work that does not depend on the other wavefronts in between, the wait finds the counter there.
// Monotonic counter, never reset.
__device__ __forceinline__ void pipe_arrive(uint32_t *ctr, int lane) {
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LDS serves a wavefront in order)
    if (lane == 0) (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// -DPPG_PIPE_DEBUG (tools/build_pipe_debug.sh -> tools/_build/libppg_hip_pipedbg.so; never the product): the wait gives up after
// PPG_PIPE_DEBUG_POLLS polls and reports (workgroup, wavefront, counter, target) through the policy's status words, which
// ppg_policy_act reads back after every launch of that build -- a lost arrival is an error message instead of a hung GPU.
#ifndef PPG_PIPE_DEBUG_POLLS
#define PPG_PIPE_DEBUG_POLLS 2000000
#endif
#ifdef PPG_PIPE_DEBUG
__device__ uint32_t *g_pipe_status;   // set per launch by the host side (hipMemcpyToSymbol): [0] count, [1..4] the first report
#endif
__device__ __forceinline__ void pipe_wait(uint32_t *ctr, uint32_t target) {
#ifdef PPG_PIPE_DEBUG
    int polls = 0;
#endif
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target) {
        __builtin_amdgcn_s_sleep(PPG_PIPE_SLEEP);
#ifdef PPG_PIPE_DEBUG
        if (++polls > PPG_PIPE_DEBUG_POLLS) {
            if ((threadIdx.x & 63u) == 0 && g_pipe_status && atomicAdd(&g_pipe_status[0], 1u) == 0u) {
                g_pipe_status[1] = blockIdx.x; g_pipe_status[2] = threadIdx.x >> 6;
                g_pipe_status[3] = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); g_pipe_status[4] = target;
            }
            break;
        }
#endif
    }
    __asm__ volatile("" ::: "memory");
}
__device__ __forceinline__ void pipe_bsync(uint32_t *ctr, uint32_t target, int lane) {
    pipe_arrive(ctr, lane);
    pipe_wait(ctr, target);
}

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
constexpr int PIPE_CHUNKS = 3;   // 8-byte chunks of observation rows a role-B thread fetches per sub-group (ppg_policy_create_spec: pipe_ni)

// Handle of global env e WITHOUT a per-lane index into the parameter block (K.obs[k] with k in a vector register is a vector load
// from the kernarg segment: a memory round trip in front of every address that depends on it): the handles' base pointers are read as
// scalars and picked by comparison.  Returns the env's index inside its handle.
template <class KP, class Bases, class PtrT>
__device__ __forceinline__ int pipe_pick_handle(const KP &K, const Bases &bases, int e, PtrT &base) {
    base = bases[0];
    int eb = 0;
#pragma unroll
    for (int q = 1; q < MAX_HANDLES; ++q) {
        const bool in = q < K.n_handles && e >= K.env_base[q];
        base = in ? bases[q] : base;
        eb = in ? K.env_base[q] : eb;
    }
    return e - eb;
}

// conv1 of the pipeline on v_mfma_f32_16x16x32_bf16 (round 5).  The 32x32x16 form pads conv1 twice -- its 16 output channels to 32 MFMA
// rows, and (channels-last 9x9 windows) its nine input channels to two blocks of eight: ten k-steps of which a quarter is real work.
// Here M = the 16 output channels, N = 16 positions, and one k-step per kernel ROW ky holds everything that row contributes: the lane
// quarters kq = 0, 1, 2 read the three taps' cells of channel block 0 (eight channels each), kq = 3 reads the position's cell of block
// 1, which the staging fills with the NINTH channel of the three taps {c8(x - 1), c8(x), c8(x + 1), 0 ...}.  Three MFMAs of 16 cycles per
// 16 positions instead of ten of 32 per 32; 12 registers of weights instead of 40; the bias is the accumulators' initial value.
// Up to nine input channels (the reference's 7x7 / 9x9 windows); wider channels-last windows run the one-role kernels.
struct Conv1X {
    bf16x8 a[3];     // weight fragments of ky = 0, 1, 2: lane (r = lane & 15: output channel, kq = lane >> 4)
    f32x4_t bias;    // of this lane's output channels 4 kq .. 4 kq + 3
    int cell[4];     // element offset (from img) of this lane's position in its tile t: block 0 of area X0, cell of the position
    int smp[4];      // ... and its sample (255: the slot holds no position)
    template <class KP>
    __device__ __forceinline__ void load(const KP &K, int lane, int bw) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) a[ky] = ((const GLOBAL_AS bf16x8 *)K.wc1x)[ky * 64 + lane];
        bias = *(const GLOBAL_AS f32x4_t *)(K.bc1x + 4 * (lane >> 4));
        const int col = lane & 15;
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // (the slot table: dconv_cells)
            const int n = 16 * (bw + 4 * t) + col;
            const uint32_t e = n < 32 * K.slot_tiles ? (uint32_t)K.slot_tab[n] : 0xFFFFu;
            const int sv = (int)(e >> 8), s = sv == 255 ? 0 : sv, p = sv == 255 ? 0 : (int)(e & 255u);
            const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
            cell[t] = __mul24(s, K.sample_stride) + (__mul24(y + 1, K.Wp) + (x + 1)) * 8;
            smp[t] = sv;
        }
    }
    __device__ __forceinline__ void landed() {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            u32x4_t v = __builtin_bit_cast(u32x4_t, a[ky]);
            __asm__ volatile("" : "+v"(v));
            a[ky] = __builtin_bit_cast(bf16x8, v);
        }
        __asm__ volatile("" : "+v"(bias));
    }
};
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// ReLU + round four accumulator registers to bf16 (relu_pack8's arithmetic)
__device__ __forceinline__ u32x2_t relu_pack4(const f32x4_t &a) {
    u32x2_t w;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x2 f = {a[2 * j], a[2 * j + 1]};
        const s16x2 zero = {0, 0};
        const s16x2 q = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2)), zero);
        w[j] = __builtin_bit_cast(uint32_t, q);
    }
    return w;
}
// One sub-group's conv1: X (input blocks at element offset xo) -> Y.  CB1 = 2: a second input block (the ninth channel's taps).
// LOOP: tile after tile (12 registers of operands) instead of the straight line below.
template <int CB1, bool LOOP = (PPG_PIPE_CONV1X_LOOP != 0), class KP>
__device__ __forceinline__ void conv1x(const KP &K, const Conv1X &W, __bf16 *img, int xo, int ns, int bw, int lane, int dummy) {
    const int kq = lane >> 4;
    const int blk = K.Wp2 * 8;
    // this lane quarter's B operand of kernel row ky = 1 relative to the position's cell: the tap kq - 1 of block 0, or (kq = 3)
    // the position's own cell of block 1 (CB1 = 1: any cell of block 0 -- its weights are zero)
    const int koff = kq < 3 ? (kq - 1) * 8 : (CB1 > 1 ? blk : 0);
    const int yoff = (kq >> 1) * blk + (kq & 1) * 4;    // where this lane's four output channels go inside the position's Y cells
    if constexpr (LOOP) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (ns <= 0) break;
        const bool valid = W.smp[t] < ns;
        const __bf16 *base = img + W.cell[t] + xo + koff;
        bf16x8 b[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) b[ky] = *(const bf16x8 *)(base + (ky - 1) * K.Wp * 8);
        f32x4_t acc = W.bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.a[ky], b[ky], acc, 0, 0, 0);
        const int at = valid ? W.cell[t] + K.off_y + yoff : dummy;
        *(u32x2_t *)(img + at) = relu_pack4(acc);
    }
    return;
    }
    // The wavefront's four tiles as ONE straight line: twelve fragment reads in front, then the four tiles' MFMA chains interleaved (a
    // tile's three MFMAs depend on each other through the accumulator: tile by tile the chain ran at the matrix pipe's LATENCY, 290
    // cycles per tile for 48 cycles of issue), then the four epilogues.  Tiles behind the sub-group's last position compute position 0
    // again and store into the dummy slot (the last sub-group of a share only).
    bf16x8 b[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const __bf16 *base = img + W.cell[t] + xo + koff;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) b[t][ky] = *(const bf16x8 *)(base + (ky - 1) * K.Wp * 8);
    }
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = W.bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.a[ky], b[t][ky], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool valid = W.smp[t] < ns;
        const int at = valid ? W.cell[t] + K.off_y + yoff : dummy;
        *(u32x2_t *)(img + at) = relu_pack4(acc[t]);
    }
}

// Role B's conv2 (16 -> 32 channels, one 32-row MFMA tile) over the wavefront's TWO position tiles at once: a tile's ten MFMAs depend on
// each other through its accumulators (64 cycles from one to the next), so tile after tile the layer ran at that latency -- 1270 cycles
// for 640 of issue.  Two independent chains interleaved fill the gaps.  Fragment reads two k-steps (four reads) ahead.
template <class KP>
__device__ __forceinline__ void conv2_pair(const KP &K, const ConvW<2, 1> &W, __bf16 *img, int in_off, int out_off, int out_blocks, int ns,
                                           int bw, int lane, int dummy, const int *cells) {
    constexpr int KS = ConvW<2, 1>::KS, KSB = ConvW<2, 1>::KS_BIAS, HB = ConvW<2, 1>::H_BIAS;
    const int h = lane >> 5;
    const int blk = K.Wp2 * 8;
    if (ns <= 0) return;
    const int in0 = in_off + h * blk;
    const int cb0 = 2 * h;
    auto fragment = [&](int cell, int ks) -> bf16x8 {
        bf16x8 v;
        if (ks == KSB && HB == 0) {   // (the whole k-step is the bias block + nothing)
            v = zero8();
            if (h == HB) { v[0] = (__bf16)1.0f; v[1] = (__bf16)1.0f; }
        } else {
            v = *(const bf16x8 *)(img + cell + in0 + W.offset(K, ks, 2 * blk));
            if (ks == KSB && h == HB) { v = zero8(); v[0] = (__bf16)1.0f; v[1] = (__bf16)1.0f; }
        }
        return v;
    };
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    bf16x8 b[3][2];   // a rolling window of three k-steps x two tiles
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) b[d][t] = fragment(cells[2 * t], d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 2 < KS) {
#pragma unroll
            for (int t = 0; t < 2; ++t) b[(ks + 2) % 3][t] = fragment(cells[2 * t], ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.a[0][ks], b[ks % 3][t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const bool valid = cells[4 + t] < ns;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool real = valid && (cb0 + j < out_blocks);
            const int at = real ? cells[2 * t] + out_off + (cb0 + j) * blk : dummy;
            *(bf16x8 *)(img + at) = relu_pack8(acc[t], 8 * j);
        }
    }
}

// The fused launch's plan, computed by every workgroup for itself (no plan launch, no trip through memory): exclusive prefix sums of the
// envs' predator / prey row counts (env words 0, 1: one 8-byte load per env) over the concatenated envs of all handles, in LDS:
// scratch = [512 threads][2] partial sums | pre_pred[n_envs] | pre_prey[n_envs].  512 threads; ends with a workgroup barrier.
constexpr int FUSED_PART_WORDS = 1024, FUSED_MAX_ENVS = 8192;
template <int NT, class KP>   // NT threads (512: the two-role kernels)
__device__ __forceinline__ void fused_prefix_sums(const KP &K, uint32_t *scratch, int tid, uint32_t &tot_pred, uint32_t &tot_prey) {
    static_assert(PPG_ENV_N_PRED_ROWS == 0 && PPG_ENV_N_PREY_ROWS == 1, "the two row counts are one 8-byte load");
    constexpr int FUSED_RUN = FUSED_MAX_ENVS / NT, NWAVES = NT / 64;
    const int per = (K.n_envs + NT - 1) / NT;
    const int lo = tid * per < K.n_envs ? tid * per : K.n_envs, hi = (lo + per) < K.n_envs ? (lo + per) : K.n_envs;
    auto count_of = [&](int e) -> u32x2_t {
        const int32_t *base;
        const int b = pipe_pick_handle(K, K.env_state, e, base);
        return *(const GLOBAL_AS u32x2_t *)(base + (size_t)b * PPG_ENV_WORDS);
    };
    u32x2_t cnt[FUSED_RUN];
    uint32_t sp = 0, sq = 0;
#pragma unroll
    for (int i = 0; i < FUSED_RUN; ++i)   // (unconditional loads of a clamped env, all in flight at once: see pipe_main's fetch)
        cnt[i] = count_of((lo + i) < K.n_envs ? (lo + i) : K.n_envs - 1);
#pragma unroll
    for (int i = 0; i < FUSED_RUN; ++i)
        if (i < per && lo + i < hi) { sp += cnt[i][0]; sq += cnt[i][1]; }
    // inclusive scan over the 512 threads: inside a wavefront on the cross-lane paths, the eight wavefronts' totals through LDS --
    // two barriers (a Hillis-Steele scan over 512 LDS words took eighteen, in front of every workgroup's first sample)
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t ip = sp, iq = sq;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t vp = (uint32_t)__shfl_up((int)ip, d, 64), vq = (uint32_t)__shfl_up((int)iq, d, 64);
        if (lane >= d) { ip += vp; iq += vq; }
    }
    u32x2_t *part = (u32x2_t *)scratch;   // [NWAVES] wavefront totals
    if (lane == 63) { const u32x2_t t = {ip, iq}; part[wave] = t; }
    __syncthreads();
    uint32_t bp = ip - sp, bq = iq - sq, allp = 0, allq = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
        const u32x2_t t = part[w];
        if (w < wave) { bp += t[0]; bq += t[1]; }
        allp += t[0]; allq += t[1];
    }
    uint32_t *pre_pred = scratch + FUSED_PART_WORDS, *pre_prey = pre_pred + K.n_envs;
#pragma unroll
    for (int i = 0; i < FUSED_RUN; ++i)
        if (i < per && lo + i < hi) {
            pre_pred[lo + i] = bp; pre_prey[lo + i] = bq;
            bp += cnt[i][0]; bq += cnt[i][1];
        }
    tot_pred = (uint32_t)__builtin_amdgcn_readfirstlane((int)allp);
    tot_prey = (uint32_t)__builtin_amdgcn_readfirstlane((int)allq);
    __syncthreads();
}

// FUSED = false: one launch per species, the plan (totals, shares, first env of every tile) comes from ppg_policy_plan[2] in memory.
// FUSED = true (ppg_policy_pipe2_*, below): BOTH species in one launch and no plan launch -- the caller (fused_main) has computed the
// envs' exclusive prefix sums of this species' row counts into LDS (`pre`, n_envs words behind the image area's start) and hands this
// workgroup its place `wg` among the `n_wgs` workgroups that serve the species and the species' total `N_`.
template <int OBS, int NCH, bool FUSED = false>
__device__ __forceinline__ void pipe_main(KPtr Kp, unsigned char *lds, int wg = 0, int n_wgs = 0, int N_ = 0, uint32_t *scratch = nullptr,
                                          const uint32_t *pre_g = nullptr) {
    constexpr int CB1 = NCH > 8 ? 2 : 1, HF = 18;
    const auto &K = *Kp;
    const uint32_t *pre = FUSED ? scratch + FUSED_PART_WORDS + (K.species ? K.n_envs : 0) : nullptr;
    const int tid = (int)threadIdx.x, lane = tid & 63, btid = tid & 255;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_b = wave >= 4;
    const int bw = wave & 3;
    unsigned long long *tab = (unsigned long long *)lds;                    // [range_tile][2]: observation row; global env index | row << 32
    float *red = (float *)(lds + K.pipe_red);                               // [2][wavefront][16 actions][16 samples]
    uint32_t *ctr = (uint32_t *)(lds + K.pipe_red + 8192);
    float *noise = (float *)(lds + K.pipe_red + 8192 + 64);                 // [2][16 samples][16 actions]: Gumbel noise of two sub-groups
    __bf16 *img = (__bf16 *)(lds + K.pipe_img);
    const int dummy = -512 + 8 * lane;   // (element index from img: this lane's 16 bytes of the 1 KB in front of the images; dconv.  Shared
                                         //  by the wavefronts: what lands there is never read)
    const int sample_stride = K.sample_stride;
    int N, share, tpw;
    if (FUSED) {   // every workgroup of the species the same number of whole sub-groups (the plan kernel's range mode)
        N = N_;
        const int sg = (N + K.ST - 1) / K.ST;
        // (integer divisions run on the vector unit: their results are pinned back into scalar registers -- the plan's words came from
        //  scalar loads, and everything derived from them is loop control)
        share = __builtin_amdgcn_readfirstlane(K.ST * ((sg + n_wgs - 1) / n_wgs));
        tpw = __builtin_amdgcn_readfirstlane(share ? (share + K.range_tile - 1) / K.range_tile : 1);
    } else {
        N = (int)K.plan[0]; share = (int)K.plan[1]; tpw = (int)K.plan[2];
        wg = (int)blockIdx.x;
    }
    const int begin = wg * share, end = (begin + share) < N ? (begin + share) : N;
    if (begin >= end) return;
    const int n_slots = (int)gridDim.x * tpw;
    auto zero_images = [&] {
        for (int i = tid; i < (K.ST * sample_stride) / 8 + 18 * 4; i += 512) ((bf16x8 *)img)[i] = zero8();
    };
    if (!FUSED) {
        zero_images();
        if (tid == 0) *ctr = 0u;
    }
    typedef typename ObsRaw<OBS, NCH>::type raw_t;
    typedef typename ObsRaw<OBS, NCH>::elem elem_t;
    // -DPPG_DIRECT_PROFILE: cycles per phase and wavefront -> K.xg [workgroup][8][16].  Role A: 0 table, 1 conv3, 2 waiting at the barrier,
    // 14 actions.  Role B: 0 table, 3 logits, 4 head, 5 staging + requests, 6 first private barrier, 7 conv1, 8 second private barrier,
    // 9 conv2, 10 waiting at the barrier, 14 actions.  15 = iterations
#ifdef PPG_DIRECT_PROFILE
    long long dp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dp_prev = (long long)clock64();
    auto dp_dump = [&] {
        if (K.xg && lane == 0)
            for (int i = 0; i < 16; ++i)   // ([13]: species + 1 -- the fused launch's workgroups serve either)
                ((unsigned long long *)K.xg)[((size_t)blockIdx.x * 8 + wave) * 16 + i] = i == 13 ? (unsigned long long)(K.species + 1) : (unsigned long long)dp_acc[i];
    };
#else
    auto dp_dump = [] {};
#endif

    const int kq = lane >> 4, colh = lane & 15;
    const int per = (K.kflat_steps + 3) >> 2, k_lo = bw * per;
    const int smp = btid >> 4, a16 = btid & 15;

    // the tile's sample table: sample -> (observation row; global env index, row), bisection over the envs' prefix sums
    auto build_table = [&](int tile, int n0, int nt_samples) {
        __syncthreads();   // the previous tile's last readers of the table are done (and the zero fill has landed)
        for (int i = tid; i < nt_samples; i += 512) {
            const uint32_t n = (uint32_t)(n0 + i);
            int lo, hi;
            if (FUSED) { lo = 0; hi = K.n_envs - 1; }
            else { lo = (int)K.tile_env[tile]; hi = tile + 1 < n_slots ? (int)K.tile_env[tile + 1] : K.n_envs - 1; }
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                // (FUSED: the first tile's prefix sums are in LDS; the images have overwritten them by the time a long share's later
                //  tiles come -- those read the copy this workgroup put into memory, fused_main)
                const uint32_t at = !FUSED ? K.plan[PLAN_HDR + mid] : n0 == begin ? pre[mid] : __builtin_nontemporal_load(&pre_g[mid]);
                if (at <= n) lo = mid; else hi = mid - 1;
            }
            const int e = lo;
            const int row = (int)(n - (!FUSED ? K.plan[PLAN_HDR + e] : n0 == begin ? pre[e] : __builtin_nontemporal_load(&pre_g[e])));
            const unsigned char *base;
            const int b = pipe_pick_handle(K, K.obs, e, base);
            tab[2 * i] = (unsigned long long)(uintptr_t)(base + ((size_t)b * K.cap + row) * (size_t)K.obs_elems * (OBS == 2 ? 2 : OBS == 1 ? 4 : 8));
            tab[2 * i + 1] = (unsigned long long)(uint32_t)e | ((unsigned long long)(uint32_t)row << 32);
        }
        __syncthreads();
        if (FUSED) {   // the prefix sums lived in the image area: only now can the halo cells be zeroed
            zero_images();
            if (tid == 0 && n0 == begin) *ctr = 0u;
            __syncthreads();
        }
    };

    // The two roles are two separate loops (not two branches inside one): inside one loop the register allocator would have to keep BOTH
    // roles' weights alive.  Both execute the same sequence of workgroup barriers.
    if (!role_b) {
        // ================= role A: conv3; logits and actions of the sub-group two iterations back =================
        ConvW<4, 2> w3c;
        w3c.load(K, K.wc3, lane, 0);
        const float bias_r = (a16 < K.n_actions) ? K.bh[a16] : 0.0f;
        // the Philox key: the launch's argument, or (PPG_POLICY_SEED_ON_DEVICE: a step replayed from a hipGraph) a word in memory
        uint32_t seed_lo = K.seed_lo, seed_hi = K.seed_hi;
        if (K.seed_dev) {
            const uint64_t sd = *K.seed_dev;
            seed_lo = (uint32_t)sd ^ (K.species ? 0x9E3779B9u : 0u);
            seed_hi = (uint32_t)(sd >> 32);
        }
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w3c.landed();
        int cells[6];   // this wavefront's two position tiles are the same positions in every sub-group: their LDS offsets once per launch
        dconv_cells(K, sample_stride, wave, 4, lane, cells);
        for (int j = 0; j < tpw; ++j) {
            const int tile = (int)blockIdx.x * tpw + j;
            const int n0 = begin + j * K.range_tile;
            if (n0 >= end) break;
            const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
            const int G = (nt_samples + K.ST - 1) / K.ST;
            build_table(tile, n0, nt_samples);
            PPG_DP(0);
            for (int it = -1; it <= G + 1; ++it) {
                // thread (sample smp, action a16) of sub-group it - 2: bias + the four partial sums in wavefront order = the logit; then
                // the sixteen lanes of a sample pick the action: argmax, or Gumbel-max -- the first maximum wins, as in a loop over the
                // actions.  The Gumbel noise was computed one iteration ago by ANOTHER wavefront (below) and waits in LDS.
                // Its LDS reads (partial sums, table entry, noise: complete since the iteration's barrier) are issued HERE, in front of
                // conv3, and consumed behind it: three dependent LDS round trips less on the wavefronts that carry the most work.
                const bool do_act = !(PPG_PIPE_ABLATE & 32) && it >= 2 && 4 * wave < K.ST && 4 * wave < nt_samples - (it - 2) * K.ST;   // (this wavefront's four samples: 4 wave .. 4 wave + 3)
                float act_part[4] = {0.0f, 0.0f, 0.0f, 0.0f}, act_noise = 0.0f;
                unsigned long long act_er = 0;
                int act_ns = 0, act_s_local = 0;
                if (do_act) {
                    const int g = it - 2, left = nt_samples - g * K.ST;
                    act_ns = left < K.ST ? left : K.ST;
                    const float *rd = red + (g & 1) * 1024;
#pragma unroll
                    for (int w = 0; w < 4; ++w) act_part[w] = rd[(w * 16 + a16) * 16 + smp];
                    act_s_local = g * K.ST + (smp < act_ns ? smp : 0);
                    act_er = tab[2 * act_s_local + 1];
                    if (K.sample) act_noise = noise[(g & 1) * 256 + btid];
                }
                PPG_DP(3);
                if (!(PPG_PIPE_ABLATE & 16) && it >= 0 && it < G) {
                    const int left = nt_samples - it * K.ST, ns = left < K.ST ? left : K.ST;
                    if (K.flat_c == 64)
                        dconv<4, 2, PPG_PIPE_B3, false, true, PPG_PIPE_D3>(K, w3c, img, sample_stride, (it & 1) ? K.pipe_x1 : 0, (it & 1) ? K.pipe_f1 : K.off_f,
                                                              K.cout_blocks[2], 64, ns, wave, 4, lane, 0, dummy, cells);
                    else
                    dconv<4, 2, PPG_PIPE_B3, false, false, PPG_PIPE_D3>(K, w3c, img, sample_stride, (it & 1) ? K.pipe_x1 : 0, (it & 1) ? K.pipe_f1 : K.off_f,
                                                      K.cout_blocks[2], K.flat_c, ns, wave, 4, lane, 0, dummy);
                }
                PPG_DP(1);
                if (do_act) {
                    const int ns = act_ns, s_local = act_s_local;
                    float v = bias_r;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += act_part[w];
                    const unsigned long long er = act_er;
                    const uint32_t e = (uint32_t)er, row = (uint32_t)(er >> 32);
                    if (K.logits && smp < ns && a16 < K.n_actions) K.logits[(size_t)(n0 + s_local) * K.n_actions + a16] = v;
                    if (K.sample) v += act_noise;
                    if (a16 >= K.n_actions) v = -INFINITY;
                    int best = a16;
                    // all-reduce over the sample's sixteen lanes on the DPP cross-lane paths: partners lane ^ 1, lane ^ 2, then 7 - lane and
                    // 15 - lane (every lane of the row ends with the same maximum)
#define PPG_PIPE_ARGMAX_STEP(CTRL)                                                                                   \
                    {                                                                                                \
                        const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false)); \
                        const int ob = __builtin_amdgcn_update_dpp(0, best, CTRL, 0xF, 0xF, false);                  \
                        const bool take = ov > v || (!(v > ov) && ob < best);                                        \
                        v = take ? ov : v;                                                                           \
                        best = take ? ob : best;                                                                     \
                    }
                    PPG_PIPE_ARGMAX_STEP(0xB1)    // quad_perm [1, 0, 3, 2]
                    PPG_PIPE_ARGMAX_STEP(0x4E)    // quad_perm [2, 3, 0, 1]
                    PPG_PIPE_ARGMAX_STEP(0x141)   // row_half_mirror
                    PPG_PIPE_ARGMAX_STEP(0x140)   // row_mirror
#undef PPG_PIPE_ARGMAX_STEP
                    if (a16 == 0 && smp < ns) {
                        int8_t *base;
                        const int b = pipe_pick_handle(K, K.actions, (int)e, base);
                        base[(size_t)b * K.S + K.slot0 + row] = (int8_t)best;
                    }
                }
                PPG_DP(14);
                // Gumbel noise of sub-group it - 1 (needed in the next iteration): -log(-log u), u from Philox keyed by (seed, env, row
                // slot, action) as in phase_head.  Sample s is served by wavefront (s / 4 + 2) & 3: with 7 samples per sub-group the
                // wavefronts 2, 3 -- which have no sample to pick actions for and would wait at the barrier -- do all of it.
                if (!(PPG_PIPE_ABLATE & 64) && K.sample && it >= 1 && it - 1 < G) {
                    const int g = it - 1, left = nt_samples - g * K.ST, ns = left < K.ST ? left : K.ST;
                    const int s0 = 4 * ((wave + 2) & 3);
                    if (s0 < ns) {
                        const int sm = s0 + (lane >> 4);
                        const unsigned long long er = tab[2 * (g * K.ST + (sm < ns ? sm : 0)) + 1];
                        uint32_t rnd[4];
                        philox((uint32_t)er, (uint32_t)K.slot0 + (uint32_t)(er >> 32), (uint32_t)(a16 >> 2), 0x504F4C31u, seed_lo, seed_hi, rnd);
                        const uint32_t r = (a16 & 3) == 0 ? rnd[0] : (a16 & 3) == 1 ? rnd[1] : (a16 & 3) == 2 ? rnd[2] : rnd[3];
                        const float u = (float)(r >> 9) * (1.0f / 8388608.0f) + (1.0f / 16777216.0f);   // 23 bits: 2^-24 <= u < 1, exactly
                        noise[(g & 1) * 256 + sm * 16 + a16] = -__logf(-__logf(u));
                    }
                }
                PPG_DP(5);
                __syncthreads();
                PPG_DP(2);
            }
        }
        dp_dump();
        return;
    }
    // ================= role B: rows -> X, conv1, conv2; head =================
    // Role B is the longer chain, and its MFMAs are few and dependent: with equal priorities the SIMD serves role A's dense MFMA stream
    // first and the head's 18 MFMAs take as long as the whole of conv3 (profiles/r04) -- B goes first whenever it has something to issue.
    __builtin_amdgcn_s_setprio(PPG_PIPE_PRIO_B);
    Conv1X w1x;
    ConvW<2, 1> w2c;
    bf16x8 hf[HF];
    w1x.load(K, lane, bw);
    w2c.load(K, K.wc2, lane);
#pragma unroll
    for (int i = 0; i < HF; ++i) hf[i] = ((const GLOBAL_AS bf16x8 *)K.whw)[((size_t)bw * HF + i) * 64 + lane];
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    w1x.landed();
    w2c.landed();
#pragma unroll
    for (int i = 0; i < HF; ++i) {
        u32x4_t v = __builtin_bit_cast(u32x4_t, hf[i]);
        __asm__ volatile("" : "+v"(v));
        hf[i] = __builtin_bit_cast(bf16x8, v);
    }
    uint32_t b_target = 0;
#if PPG_PIPE_HOFF
    int hoff[HF];   // where this lane's eighteen head operands lie in a sample's area F: the same in every sub-group -- once per launch
#pragma unroll
    for (int i = 0; i < HF; ++i) { hoff[i] = f_koff(K, k_lo + i, kq); __asm__ volatile("" : "+v"(hoff[i])); }
#endif
    int cells[6];   // (as in role A: the positions of this wavefront's two tiles, once per launch)
    dconv_cells(K, sample_stride, bw, 4, lane, cells);
    // bfloat16 rows whose size is a multiple of 8 bytes (K.pipe_ni > 0) come in as they lie in HBM: ALIGNED 8-byte chunks, consecutive
    // lanes consecutive chunks (two or three loads per thread and sub-group), parked in the LDS area `raw` one iteration later and picked
    // apart by the position threads from there.  One 2-byte load per channel and position -- 64 scattered lanes per instruction, nine
    // instructions per wavefront -- kept the texture addresser busy for 1500 cycles per sub-group; so did unaligned 16-byte loads.
    const bool row_chunks = OBS == 2 && K.pipe_ni > 0;
    unsigned char *raw = lds + K.pipe_raw;
    auto row_of = [&](int s_tile) -> const GLOBAL_AS unsigned char * { return (const GLOBAL_AS unsigned char *)(uintptr_t)tab[2 * s_tile]; };
    // a thread's place in the sub-group never changes: the position it stages, and (row chunks) chunk ch_w of the samples ch_q,
    // ch_q + K.pipe_slots, ...: consecutive lanes consecutive chunks of one row
    // (thread t stages the position of slot t: the eight lanes of a store group then write eight different bank groups)
    const uint32_t st_e = btid < 32 * K.slot_tiles ? (uint32_t)K.slot_tab[btid] : 0xFFFFu;
    const int st_sv = (int)(st_e >> 8), st_s = st_sv == 255 ? 0 : st_sv, st_p = st_sv == 255 ? 0 : (int)(st_e & 255u);
    const int st_y = div_small(st_p, K.magic_R), st_x = st_p - __mul24(st_y, K.IW);
    const int st_img = __mul24(st_s, sample_stride) + (__mul24(st_y + 1, K.Wp) + (st_x + 1)) * 8;
    const int st_raw = __mul24(st_s, K.obs_elems) + st_p * K.p_stride;
    const int cpr = K.obs_elems >> 2;   // 8-byte chunks per bfloat16 row
    const int ch_q = row_chunks ? (int)__umulhi((uint32_t)btid, K.pipe_magic) : 0, ch_w = btid - ch_q * cpr;
    // (the tile loop exists twice, with and without row chunks: a run-time switch inside the loop would put the chunk registers into
    //  merges with undefined values -- register copies behind the loads, i.e. waits)
    auto b_tiles = [&](auto ch_tag) {
    constexpr bool CH = decltype(ch_tag)::value;
    for (int j = 0; j < tpw; ++j) {
        const int tile = (int)blockIdx.x * tpw + j;
        const int n0 = begin + j * K.range_tile;
        if (n0 >= end) break;
        const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
        const int G = (nt_samples + K.ST - 1) / K.ST;
        build_table(tile, n0, nt_samples);
        auto group_ns = [&](int g) { const int left = nt_samples - g * K.ST; return left < K.ST ? left : K.ST; };
        raw_t pre[NCH];
        u32x2_t chunk[PIPE_CHUNKS];
        auto fetch = [&](int g) {   // row chunks: chunk ch_w of this thread's samples of sub-group g -> registers
            // (every load unconditional, the sample clamped: a load under a condition is merged with its default value by a register
            //  copy right behind it -- a wait for the whole memory round trip, 1500-3700 cycles per sub-group: profiles/r04)
            const int last = group_ns(g) - 1;
#pragma unroll
            for (int k = 0; k < PIPE_CHUNKS; ++k) {
                const int s0 = ch_q + K.pipe_slots * k, s = s0 < last ? s0 : last;
                chunk[k] = *(const GLOBAL_AS u32x2_t *)(row_of(g * K.ST + s) + 8 * ch_w);
            }
        };
        auto park = [&](int g, bool valid) {    // ... -> `raw` (behind conv1 and conv2: the loads have had that long to land)
            const int ns = valid && ch_q < K.pipe_slots ? group_ns(g) : 0;
#pragma unroll
            for (int k = 0; k < PIPE_CHUNKS; ++k) {
                const int s = ch_q + K.pipe_slots * k;
                if (s < ns) ((u32x2_t *)raw)[s * cpr + ch_w] = chunk[k];
            }
        };
        auto request = [&](int g) {   // (a B thread stages one position: ST * P <= 256, ppg_policy_create_spec)
            const int ns = group_ns(g);
#pragma unroll
            for (int c = 0; c < NCH; ++c) pre[c] = (raw_t)0;
            if (st_sv < ns) {
                const GLOBAL_AS elem_t *src = (const GLOBAL_AS elem_t *)row_of(g * K.ST + st_s) + st_p * K.p_stride;
#pragma unroll
                for (int c = 0; c < (NCH < 9 ? NCH : 9); ++c) if (c < K.cin) pre[c] = (raw_t)src[c * K.c_stride];
                if constexpr (CB1 > 1) {   // the ninth channel of the two neighbours along the image row (conv1x's block 1)
                    pre[9] = (raw_t)src[8 * K.c_stride - (st_x > 0 ? K.p_stride : 0)];
                    pre[10] = (raw_t)src[8 * K.c_stride + (st_x < K.IW - 1 ? K.p_stride : 0)];
                }
            }
        };
        auto stage = [&](int g) {
            const int ns = group_ns(g);
            if (st_sv < ns) {
                if constexpr (CH) {
                    const uint16_t *rh = (const uint16_t *)raw + st_raw;
#pragma unroll
                    for (int c = 0; c < (NCH < 9 ? NCH : 9); ++c) pre[c] = (c < K.cin) ? (raw_t)rh[c * K.c_stride] : (raw_t)0;
                    if constexpr (CB1 > 1) {
                        pre[9] = (raw_t)rh[8 * K.c_stride - (st_x > 0 ? K.p_stride : 0)];
                        pre[10] = (raw_t)rh[8 * K.c_stride + (st_x < K.IW - 1 ? K.p_stride : 0)];
                    }
                }
                {   // block 0: channels 0-7 of the position
                    bf16x8 v = zero8();
#pragma unroll
                    for (int c = 0; c < (NCH < 8 ? NCH : 8); ++c) v[c] = ObsRaw<OBS, NCH>::to_bf16(pre[c]);
                    *(bf16x8 *)(img + st_img + ((g & 1) ? K.pipe_x1 : 0)) = v;
                }
                if constexpr (CB1 > 1) {   // block 1: the ninth channel of the taps x - 1, x, x + 1 (conv1x); zero outside the image row
                    bf16x8 v = zero8();
                    const __bf16 z = (__bf16)0.0f;
                    v[0] = st_x > 0 ? ObsRaw<OBS, NCH>::to_bf16(pre[9]) : z;
                    v[1] = ObsRaw<OBS, NCH>::to_bf16(pre[8]);
                    v[2] = st_x < K.IW - 1 ? ObsRaw<OBS, NCH>::to_bf16(pre[10]) : z;
                    *(bf16x8 *)(img + st_img + ((g & 1) ? K.pipe_x1 : 0) + K.Wp2 * 8) = v;
                }
            }
        };
        if constexpr (CH) {   // sub-group 0's rows are in `raw` before any B wavefront stages a position
            fetch(0);
            park(0, true);
            b_target += 4;
            pipe_bsync(ctr, b_target, lane);
        } else {
            request(0);
        }
        PPG_DP(0);
        for (int it = -1; it <= G + 1; ++it) {
            // order inside the iteration: stage -> ARRIVE -> head (needs nothing of the other B wavefronts) -> WAIT -> conv1 -> ARRIVE -> row
            // fetch -> WAIT -> conv2: the waits of the two private barriers find most arrivals done
            const bool more = it + 1 < G;   // sub-group it + 1: rows -> X, conv1, conv2
            if (more) {
                if (!(PPG_PIPE_ABLATE & 2)) stage(it + 1);
                b_target += 4;
                pipe_arrive(ctr, lane);
#if PPG_PIPE_FETCH_EARLY
                if constexpr (CH) fetch(it + 2 < G ? it + 2 : G - 1);
#endif
            }
            PPG_DP(11);
            if (!(PPG_PIPE_ABLATE & 1) && it >= 1 && it - 1 < G) {   // head of sub-group it - 1: this wavefront's k-steps
                const int g = it - 1, ns = group_ns(g);
                const __bf16 *fb = img + __mul24(colh < ns ? colh : 0, sample_stride) + ((g & 1) ? K.pipe_f1 : K.off_f);
                f32x4_t hacc, hacc1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { hacc[i] = 0.0f; hacc1[i] = 0.0f; }
                bf16x8 fv[HF];
#pragma unroll
#if PPG_PIPE_HOFF
                for (int i = 0; i < HF; ++i) fv[i] = *(const bf16x8 *)(fb + hoff[i]);
#else
                for (int i = 0; i < HF; ++i) fv[i] = *(const bf16x8 *)(fb + f_koff(K, k_lo + i, kq));
#endif
                if (PPG_PIPE_HEAD_CHAINS == 2) {
                    // two chains (k-steps 0-8, 9-17), added at the end: the SAME partial sums as one chain of eighteen would NOT come out bit
                    // for bit -- the one-role kernels keep one chain; the logits differ in the last bits (tests: tolerance, not equality)
#pragma unroll
                    for (int i = 0; i < HF / 2; ++i) {
                        hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[i], fv[i], hacc, 0, 0, 0);
                        hacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[HF / 2 + i], fv[HF / 2 + i], hacc1, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) hacc[i] += hacc1[i];
                } else {
#pragma unroll
                for (int i = 0; i < HF; ++i) hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[i], fv[i], hacc, 0, 0, 0);
                }
                float *wr = red + (g & 1) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) wr[(bw * 16 + 4 * kq + i) * 16 + colh] = hacc[i];
            }
            PPG_DP(4);
            if (more) {
                const int g = it + 1, ns = group_ns(g), xo = (g & 1) ? K.pipe_x1 : 0;
                pipe_wait(ctr, b_target);
                PPG_DP(6);
                if (!(PPG_PIPE_ABLATE & 4)) conv1x<CB1>(K, w1x, img, xo, ns, bw, lane, dummy);
                PPG_DP(7);
                b_target += 4;
                pipe_arrive(ctr, lane);
                if (PPG_PIPE_ABLATE & 128) { }
#if PPG_PIPE_FETCH_EARLY
                else if constexpr (CH) { }
#else
                else if constexpr (CH) fetch(g + 1 < G ? g + 1 : g);   // (unconditional, like the loads in it; the last one is not parked)
#endif
                else if (g + 1 < G) request(g + 1);
                PPG_DP(5);
                pipe_wait(ctr, b_target);
                PPG_DP(8);
#if PPG_PIPE_CONV2_PAIR
                conv2_pair(K, w2c, img, K.off_y, xo, K.cout_blocks[1], ns, bw, lane, dummy, cells);
#else
                if (!(PPG_PIPE_ABLATE & 8))
                dconv<2, 1, PPG_PIPE_B12, PPG_PIPE_SWP_B, false, PPG_PIPE_D2>(K, w2c, img, sample_stride, K.off_y, xo, K.cout_blocks[1], 0, ns, bw, 4, lane, 0, dummy, cells);
#endif
                PPG_DP(9);
                if (PPG_PIPE_ABLATE & 128) { }
                else if constexpr (CH) park(g + 1, g + 1 < G);   // (every B wavefront has staged sub-group g out of `raw`: two private barriers ago)
            }
            __syncthreads();
            PPG_DP(10);
#ifdef PPG_DIRECT_PROFILE
            dp_acc[15] += 1;
#endif
        }
    }
    };
    if (row_chunks) b_tiles(std::true_type{}); else b_tiles(std::false_type{});
    dp_dump();
}

#define PPG_POLICY_PIPE_KERNEL(name, OBS, NCH)                                                   \
    extern "C" __global__ void __launch_bounds__(512, 1) name(const PolParams K) {               \
        extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                      \
        pipe_main<OBS, NCH>((KPtr)__builtin_amdgcn_kernarg_segment_ptr(), lds);                  \
    }
PPG_POLICY_PIPE_KERNEL(ppg_policy_pipe8_f64, 0, 8)
PPG_POLICY_PIPE_KERNEL(ppg_policy_pipe8_f32, 1, 8)
PPG_POLICY_PIPE_KERNEL(ppg_policy_pipe8_bf16, 2, 8)
PPG_POLICY_PIPE_KERNEL(ppg_policy_pipe16_f64, 0, 16)
PPG_POLICY_PIPE_KERNEL(ppg_policy_pipe16_f32, 1, 16)
PPG_POLICY_PIPE_KERNEL(ppg_policy_pipe16_bf16, 2, 16)

// ---- BOTH species in ONE launch, no plan launch (round 5) ----------------------------------------------------------------------
// Rounds 2-4: a plan launch (one workgroup, 11 us), the prey's forward launch, the predators' forward launch -- the last one short
// (ten pipeline iterations per workgroup at the benchmark, i.e. a third of its time filling and draining, matrix pipes busy 35 %) and
// serialised behind the first, whose workgroups need a CU's whole LDS.  Here every workgroup computes the plan for itself (the envs' row
// counts are 8 bytes per env: fused_prefix_sums), and the launch's workgroups are DIVIDED between the species in proportion to their
// work: n_q workgroups run the prey network on all prey rows, the others the predators' network on all predator rows -- each pipeline
// fills and drains once, over six times as many iterations for the predators, and nothing waits for a launch boundary.
struct PolParams2 {
    PolParams q, p;            // prey, predators (same envs, handles and action tensors)
    int32_t iter_q, iter_p;    // cycles of one pipeline iteration (a sub-group of ST samples) of either network: the split's weights
    int32_t scratch_off;       // byte offset of the prologue's LDS scratch: behind BOTH species' fixed areas (max of their pipe_img)
    uint32_t *pre_g;           // library-owned [2][n_envs]: the prefix sums (predators, prey) for workgroups whose share is several tiles
};
typedef const __attribute__((address_space(4))) PolParams2 *K2Ptr;

// How many of the launch's G workgroups serve the prey (the others: the predators), from the species' sub-group counts.  false: no
// observation row anywhere.
__device__ __forceinline__ bool fused_split(K2Ptr K2, int G, int sgq, int sgp, int &n_q) {
    if (sgq == 0 && sgp == 0) return false;
    if (sgp == 0) n_q = G;
    else if (sgq == 0) n_q = 0;
    else if (G < 2) n_q = G;   // (one workgroup: the prey only -- never the case on a GPU)
    else {
        // the split that finishes first: a workgroup's time = (its sub-groups + the pipeline's three fill / drain iterations) x the
        // species' cycles per iteration; the real-valued optimum, then the integers around it (scalar arithmetic, every thread the same)
        const float wq = (float)sgq * (float)K2->iter_q, wp = (float)sgp * (float)K2->iter_p;
        int guess = (int)((float)G * wq / (wq + wp) + 0.5f);
        int best = 1;
        long long best_t = 0x7FFFFFFFFFFFFFFFll;
        for (int d = -3; d <= 3; ++d) {
            int n = guess + d;
            n = n < 1 ? 1 : n > G - 1 ? G - 1 : n;
            const long long tq = (long long)((sgq + n - 1) / n + 3) * K2->iter_q, tp = (long long)((sgp + (G - n) - 1) / (G - n) + 3) * K2->iter_p;
            const long long t = tq > tp ? tq : tp;
            if (t < best_t) { best_t = t; best = n; }
        }
        n_q = best;
    }
    n_q = __builtin_amdgcn_readfirstlane(n_q);
    return true;
}
// A share longer than the workgroup's sample table comes as several tiles, and the first tile's images overwrite the prefix sums in LDS:
// such a workgroup puts them into memory first (K2->pre_g: every workgroup would write the SAME words there; it reads back its own
// stores) and bisects there for its later tiles -- twelve dependent L2 reads per sample instead of LDS reads.
__device__ __forceinline__ void fused_long_share(K2Ptr K2, const uint32_t *scratch, int tid, int n_threads, int wg, int n_q, int G, int sgq, int sgp) {
    const bool is_q = wg < n_q;
    const int st = is_q ? K2->q.ST : K2->p.ST, rt = is_q ? K2->q.range_tile : K2->p.range_tile, n_w = is_q ? n_q : G - n_q;
    const int sg = is_q ? sgq : sgp;
    const int share = __builtin_amdgcn_readfirstlane(st * ((sg + n_w - 1) / n_w));
    if (share > rt) {
        const uint32_t *src = scratch + FUSED_PART_WORDS;
        for (int i = tid; i < 2 * K2->q.n_envs; i += n_threads) K2->pre_g[i] = src[i];
        __threadfence();
        __syncthreads();
    }
}

template <int OBS, int NCHQ, int NCHP>
__device__ __forceinline__ void fused_main(K2Ptr K2, unsigned char *lds) {
    const int tid = (int)threadIdx.x;
    uint32_t *scratch = (uint32_t *)(lds + K2->scratch_off);
    uint32_t n_pred, n_prey;
    fused_prefix_sums<512>(K2->q, scratch, tid, n_pred, n_prey);
    const int G = (int)gridDim.x;
    const int sgq = ((int)n_prey + K2->q.ST - 1) / K2->q.ST, sgp = ((int)n_pred + K2->p.ST - 1) / K2->p.ST;   // sub-groups of either species
    int n_q;
    if (!fused_split(K2, G, sgq, sgp, n_q)) return;
    const int wg = (int)blockIdx.x;
    fused_long_share(K2, scratch, tid, 512, wg, n_q, G, sgq, sgp);
    // (each species' parameter block through an address the optimiser cannot see through: known to be kernel arguments, the scalar loads
    //  of BOTH blocks' fields are hoisted into this function's entry and kept -- 228 spilled scalar registers, their reloads in the
    //  head's and the convolutions' loops)
    if (wg < n_q) {
        uintptr_t kp = (uintptr_t)&K2->q;
        __asm__ volatile("" : "+s"(kp));
        pipe_main<OBS, NCHQ, true>((KPtr)kp, lds, wg, n_q, (int)n_prey, scratch, K2->pre_g + K2->q.n_envs);
    } else {
        uintptr_t kp = (uintptr_t)&K2->p;
        __asm__ volatile("" : "+s"(kp));
        pipe_main<OBS, NCHP, true>((KPtr)kp, lds, wg - n_q, G - n_q, (int)n_pred, scratch, K2->pre_g);
    }
}

#define PPG_POLICY_PIPE2_KERNEL(name, OBS, NCHQ, NCHP)                                           \
    extern "C" __global__ void __launch_bounds__(512, 1) name(const PolParams2 K) {              \
        extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                      \
        fused_main<OBS, NCHQ, NCHP>((K2Ptr)__builtin_amdgcn_kernarg_segment_ptr(), lds);         \
    }
// name: ppg_policy_pipe2_<prey channel slots>_<predator channel slots>_<row dtype>
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_16_8_bf16, 2, 16, 8)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_16_8_f32, 1, 16, 8)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_16_8_f64, 0, 16, 8)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_8_8_bf16, 2, 8, 8)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_8_8_f32, 1, 8, 8)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_8_8_f64, 0, 8, 8)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_16_16_bf16, 2, 16, 16)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_16_16_f32, 1, 16, 16)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_16_16_f64, 0, 16, 16)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_8_16_bf16, 2, 8, 16)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_8_16_f32, 1, 8, 16)
PPG_POLICY_PIPE2_KERNEL(ppg_policy_pipe2_8_16_f64, 0, 8, 16)

}  // namespace ppgpol
