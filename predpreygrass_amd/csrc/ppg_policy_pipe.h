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

namespace ppgpol {

// role B's private barrier: the `target`-th arrival releases the four wavefronts (monotonic counter, never reset)
__device__ __forceinline__ void pipe_bsync(uint32_t *ctr, uint32_t target, int lane) {
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wavefront's LDS writes have landed (LDS serves a wavefront in order)
    if (lane == 0) (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        __builtin_amdgcn_s_sleep(1);
    __asm__ volatile("" ::: "memory");
}

template <int OBS, int NCH>
__device__ __forceinline__ void pipe_main(KPtr Kp, unsigned char *lds) {
    constexpr int CB1 = NCH > 8 ? 2 : 1, HF = 18;
    const auto &K = *Kp;
    const int tid = (int)threadIdx.x, lane = tid & 63, btid = tid & 255;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_b = wave >= 4;
    const int bw = wave & 3;
    unsigned long long *tab = (unsigned long long *)lds;                    // [range_tile][2]: observation row, action slot
    float *red = (float *)(lds + K.pipe_red);                               // [2][wavefront][16 actions][16 samples]
    uint32_t *ctr = (uint32_t *)(lds + K.pipe_red + 8192);
    __bf16 *img = (__bf16 *)(lds + K.pipe_img);
    const int dummy = -4096 + 8 * tid;   // (element index from img: this thread's 16 bytes of the 8 KB in front of the images; dconv)
    const int sample_stride = K.sample_stride;
    const int N = (int)K.plan[0], share = (int)K.plan[1], tpw = (int)K.plan[2];
    const int begin = (int)blockIdx.x * share, end = (begin + share) < N ? (begin + share) : N;
    if (begin >= end) return;
    const int n_slots = (int)gridDim.x * tpw;
    const int apad = 16;
    float *lgs = K.lgs + (size_t)blockIdx.x * K.range_tile * apad;
    for (int i = tid; i < (K.ST * sample_stride) / 8 + 18 * 4; i += 512) ((bf16x8 *)img)[i] = zero8();
    if (tid == 0) *ctr = 0u;
    typedef typename ObsRaw<OBS, NCH>::type raw_t;
    typedef typename ObsRaw<OBS, NCH>::elem elem_t;
    // -DPPG_DIRECT_PROFILE: cycles per phase and wavefront -> K.xg [workgroup][8][16].  Role A: 0 table, 1 conv3, 2 waiting at the barrier,
    // 14 actions.  Role B: 0 table, 3 logits, 4 head, 5 staging + requests, 6 first private barrier, 7 conv1, 8 second private barrier,
    // 9 conv2, 10 waiting at the barrier, 14 actions.  15 = iterations
#ifdef PPG_DIRECT_PROFILE
    long long dp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dp_prev = (long long)clock64();
    auto dp_dump = [&] {
        if (K.xg && lane == 0)
            for (int i = 0; i < 16; ++i) ((unsigned long long *)K.xg)[((size_t)blockIdx.x * 8 + wave) * 16 + i] = (unsigned long long)dp_acc[i];
    };
#else
    auto dp_dump = [] {};
#endif

    const int kq = lane >> 4, colh = lane & 15;
    const int per = (K.kflat_steps + 3) >> 2, k_lo = bw * per;
    const int smp = btid >> 4, a16 = btid & 15;

    // the tile's sample table: sample -> (observation row, action slot), bisection over the envs' prefix sums
    auto build_table = [&](int tile, int n0, int nt_samples) {
        __syncthreads();   // the previous tile's last readers of the table are done (and the zero fill has landed)
        for (int i = tid; i < nt_samples; i += 512) {
            const uint32_t n = (uint32_t)(n0 + i);
            int lo = (int)K.tile_env[tile], hi = tile + 1 < n_slots ? (int)K.tile_env[tile + 1] : K.n_envs - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (K.plan[PLAN_HDR + mid] <= n) lo = mid; else hi = mid - 1;
            }
            const int e = lo, row = (int)(n - K.plan[PLAN_HDR + e]);
            const int k = handle_of(K.env_base, K.n_handles, e);
            const int b = e - K.env_base[k];
            tab[2 * i] = (unsigned long long)(uintptr_t)(K.obs[k] + ((size_t)b * K.cap + row) * (size_t)K.obs_elems * (OBS == 2 ? 2 : OBS == 1 ? 4 : 8));
            tab[2 * i + 1] = (unsigned long long)(uintptr_t)(K.actions[k] + (size_t)b * K.S + K.slot0 + row);
        }
        __syncthreads();
    };
    // the tile's actions: one lane per sample (argmax, or Gumbel-max with Philox keyed by (seed, env, row))
    auto select_actions = [&](int nt_samples) {
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's scratch rows have been written ...
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ... and stale L1 lines of the previous tile's rows are dropped
        for (int i = tid; i < nt_samples; i += 512) {
            const float *row = lgs + i * apad;
            int8_t *dst = (int8_t *)(uintptr_t)tab[2 * i + 1];
            uint32_t c_env = 0, c_slot = 0;   // Philox counter of this agent = (global env index, row slot), as in phase_head
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
            for (int a4 = 0; a4 < K.n_actions; a4 += 4) {
                const f32x4_t q = *(const GLOBAL_AS f32x4_t *)(row + a4);
                if (K.sample) philox(c_env, c_slot, (uint32_t)(a4 >> 2), 0x504F4C31u, K.seed_lo, K.seed_hi, rnd);
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    if (a4 + i4 >= K.n_actions) continue;
                    float v = q[i4];
                    if (K.sample) {   // Gumbel-max: argmax(logit - log(-log u)) ~ softmax(logits)
                        const float u = (float)(rnd[i4] >> 9) * (1.0f / 8388608.0f) + (1.0f / 16777216.0f);   // 23 bits: 2^-24 <= u < 1, exactly
                        v -= __logf(-__logf(u));
                    }
                    if (v > bestv) { bestv = v; best = a4 + i4; }
                }
            }
            *dst = (int8_t)best;
        }
    };

    // The two roles are two separate loops (not two branches inside one): inside one loop the register allocator would have to keep BOTH
    // roles' weights alive.  Both execute the same sequence of workgroup barriers.
    if (!role_b) {
        // ================= role A: conv3 =================
        ConvW<4, 2> w3c;
        w3c.load(K, K.wc3, lane, 0);
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w3c.landed();
        for (int j = 0; j < tpw; ++j) {
            const int tile = (int)blockIdx.x * tpw + j;
            const int n0 = begin + j * K.range_tile;
            if (n0 >= end) break;
            const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
            const int G = (nt_samples + K.ST - 1) / K.ST;
            build_table(tile, n0, nt_samples);
            PPG_DP(0);
            for (int it = -1; it <= G + 1; ++it) {
                if (it >= 0 && it < G) {
                    const int left = nt_samples - it * K.ST, ns = left < K.ST ? left : K.ST;
                    dconv<4, 2, PPG_DIRECT_B3, false>(K, w3c, img, sample_stride, (it & 1) ? K.pipe_x1 : 0, (it & 1) ? K.pipe_f1 : K.off_f,
                                                      K.cout_blocks[2], K.flat_c, ns, wave, 4, lane, 0, dummy);
                }
                PPG_DP(1);
                __syncthreads();
                PPG_DP(2);
            }
            select_actions(nt_samples);
            PPG_DP(14);
        }
        dp_dump();
        return;
    }
    // ================= role B: rows -> X, conv1, conv2; head; logits =================
    // Role B is the longer chain, and its MFMAs are few and dependent: with equal priorities the SIMD serves role A's dense MFMA stream
    // first and the head's 18 MFMAs take as long as the whole of conv3 (profiles/r04) -- B goes first whenever it has something to issue.
    __builtin_amdgcn_s_setprio(3);
    ConvW<CB1, 1> w1c;
    ConvW<2, 1> w2c;
    bf16x8 hf[HF];
    w1c.load(K, K.wc1, lane);
    w2c.load(K, K.wc2, lane);
#pragma unroll
    for (int i = 0; i < HF; ++i) hf[i] = ((const GLOBAL_AS bf16x8 *)K.whw)[((size_t)bw * HF + i) * 64 + lane];
    const float bias_r = (a16 < K.n_actions) ? K.bh[a16] : 0.0f;
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    w1c.landed();
    w2c.landed();
#pragma unroll
    for (int i = 0; i < HF; ++i) {
        u32x4_t v = __builtin_bit_cast(u32x4_t, hf[i]);
        __asm__ volatile("" : "+v"(v));
        hf[i] = __builtin_bit_cast(bf16x8, v);
    }
    uint32_t b_target = 0;
    for (int j = 0; j < tpw; ++j) {
        const int tile = (int)blockIdx.x * tpw + j;
        const int n0 = begin + j * K.range_tile;
        if (n0 >= end) break;
        const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
        const int G = (nt_samples + K.ST - 1) / K.ST;
        build_table(tile, n0, nt_samples);
        auto group_ns = [&](int g) { const int left = nt_samples - g * K.ST; return left < K.ST ? left : K.ST; };
        raw_t pre[NCH];
        auto request = [&](int g) {   // (a B thread stages one position: ST * P <= 256, ppg_policy_create_spec)
            const int ns = group_ns(g);
#pragma unroll
            for (int c = 0; c < NCH; ++c) pre[c] = (raw_t)0;
            if (btid < ns * K.P) {
                const int s = div_small(btid, K.magic_P), p = btid - __mul24(s, K.P);
                const GLOBAL_AS elem_t *src = (const GLOBAL_AS elem_t *)(uintptr_t)tab[2 * (g * K.ST + s)] + p * K.p_stride;
#pragma unroll
                for (int c = 0; c < NCH; ++c) if (c < K.cin) pre[c] = (raw_t)src[c * K.c_stride];
            }
        };
        auto stage = [&](int g) {
            const int ns = group_ns(g);
            if (btid < ns * K.P) {
                const int s = div_small(btid, K.magic_P), p = btid - __mul24(s, K.P);
                const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
#pragma unroll
                for (int cb = 0; cb < CB1; ++cb) {
                    bf16x8 v = zero8();
#pragma unroll
                    for (int c = 0; c < (NCH < 8 ? NCH : 8); ++c) v[c] = ObsRaw<OBS, NCH>::to_bf16(pre[8 * cb + c]);
                    *(bf16x8 *)(img + __mul24(s, sample_stride) + ((g & 1) ? K.pipe_x1 : 0) + (cb * K.Wp2 + __mul24(y + 1, K.Wp) + (x + 1)) * 8) = v;
                }
            }
        };
        request(0);
        PPG_DP(0);
        for (int it = -1; it <= G + 1; ++it) {
            if (it >= 2 && it - 2 < G) {   // logits of sub-group it - 2: bias + the four partial sums in wavefront order
                const int g = it - 2, ns = group_ns(g);
                if (smp < ns) {
                    const float *rd = red + (g & 1) * 1024;
                    float v = bias_r;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += rd[(w * 16 + a16) * 16 + smp];
                    const int s_local = g * K.ST + smp;
                    lgs[s_local * apad + a16] = v;
                    if (K.logits && a16 < K.n_actions) K.logits[(size_t)(n0 + s_local) * K.n_actions + a16] = v;
                }
            }
            PPG_DP(3);
            if (it >= 1 && it - 1 < G) {   // head of sub-group it - 1: this wavefront's k-steps
                const int g = it - 1, ns = group_ns(g);
                const __bf16 *fb = img + __mul24(colh < ns ? colh : 0, sample_stride) + ((g & 1) ? K.pipe_f1 : K.off_f) + 8 * kq;
                f32x4_t hacc;
#pragma unroll
                for (int i = 0; i < 4; ++i) hacc[i] = 0.0f;
                bf16x8 fv[HF];
#pragma unroll
                for (int i = 0; i < HF; ++i) fv[i] = *(const bf16x8 *)(fb + 32 * (k_lo + i));
#pragma unroll
                for (int i = 0; i < HF; ++i) hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[i], fv[i], hacc, 0, 0, 0);
                float *wr = red + (g & 1) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) wr[(bw * 16 + 4 * kq + i) * 16 + colh] = hacc[i];
            }
            PPG_DP(4);
            if (it + 1 < G) {   // sub-group it + 1: rows -> X, conv1, conv2
                const int g = it + 1, ns = group_ns(g), xo = (g & 1) ? K.pipe_x1 : 0;
                stage(g);
                if (g + 1 < G) request(g + 1);
                PPG_DP(5);
                b_target += 4;
                pipe_bsync(ctr, b_target, lane);
                PPG_DP(6);
                dconv<CB1, 1, PPG_DIRECT_B12>(K, w1c, img, sample_stride, xo, K.off_y, K.cout_blocks[0], 0, ns, bw, 4, lane, 0, dummy);
                PPG_DP(7);
                b_target += 4;
                pipe_bsync(ctr, b_target, lane);
                PPG_DP(8);
                dconv<2, 1, PPG_DIRECT_B12>(K, w2c, img, sample_stride, K.off_y, xo, K.cout_blocks[1], 0, ns, bw, 4, lane, 0, dummy);
                PPG_DP(9);
            }
            __syncthreads();
            PPG_DP(10);
#ifdef PPG_DIRECT_PROFILE
            dp_acc[15] += 1;
#endif
        }
        select_actions(nt_samples);
        PPG_DP(14);
    }
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

}  // namespace ppgpol
