// ppg_policy_pipe4.h -- the two-role pipeline (ppg_policy_pipe.h) cut into FOUR roles: sixteen wavefronts per workgroup, one workgroup
// per CU, four wavefronts per SIMD that run four different programs (round 5).
//
// What the two-role kernels leave on the table (profiles/r05/h_*, f_*): a SIMD holds two wavefronts, each a dependent chain of LDS round
// trips and MFMAs; role B's chain (head, staging, conv1, conv2: 6800 cycles per sub-group) is the longer one, role A (conv3) waits a fifth
// of every iteration for it at the barrier, and the matrix pipe is busy half of the cycles.  Deeper prefetch, interleaved MFMA chains, more
// registers for addresses: all measured, none helps -- two chains per SIMD are too few to cover each other's stalls.  The registers are why
// there are two: conv3's weights alone are 152 per lane.  Split by OUTPUT CHANNEL HALVES they are 76, and role B splits into two chains
// that never exchange data inside an iteration:
//   wavefronts  0- 3  role A   conv3, output channels  0-31:  X -> F         + logits and actions of the sub-group two iterations back
//   wavefronts  4- 7  role B1  staging, conv1, conv2 (the private barriers stay among these four)
//   wavefronts  8-11  role C   conv3, output channels 32-63:  X -> F         + the Gumbel noise one iteration ahead
//   wavefronts 12-15  role B2  the head's partial sums; the observation rows' fetch from memory and their parking in LDS
// 128 registers each (1024 threads), four chains per SIMD.  Every output is computed by the same instructions in the same order as in
// the two-role kernels (a conv3 output channel's k-steps, the head's eighteen MFMAs per wavefront): logits and actions are bit-identical.
// The price: conv3's B fragments are read from LDS by both halves (conv3's LDS reads double).  Same LDS layout, same slot table, same
// sub-group size as the two-role kernels (ppg_pipe_layout); the fused launch's form only (both species, plan in the prologue).
//
// MEASURED, NOT KEPT (profiles/r05/i_policy_four_role_pipeline_ab_not_kept.txt): bit-identical and 18 % slower.  That price is the whole
// story: 304 KB of conv3 fragments per sub-group for 1216 MFMA cycles per wavefront is the LDS's entire bandwidth -- a conv3 half takes
// 6000-6600 cycles where the whole conv3 took 4100.  Built only with -DPPG_WITH_PIPE4 (the test for it skips otherwise).
#pragma once

namespace ppgpol {

#ifndef PPG_PIPE4_PRIO_B1
#define PPG_PIPE4_PRIO_B1 3   // s_setprio of the roles (A / C: 0)
#endif
#ifndef PPG_PIPE4_PRIO_B2
#define PPG_PIPE4_PRIO_B2 2
#endif
#ifndef PPG_PIPE4_HEAD_BATCH
#define PPG_PIPE4_HEAD_BATCH 3   // operands of the head read per batch (eighteen at once do not fit role B2's 128 registers)
#endif

template <int OBS, int NCH>
__device__ __forceinline__ void pipe4_main(KPtr Kp, unsigned char *lds, int wg, int n_wgs, int N_, uint32_t *scratch, const uint32_t *pre_g) {
    constexpr int CB1 = NCH > 8 ? 2 : 1, HF = 18, NT = 1024;
    const auto &K = *Kp;
    const uint32_t *pre = scratch + FUSED_PART_WORDS + (K.species ? K.n_envs : 0);
    const int tid = (int)threadIdx.x, lane = tid & 63, btid = tid & 255;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;   // 0 A, 1 B1, 2 C, 3 B2
    const int bw = wave & 3;
    unsigned long long *tab = (unsigned long long *)lds;                    // [range_tile][2]: observation row; global env index | row << 32
    float *red = (float *)(lds + K.pipe_red);                               // [2][wavefront][16 actions][16 samples]
    uint32_t *ctr = (uint32_t *)(lds + K.pipe_red + 8192);
    float *noise = (float *)(lds + K.pipe_red + 8192 + 64);                 // [2][16 samples][16 actions]: Gumbel noise of two sub-groups
    __bf16 *img = (__bf16 *)(lds + K.pipe_img);
    const int dummy = -512 + 8 * lane;
    const int sample_stride = K.sample_stride;
    const int N = N_;
    const int sg = (N + K.ST - 1) / K.ST;
    const int share = __builtin_amdgcn_readfirstlane(K.ST * ((sg + n_wgs - 1) / n_wgs));
    const int tpw = __builtin_amdgcn_readfirstlane(share ? (share + K.range_tile - 1) / K.range_tile : 1);
    const int begin = wg * share, end = (begin + share) < N ? (begin + share) : N;
    if (begin >= end) return;
    typedef typename ObsRaw<OBS, NCH>::type raw_t;
    typedef typename ObsRaw<OBS, NCH>::elem elem_t;
    // -DPPG_DIRECT_PROFILE: cycles per phase and wavefront -> K.xg [workgroup][16][16] (the indices of ppg_policy_pipe.h; 15 = iterations)
#ifdef PPG_DIRECT_PROFILE
    long long dp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dp_prev = (long long)clock64();
    auto dp_dump = [&] {
        if (K.xg && lane == 0)
            for (int i = 0; i < 16; ++i)
                ((unsigned long long *)K.xg)[((size_t)blockIdx.x * 16 + wave) * 16 + i] = i == 13 ? (unsigned long long)(K.species + 1) : (unsigned long long)dp_acc[i];
    };
#else
    auto dp_dump = [] {};
#endif
    const int kq = lane >> 4, colh = lane & 15;
    const int per = (K.kflat_steps + 3) >> 2, k_lo = bw * per;
    const int smp = btid >> 4, a16 = btid & 15;
    const bool row_chunks = OBS == 2 && K.pipe_ni > 0;

    // the tile's sample table (pipe_main's, FUSED form) and the zero fill of the images behind it
    auto build_table = [&](int n0, int nt_samples) {
        __syncthreads();
        for (int i = tid; i < nt_samples; i += NT) {
            const uint32_t n = (uint32_t)(n0 + i);
            int lo = 0, hi = K.n_envs - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                const uint32_t at = n0 == begin ? pre[mid] : __builtin_nontemporal_load(&pre_g[mid]);
                if (at <= n) lo = mid; else hi = mid - 1;
            }
            const int e = lo;
            const int row = (int)(n - (n0 == begin ? pre[e] : __builtin_nontemporal_load(&pre_g[e])));
            const unsigned char *base;
            const int b = pipe_pick_handle(K, K.obs, e, base);
            tab[2 * i] = (unsigned long long)(uintptr_t)(base + ((size_t)b * K.cap + row) * (size_t)K.obs_elems * (OBS == 2 ? 2 : OBS == 1 ? 4 : 8));
            tab[2 * i + 1] = (unsigned long long)(uint32_t)e | ((unsigned long long)(uint32_t)row << 32);
        }
        __syncthreads();
        for (int i = tid; i < (K.ST * sample_stride) / 8 + 18 * 4; i += NT) ((bf16x8 *)img)[i] = zero8();
        if (tid == 0 && n0 == begin) *ctr = 0u;   // (role B1's counter runs on from tile to tile)
        __syncthreads();
    };
    auto row_of = [&](int s_tile) -> const GLOBAL_AS unsigned char * { return (const GLOBAL_AS unsigned char *)(uintptr_t)tab[2 * s_tile]; };

    // Every role runs the same tile loop and the same sequence of workgroup barriers: build_table (three), one behind the first rows'
    // parking, one per iteration.
    if (role == 0 || role == 2) {
        // ================= roles A, C: conv3, one half of the output channels each =================
        const int half = role >> 1;
        ConvW<4, 1> w3h;
        w3h.load(K, K.wc3, lane, half);
        const float bias_r = (a16 < K.n_actions) ? K.bh[a16] : 0.0f;
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w3h.landed();
        int cells[6];
        dconv_cells(K, sample_stride, bw, 4, lane, cells);
        for (int j = 0; j < tpw; ++j) {
            const int n0 = begin + j * K.range_tile;
            if (n0 >= end) break;
            const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
            const int G = (nt_samples + K.ST - 1) / K.ST;
            build_table(n0, nt_samples);
            __syncthreads();   // (role B2 has parked the first sub-group's rows)
            PPG_DP(0);
            for (int it = -1; it <= G + 1; ++it) {
                // role A, wavefronts 0-1: logits and action of (sample smp, action a16) of sub-group it - 2, behind conv3 (its LDS reads too:
                // held across conv3 -- the two-role kernels do that -- they do not fit this role's 128 registers)
                const bool do_act = role == 0 && it >= 2 && 4 * bw < K.ST && 4 * bw < nt_samples - (it - 2) * K.ST;
                PPG_DP(3);
                if (it >= 0 && it < G) {
                    const int left = nt_samples - it * K.ST, ns = left < K.ST ? left : K.ST;
                    if (K.flat_c == 64)
                        dconv<4, 1, PPG_PIPE_B3, false, true>(K, w3h, img, sample_stride, (it & 1) ? K.pipe_x1 : 0, (it & 1) ? K.pipe_f1 : K.off_f,
                                                              K.cout_blocks[2], 64, ns, bw, 4, lane, half, dummy, cells);
                    else
                        dconv<4, 1, PPG_PIPE_B3, false, false>(K, w3h, img, sample_stride, (it & 1) ? K.pipe_x1 : 0, (it & 1) ? K.pipe_f1 : K.off_f,
                                                               K.cout_blocks[2], K.flat_c, ns, bw, 4, lane, half, dummy, cells);
                }
                PPG_DP(1);
                if (do_act) {
                    const int g = it - 2, left = nt_samples - g * K.ST, ns = left < K.ST ? left : K.ST;
                    const float *rd = red + (g & 1) * 1024;
                    float v = bias_r;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += rd[(w * 16 + a16) * 16 + smp];
                    const int s_local = g * K.ST + (smp < ns ? smp : 0);
                    const unsigned long long er = tab[2 * s_local + 1];
                    const uint32_t e = (uint32_t)er, row = (uint32_t)(er >> 32);
                    if (K.logits && smp < ns && a16 < K.n_actions) K.logits[(size_t)(n0 + s_local) * K.n_actions + a16] = v;
                    if (K.sample) v += noise[(g & 1) * 256 + btid];
                    if (a16 >= K.n_actions) v = -INFINITY;
                    int best = a16;
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
                // role C: the Gumbel noise of sub-group it - 1 (needed in the next iteration); wavefront bw serves the samples 4 bw .. 4 bw + 3
                if (role == 2 && K.sample && it >= 1 && it - 1 < G) {
                    const int g = it - 1, left = nt_samples - g * K.ST, ns = left < K.ST ? left : K.ST;
                    const int s0 = 4 * bw;
                    if (s0 < ns) {
                        const int sm = s0 + (lane >> 4);
                        const unsigned long long er = tab[2 * (g * K.ST + (sm < ns ? sm : 0)) + 1];
                        uint32_t rnd[4];
                        philox((uint32_t)er, (uint32_t)K.slot0 + (uint32_t)(er >> 32), (uint32_t)(a16 >> 2), 0x504F4C31u, K.seed_lo, K.seed_hi, rnd);
                        const uint32_t r = (a16 & 3) == 0 ? rnd[0] : (a16 & 3) == 1 ? rnd[1] : (a16 & 3) == 2 ? rnd[2] : rnd[3];
                        const float u = (float)(r >> 9) * (1.0f / 8388608.0f) + (1.0f / 16777216.0f);   // 23 bits: 2^-24 <= u < 1, exactly
                        noise[(g & 1) * 256 + sm * 16 + a16] = -__logf(-__logf(u));
                    }
                }
                PPG_DP(5);
                __syncthreads();
                PPG_DP(2);
#ifdef PPG_DIRECT_PROFILE
                dp_acc[15] += 1;
#endif
            }
        }
        dp_dump();
        return;
    }
    // the position a B1 thread stages (slot table) and the row chunk a B2 thread fetches: the same in every sub-group
    unsigned char *raw = lds + K.pipe_raw;
    const int cpr = K.obs_elems >> 2;   // 8-byte chunks per bfloat16 row
    if (role == 1) {
        // ================= role B1: rows -> X, conv1, conv2 =================
        __builtin_amdgcn_s_setprio(PPG_PIPE4_PRIO_B1);
        Conv1X w1x;
        ConvW<2, 1> w2c;
        w1x.load(K, lane, bw);
        w2c.load(K, K.wc2, lane);
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w1x.landed();
        w2c.landed();
        uint32_t b_target = 0;
        int cells[6];
        dconv_cells(K, sample_stride, bw, 4, lane, cells);
        const uint32_t st_e = btid < 32 * K.slot_tiles ? (uint32_t)K.slot_tab[btid] : 0xFFFFu;
        const int st_sv = (int)(st_e >> 8), st_s = st_sv == 255 ? 0 : st_sv, st_p = st_sv == 255 ? 0 : (int)(st_e & 255u);
        const int st_y = div_small(st_p, K.magic_R), st_x = st_p - __mul24(st_y, K.IW);
        const int st_img = __mul24(st_s, sample_stride) + (__mul24(st_y + 1, K.Wp) + (st_x + 1)) * 8;
        const int st_raw = __mul24(st_s, K.obs_elems) + st_p * K.p_stride;
        auto b1_tiles = [&](auto ch_tag) {
        constexpr bool CH = decltype(ch_tag)::value;
        for (int j = 0; j < tpw; ++j) {
            const int n0 = begin + j * K.range_tile;
            if (n0 >= end) break;
            const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
            const int G = (nt_samples + K.ST - 1) / K.ST;
            build_table(n0, nt_samples);
            auto group_ns = [&](int g) { const int left = nt_samples - g * K.ST; return left < K.ST ? left : K.ST; };
            raw_t pre[NCH > 8 ? 11 : NCH];   // (channels 0-8, then the ninth channel of the two neighbours)
            auto request = [&](int g) {   // rows that do not come as chunks: this thread's position straight from memory, one sub-group ahead
                const int ns = group_ns(g);
#pragma unroll
                for (int c = 0; c < (NCH > 8 ? 11 : NCH); ++c) pre[c] = (raw_t)0;
                if (st_sv < ns) {
                    const GLOBAL_AS elem_t *src = (const GLOBAL_AS elem_t *)row_of(g * K.ST + st_s) + st_p * K.p_stride;
#pragma unroll
                    for (int c = 0; c < (NCH < 9 ? NCH : 9); ++c) if (c < K.cin) pre[c] = (raw_t)src[c * K.c_stride];
                    if constexpr (CB1 > 1) {
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
                    {
                        bf16x8 v = zero8();
#pragma unroll
                        for (int c = 0; c < (NCH < 8 ? NCH : 8); ++c) v[c] = ObsRaw<OBS, NCH>::to_bf16(pre[c]);
                        *(bf16x8 *)(img + st_img + ((g & 1) ? K.pipe_x1 : 0)) = v;
                    }
                    if constexpr (CB1 > 1) {
                        bf16x8 v = zero8();
                        const __bf16 z = (__bf16)0.0f;
                        v[0] = st_x > 0 ? ObsRaw<OBS, NCH>::to_bf16(pre[9]) : z;
                        v[1] = ObsRaw<OBS, NCH>::to_bf16(pre[8]);
                        v[2] = st_x < K.IW - 1 ? ObsRaw<OBS, NCH>::to_bf16(pre[10]) : z;
                        *(bf16x8 *)(img + st_img + ((g & 1) ? K.pipe_x1 : 0) + K.Wp2 * 8) = v;
                    }
                }
            };
            if constexpr (!CH) request(0);
            __syncthreads();   // (role B2 has parked the first sub-group's rows)
            PPG_DP(0);
            for (int it = -1; it <= G + 1; ++it) {
                if (it + 1 < G) {   // sub-group it + 1: rows -> X, conv1 -> Y, conv2 -> X
                    const int g = it + 1, ns = group_ns(g), xo = (g & 1) ? K.pipe_x1 : 0;
                    stage(g);
                    b_target += 4;
                    pipe_bsync(ctr, b_target, lane);
                    PPG_DP(11);
                    conv1x<CB1, true>(K, w1x, img, xo, ns, bw, lane, dummy);
                    PPG_DP(7);
                    b_target += 4;
                    pipe_arrive(ctr, lane);
                    if constexpr (!CH) { if (g + 1 < G) request(g + 1); }
                    PPG_DP(5);
                    pipe_wait(ctr, b_target);
                    PPG_DP(8);
                    dconv<2, 1, PPG_PIPE_B12, false>(K, w2c, img, sample_stride, K.off_y, xo, K.cout_blocks[1], 0, ns, bw, 4, lane, 0, dummy, cells);
                    PPG_DP(9);
                }
                __syncthreads();
                PPG_DP(10);
#ifdef PPG_DIRECT_PROFILE
                dp_acc[15] += 1;
#endif
            }
        }
        };
        if (row_chunks) b1_tiles(std::true_type{}); else b1_tiles(std::false_type{});
        dp_dump();
        return;
    }
    // ================= role B2: the head's partial sums; row chunks: memory -> registers -> `raw` =================
    __builtin_amdgcn_s_setprio(PPG_PIPE4_PRIO_B2);
    bf16x8 hf[HF];
#pragma unroll
    for (int i = 0; i < HF; ++i) hf[i] = ((const GLOBAL_AS bf16x8 *)K.whw)[((size_t)bw * HF + i) * 64 + lane];
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < HF; ++i) {
        u32x4_t v = __builtin_bit_cast(u32x4_t, hf[i]);
        __asm__ volatile("" : "+v"(v));
        hf[i] = __builtin_bit_cast(bf16x8, v);
    }
    const int ch_q = row_chunks ? (int)__umulhi((uint32_t)btid, K.pipe_magic) : 0, ch_w = btid - ch_q * cpr;
    uint32_t b1_target = 0;   // role B1's counter as THIS role follows it: + 4 when the four B1 wavefronts have staged, + 4 behind conv1
    for (int j = 0; j < tpw; ++j) {
        const int n0 = begin + j * K.range_tile;
        if (n0 >= end) break;
        const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
        const int G = (nt_samples + K.ST - 1) / K.ST;
        build_table(n0, nt_samples);
        auto group_ns = [&](int g) { const int left = nt_samples - g * K.ST; return left < K.ST ? left : K.ST; };
        u32x2_t chunk[PIPE_CHUNKS];
        auto fetch = [&](int g) {   // (unconditional loads of clamped samples: pipe_main)
            const int last = group_ns(g) - 1;
#pragma unroll
            for (int k = 0; k < PIPE_CHUNKS; ++k) {
                const int s0 = ch_q + K.pipe_slots * k, s = s0 < last ? s0 : last;
                chunk[k] = *(const GLOBAL_AS u32x2_t *)(row_of(g * K.ST + s) + 8 * ch_w);
            }
        };
        auto park = [&](int g, bool valid) {
            const int ns = valid && ch_q < K.pipe_slots ? group_ns(g) : 0;
#pragma unroll
            for (int k = 0; k < PIPE_CHUNKS; ++k) {
                const int s = ch_q + K.pipe_slots * k;
                if (s < ns) ((u32x2_t *)raw)[s * cpr + ch_w] = chunk[k];
            }
        };
        if (row_chunks) { fetch(0); park(0, true); }
        __syncthreads();   // the first sub-group's rows are in `raw` before role B1 stages a position
        PPG_DP(0);
        for (int it = -1; it <= G + 1; ++it) {
            const bool more = it + 1 < G;
            if (row_chunks && more) fetch(it + 2 < G ? it + 2 : it + 1);   // (unconditional; the last one is not parked)
            PPG_DP(5);
            if (it >= 1 && it - 1 < G) {   // head of sub-group it - 1: this wavefront's k-steps, the operands in batches
                const int g = it - 1, ns = group_ns(g);
                const __bf16 *fb = img + __mul24(colh < ns ? colh : 0, sample_stride) + ((g & 1) ? K.pipe_f1 : K.off_f);
                f32x4_t hacc;
#pragma unroll
                for (int i = 0; i < 4; ++i) hacc[i] = 0.0f;
                constexpr int HB = PPG_PIPE4_HEAD_BATCH;
#pragma unroll
                for (int i0 = 0; i0 < HF; i0 += HB) {
                    bf16x8 fv[HB];
#pragma unroll
                    for (int i = 0; i < HB; ++i) if (i0 + i < HF) fv[i] = *(const bf16x8 *)(fb + f_koff(K, k_lo + i0 + i, kq));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < HB; ++i) if (i0 + i < HF) hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[i0 + i], fv[i], hacc, 0, 0, 0);
                }
                float *wr = red + (g & 1) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) wr[(bw * 16 + 4 * kq + i) * 16 + colh] = hacc[i];
            }
            PPG_DP(4);
            if (more) {   // role B1 stages sub-group it + 1 out of `raw` in this iteration: the next rows go in behind that
                b1_target += 4;
                if (row_chunks) {
                    pipe_wait(ctr, b1_target);
                    park(it + 2, it + 2 < G);
                }
                b1_target += 4;
            }
            PPG_DP(6);
            __syncthreads();
            PPG_DP(10);
#ifdef PPG_DIRECT_PROFILE
            dp_acc[15] += 1;
#endif
        }
    }
    dp_dump();
}

template <int OBS, int NCHQ, int NCHP>
__device__ __forceinline__ void fused4_main(K2Ptr K2, unsigned char *lds) {
    const int tid = (int)threadIdx.x;
    uint32_t *scratch = (uint32_t *)(lds + K2->scratch_off);
    uint32_t n_pred, n_prey;
    fused_prefix_sums<1024>(K2->q, scratch, tid, n_pred, n_prey);
    int n_q;
    const int G = (int)gridDim.x;
    const int sgq = ((int)n_prey + K2->q.ST - 1) / K2->q.ST, sgp = ((int)n_pred + K2->p.ST - 1) / K2->p.ST;
    if (!fused_split(K2, G, sgq, sgp, n_q)) return;
    const int wg = (int)blockIdx.x;
    fused_long_share(K2, scratch, tid, 1024, wg, n_q, G, sgq, sgp);
    if (wg < n_q) {
        uintptr_t kp = (uintptr_t)&K2->q;
        __asm__ volatile("" : "+s"(kp));
        pipe4_main<OBS, NCHQ>((KPtr)kp, lds, wg, n_q, (int)n_prey, scratch, K2->pre_g + K2->q.n_envs);
    } else {
        uintptr_t kp = (uintptr_t)&K2->p;
        __asm__ volatile("" : "+s"(kp));
        pipe4_main<OBS, NCHP>((KPtr)kp, lds, wg - n_q, G - n_q, (int)n_pred, scratch, K2->pre_g);
    }
}

#define PPG_POLICY_PIPE4_KERNEL(name, OBS, NCHQ, NCHP)                                           \
    extern "C" __global__ void __launch_bounds__(1024, 1) name(const PolParams2 K) {             \
        extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                      \
        fused4_main<OBS, NCHQ, NCHP>((K2Ptr)__builtin_amdgcn_kernarg_segment_ptr(), lds);        \
    }
// name: ppg_policy_pipe4_<prey channel slots>_<predator channel slots>_<row dtype>
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_16_8_bf16, 2, 16, 8)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_16_8_f32, 1, 16, 8)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_16_8_f64, 0, 16, 8)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_8_8_bf16, 2, 8, 8)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_8_8_f32, 1, 8, 8)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_8_8_f64, 0, 8, 8)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_16_16_bf16, 2, 16, 16)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_16_16_f32, 1, 16, 16)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_16_16_f64, 0, 16, 16)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_8_16_bf16, 2, 8, 16)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_8_16_f32, 1, 8, 16)
PPG_POLICY_PIPE4_KERNEL(ppg_policy_pipe4_8_16_f64, 0, 8, 16)

}  // namespace ppgpol
