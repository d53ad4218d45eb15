// ppg_policy_direct.h -- the policy network RLlib REALLY builds for the reference's PPO setup, evaluated entirely in LDS.
//
// tune_ppo_base_environment.py:106-141 asks for conv_filters 16/32/64 and fcnet_hiddens [256, 256]; for an image observation RLlib's
// catalog builds a CNN encoder from conv_filters, ignores fcnet_hiddens, and puts ONE Linear(flat -> n_actions) behind the flattened
// encoder output as the policy head (pi.net.mlp.0; include/ppg.h, tests/golden/rllib_checkpoint/).  So per sample:
//     (C,R,R) row, read as a C x R image with R channels -> L x [zero-pad 1, conv3x3, ReLU] -> flatten [row][column][channel]
//                                                       -> Linear -> n_actions logits
// and nothing in it needs a tile of 128 samples or a trip through HBM: no hidden fully connected layer, no weight matrix that
// does not fit next to a CU.  (included from ppg_policy.h, namespace ppgpol; shares its fragment conventions, ConvW, the plan kernel)
//
// ONE persistent launch per species.  A workgroup (4 wavefronts, two per CU) walks over the plan's tiles of up to 128 samples in
// SUB-GROUPS of ST samples (7-9 at the default windows: chosen so that ST * P positions are a multiple of four 32-position MFMA
// tiles).  LDS per sample: area X = 4 channel blocks of [padded position][8] (the input image, then conv2's 32 channels), area
// Y = 2 blocks (conv1's 16 channels), area F = the last convolution's output UNPADDED as [position][channels] -- exactly the
// flatten order of RLlib, so the head's B fragment of k-step ks is 16 contiguous bytes at F + 64 ks -- and, for networks deeper than
// three convolutions, areas D0 / D1 of 8 blocks each.  The padded pitch is W + 1 (a row's right halo cell IS the next row's
// left one): (H + 2)(W + 1) + 1 cells per block instead of (H + 2)(W + 2).
//   stage   the sub-group's observation rows (requested from HBM one sub-group earlier, converted to bf16 here) -> X
//   conv1   X -> Y          every wavefront its share of the position tiles, weights resident in registers
//   conv2   Y -> X
//   conv3   X -> F (D0)     wavefronts 0,1 / 2,3 compute output channels 0-31 / 32-63 for every other tile
//   [conv4.. D0 -> F ...    deeper layers: the weights of a layer are fetched from L2 per sub-group (148 registers per wavefront)]
//   head    logits^T = W_head x F^T on v_mfma_f32_16x16x32_bf16: M = actions (1-2 tiles of 16), N = the sub-group's samples, K = the
//           flattened features split over the four wavefronts; partial sums meet in LDS; one lane per sample adds bias, writes
//           logits, picks the action (argmax / Gumbel-max) and stores one int8.
// Four workgroup barriers per sub-group; the next sub-group's input is written into X while the head runs.
#pragma once

namespace ppgpol {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// timing-only ablation builds (tools/gpu_direct_ablate.sh; never defined in the product -- the results are then meaningless):
// 1 no head k-loop, 2 no action selection, 4 no conv3, 8 no conv1 / conv2, 16 no observation staging, 32 no LDS fragment reads in the
// convolutions, 64 no convolution epilogues
#ifndef PPG_DIRECT_ABLATE
#define PPG_DIRECT_ABLATE 0
#endif
// -DPPG_DIRECT_PROFILE (tools/gpu_direct_profile.sh): cycles per phase, summed per wavefront over the launch, into K.xg (unused by the
// direct path) as [workgroup][wavefront][16] -- 0 tile set-up, 1 conv1, 2 its barrier, 3 conv2, 4 barrier, 5 conv3 (+ deeper), 6 barrier,
// 7 partial sums to LDS, 8 barrier, 9 logits, 10 staging, 11 requests of the rows two sub-groups ahead, 12 head, 15 sub-groups
// PPG_DIRECT_W1 (default 1): the direct-head kernels run ONE workgroup per CU (one wavefront per SIMD: 512 registers --
// every weight of the network and the head's fragments resident, both of conv3's row tiles in one wavefront, 160 KB of LDS = twice the
// samples per sub-group); 0 = two workgroups per CU with 256 registers each (conv3's row tiles split over wavefront pairs, the head's
// fragments from L2 six at a time) -- the A/B build
#ifndef PPG_DIRECT_W1
#define PPG_DIRECT_W1 1
#endif
#ifndef PPG_DIRECT_B12
#define PPG_DIRECT_B12 0
#endif
#ifndef PPG_DIRECT_B3
#define PPG_DIRECT_B3 5
#endif
#ifdef PPG_DIRECT_PROFILE
#define PPG_DP(i) do { const long long now_ = (long long)clock64(); dp_acc[i] += now_ - dp_prev; dp_prev = now_; } while (0)
#else
#define PPG_DP(i) do { } while (0)
#endif

// partial logits in LDS: [wavefront][action tile (head_mt of them)][action row][sample column] floats = head_mt * 4 KB

// The words per tile dconv() takes as `pre` (the pipeline kernels): which position a lane computes in the wavefront's tiles nt_first and
// nt_first + nt_step comes from the policy's SLOT TABLE (PolParams::slot_tab, built by ppg_slot_table on the host): slot n of a
// sub-group -> (sample, position), ordered so that slot n's cell of the padded image lies in LDS bank group n mod 16 -- a tile's
// sixteen-lane read groups and eight-lane write groups then hit every bank once, whatever the image's row length (consecutive
// positions do that only inside an image row: 45 % of the pipeline kernels' LDS cycles were conflicts of tiles crossing row ends).
// pre[2 t] = sample base + 8 x padded position (the element of its cell in channel block 0 of area 0), pre[2 t + 1] = (sample base + 64 x
// position) | (position & 7), pre[4 + t] = the sample (255: the slot holds no position -- its results are never stored).
template <class KP>
__device__ __forceinline__ void dconv_cells(const KP &K, int sample_stride, int nt_first, int nt_step, int lane, int (&pre)[6]) {
    const int col = lane & 31;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t e = K.slot_tab[32 * (nt_first + t * nt_step) + col];
        const int sv = (int)(e >> 8), s = sv == 255 ? 0 : sv, p = sv == 255 ? 0 : (int)(e & 255u);
        const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
        const int sb = __mul24(s, sample_stride);
        pre[2 * t] = sb + (__mul24(y + 1, K.Wp) + (x + 1)) * 8;
        pre[2 * t + 1] = (sb + p * 64) | (p & 7);
        pre[4 + t] = sv;
    }
}

// One convolution layer of the direct path over the `ns` samples of a sub-group, this wavefront's share of the 32-position tiles.
//   in_off / out_off   element offsets of the input / output area inside a sample's LDS region (blocks of an area are contiguous)
//   out_blocks         real output channel blocks (of 8) of the layer
//   flat_c             0: the output is a padded image (the next convolution's input); > 0: the output is area F,
//                      [position][flat_c channels] (the head's input)
//   dummy              element index (from img) of 16 bytes of LDS that belong to this lane alone: stores of lanes without a position
//                      (the last tile's tail) or without a real channel block (conv1's upper half) go there -- no branch in the epilogue
// ONE wavefront per SIMD runs this (PPG_DIRECT_W1), so nothing hides a stall but the code itself: the loop is software-pipelined by
// hand -- the epilogue of tile t (accumulators -> ReLU -> bf16 -> LDS) sits between the first fragment reads of tile t + 1 and its first
// MFMA, where it covers the LDS latency, and the positions of tile t + 1 are computed behind the MFMAs of tile t.
struct TileCtx { int in_base, out_base, swz; bool valid; };

// Area F is [position][flat_c channels] -- RLlib's flatten order.  With 64 channels a position is 128 bytes = ALL the banks of a wide
// store's lane group: conv3's epilogue stores 16 bytes per lane, one POSITION per lane, so the eight lanes of a group hit the same four
// banks (8-way conflicts; 61 % of the pipeline kernels' LDS cycles were bank conflicts, 78 % of the epilogues': profiles/r04).  The
// 16-byte chunk c of position p therefore lives at chunk c ^ (p & 7) of its row (F_SWIZZLE): no padding, every group conflict-free.
// The head reads k-step k (32 features = chunks 4 (k & 1) .. + 3 of position k >> 1) through the same permutation.
template <class KP>
__device__ __forceinline__ int f_koff(const KP &K, int k, int kq) {   // element offset of lane quarter kq's 16 bytes of k-step k
    if (K.flat_c != 64) return 32 * k + 8 * kq;
    const int q = k >> 1;
    return q * 64 + (((4 * (k & 1) + kq) ^ (q & 7)) << 3);
}

// SWP = false (ppg_policy_pipe.h, where a second wavefront on the SIMD fills the gaps): tile by tile, no second set of accumulators.
// F64 = true: the output is an area F of 64 channels per position (F_SWIZZLE above; the caller checks K.flat_c == 64).
// pre: nullptr, or this lane's two precomputed words per tile of this wavefront (tile t = nt_first + t nt_step, t < 2; ppg_policy_pipe.h,
// where a wavefront's tiles are the same positions in every sub-group): pre[2 t] = sample base + 8 x padded position (the element of its
// cell in channel block 0 of area 0), pre[2 t + 1] = (sample base + 64 x position) | (position & 7) -- dconv_cells() below.
// DEPTH: batches of fragment reads in flight ahead of the batch the MFMAs consume (1: the read of batch n + 1 is issued before the MFMAs of
// batch n -- with one-fragment batches the next fragment has 64 cycles of MFMA to arrive, less than an LDS round trip under load: the
// convolution then runs at the LDS latency, not at the matrix pipe's pace; 2-3: a rolling window, still ONE read issued at a time).
template <int CBIN, int MT, int BATCH, bool SWP = true, bool F64 = false, int DEPTH = 1, class KP>
__device__ __forceinline__ void dconv(const KP &K, const ConvW<CBIN, MT> &W, __bf16 *img, int sample_stride, int in_off, int out_off,
                                      int out_blocks, int flat_c, int ns, int nt_first, int nt_step, int lane, int mt_base, int dummy,
                                      const int *pre = nullptr) {
    constexpr int KS = ConvW<CBIN, MT>::KS;
    constexpr int NB = BATCH ? (KS + BATCH - 1) / BATCH : 1, BS = BATCH ? BATCH : KS;   // batches of fragment reads per tile
    const int h = lane >> 5, col = lane & 31;
    const int n_pos = ns * K.P;
    // (with a slot table every tile holds positions of every sample: all of them run as long as there is a sample at all)
    const int n_tiles = pre ? (ns > 0 ? K.slot_tiles : 0) : (n_pos + 31) / 32;
    if (nt_first >= n_tiles) return;
    const int blk = K.Wp2 * 8;
    const int in0 = in_off + (CBIN > 1 ? h : 0) * blk;
    const int pair = 2 * blk;   // distance between the channel-block pairs a k-step's two lane halves read
    const int cb0 = 4 * mt_base + 2 * h;                  // this lane's first channel block
    const int cb_step = flat_c ? 8 : blk;                 // elements from one channel block to the next in the output area
    auto context = [&](int nt) -> TileCtx {
        const int n = 32 * nt + col;
        TileCtx c;
        c.valid = n < n_pos;
        if (pre) {   // (compile-time after inlining: the caller passes an array or nothing)
            const int t = nt == nt_first ? 0 : 1;
            const int cell = t ? pre[2] : pre[0], fp = t ? pre[3] : pre[1];
            c.valid = (t ? pre[5] : pre[4]) < ns;
            c.in_base = cell + in0;
            c.out_base = F64 ? (fp & ~7) + out_off : cell + out_off + __mul24(cb0, cb_step);
            c.swz = F64 ? (fp & 7) : 0;
            return c;
        }
        const int nn = c.valid ? n : 0;
        const int s = div_small(nn, K.magic_P), p = nn - __mul24(s, K.P);
        const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
        const int pidx = __mul24(y + 1, K.Wp) + (x + 1);
        const int sb = __mul24(s, sample_stride);
        c.in_base = sb + in0 + pidx * 8;
        c.out_base = sb + out_off + (flat_c ? __mul24(p, flat_c) : pidx * 8) + (F64 ? 0 : __mul24(cb0, cb_step));
        c.swz = F64 ? (p & 7) : 0;
        return c;
    };
    auto fragment = [&](const TileCtx &c, int ks) -> bf16x8 {
        constexpr int KSB = ConvW<CBIN, MT>::KS_BIAS, HB = ConvW<CBIN, MT>::H_BIAS;
        bf16x8 v;
        if (ks == KSB && CBIN > 1 && HB == 0) {   // (the whole k-step is the bias block + nothing)
            v = zero8();
            if (h == HB) { v[0] = (__bf16)1.0f; v[1] = (__bf16)1.0f; }
        } else {
            if (PPG_DIRECT_ABLATE & 32) {   // (timing-only build: no fragment reads)
                v = zero8();
                v[0] = (__bf16)(float)(c.in_base + ks);
            } else
            v = *(const bf16x8 *)(img + c.in_base + W.offset(K, ks, pair));
            if (ks == KSB && h == HB) { v = zero8(); v[0] = (__bf16)1.0f; v[1] = (__bf16)1.0f; }
        }
        return v;
    };
    auto epilogue = [&](const TileCtx &c, const f32x16 (&acc)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {   // this lane's channels 32 (mt_base + mt) + 16 h + 8 j .. + 7 = channel block cb0 + 4 mt + j
                const bool real = c.valid && (cb0 + 4 * mt + j < out_blocks);
                const int at = !real ? dummy : F64 ? c.out_base + (((cb0 + 4 * mt + j) ^ c.swz) << 3) : c.out_base + (4 * mt + j) * cb_step;
                if (PPG_DIRECT_ABLATE & 64) {   // (timing-only build: no ReLU / pack / store -- one value kept so that the MFMAs stay)
                    if (acc[mt][8 * j] == 123.0f) *(float *)(img + dummy) = acc[mt][8 * j];
                } else
                *(bf16x8 *)(img + at) = relu_pack8(acc[mt], 8 * j);
            }
    };
    // the MFMAs of one tile; `between` runs behind the first batch of fragment reads (the previous tile's epilogue)
    auto tile = [&](const TileCtx &c, f32x16 (&acc)[MT], auto between) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
        constexpr int NSLOT = DEPTH + 1;
        bf16x8 b[NSLOT][BS];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < BS; ++i) if (d * BS + i < KS) b[d][i] = fragment(c, d * BS + i);
        __builtin_amdgcn_sched_barrier(0);   // (keep the reads together and in front: left alone, the scheduler re-pairs each with its MFMA)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (nb + DEPTH < NB) {
#pragma unroll
                for (int i = 0; i < BS; ++i) if ((nb + DEPTH) * BS + i < KS) b[(nb + DEPTH) % NSLOT][i] = fragment(c, (nb + DEPTH) * BS + i);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (nb == 0) between();   // (no fence between the previous tile's epilogue and this batch's MFMAs: the scheduler interleaves them.
                                      //  Measured the same as the epilogue in front of the MFMAs; so did two position tiles per iteration in
                                      //  conv1 / conv2 and a cross-tile fragment prefetch: profiles/r04/b_policy_direct_*)
#pragma unroll
            for (int i = 0; i < BS; ++i)
                if (nb * BS + i < KS) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.a[mt][nb * BS + i], b[nb % NSLOT][i], acc[mt], 0, 0, 0);
                }
        }
    };
    int nt = nt_first;
    if (!SWP) {
        for (; nt < n_tiles; nt += nt_step) {
            const TileCtx c = context(nt);
            f32x16 acc[MT];
            tile(c, acc, [] {});
            epilogue(c, acc);
        }
        return;
    }
    TileCtx cur = context(nt);
    f32x16 acc_prev[MT];
    tile(cur, acc_prev, [] {});
    TileCtx prev = cur;
    for (nt += nt_step; nt < n_tiles; nt += nt_step) {
        cur = context(nt);
        f32x16 acc[MT];
        tile(cur, acc, [&] { epilogue(prev, acc_prev); });   // (requesting tile t + 1's first fragments behind tile t's last batch as
                                                             //  well -- a cross-tile prefetch -- measured the same: profiles/r04)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc_prev[mt] = acc[mt];
        prev = cur;
    }
    epilogue(prev, acc_prev);
}

// DEEP = false: one to three convolutions: their weights stay in registers for the whole launch.
// DEEP = true: more than three convolutions: every layer's weights come from L2 when the layer starts.
template <int OBS, int NCH, bool DEEP>
__device__ __forceinline__ void direct_main(KPtr Kp, unsigned char *lds) {
    constexpr bool W1 = PPG_DIRECT_W1, WRES = W1 && !DEEP;   // (DEEP: 512 registers too, but every layer's weights come per sub-group)
    constexpr int MT3 = WRES ? 2 : 1;    // conv3 row tiles per wavefront
    constexpr int NPOS = W1 ? 2 : 1;     // positions a thread stages per sub-group (ST * P <= 256 NPOS)
    constexpr int HF = WRES ? 18 : 1;    // head fragments of this wavefront that stay in registers
    constexpr int CB1 = NCH > 8 ? 2 : 1;
    constexpr int B12 = PPG_DIRECT_B12;   // fragments per batch in conv1 / conv2 (0 = all of a tile's at once)
    const auto &K = *Kp;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: tile loops, k-step ranges and their branches stay scalar)
    unsigned long long *tab = (unsigned long long *)lds;                    // [TILE][2]: observation row, action slot
    float *red = (float *)(lds + TILE * 16);
    __bf16 *img = (__bf16 *)(lds + TILE * 16 + K.head_mt * 4096 + 4096);
    const int dummy = -2048 + 8 * tid;   // (element index from img: this thread's 16 bytes of the 4 KB in front of the images; dconv)
    const int sample_stride = K.sample_stride;
    const int n_conv = K.n_conv;
    // RANGE MODE of the plan (ppg_policy.h): this workgroup owns samples [begin, end) as tiles of K.range_tile
    const int N = (int)K.plan[0], share = (int)K.plan[1], tpw = (int)K.plan[2];
    const int begin = (int)blockIdx.x * share, end = (begin + share) < N ? (begin + share) : N;
    if (begin >= end) return;
    const int n_slots = (int)gridDim.x * tpw;
    // (A rotation of conv1's / conv2's weights and the head's fragments through the same registers -- each on its way from L2 while the
    // other computes -- was tried: hipcc answered with 40-70 spilled registers in every formulation; profiles/r04.)
    ConvW<CB1, 1> w1c;
    ConvW<2, 1> w2c;
    ConvW<4, MT3> w3c;
    if (!DEEP) {
        w1c.load(K, K.wc1, lane);
        if (n_conv > 1) w2c.load(K, K.wc2, lane);
        if (n_conv > 2) w3c.load(K, K.wc3, lane, WRES ? 0 : wave >> 1);
    }
    // this thread's logits: sample tid >> 4 of the sub-group, actions tid & 15 (+ 16)
    const int smp = tid >> 4, a16 = tid & 15;
    float bias_r[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) bias_r[m] = (16 * m + a16 < K.n_actions) ? K.bh[16 * m + a16] : 0.0f;
    const int apad = 16 * K.head_mt;
    float *lgs = K.lgs + (size_t)blockIdx.x * TILE * apad;    // this workgroup's logits of the current tile (global scratch, L2)
    // halo cells (and area F's slack) are zero and stay zero: only interiors are ever written
    for (int i = tid; i < (K.ST * sample_stride) / 8 + 18 * 4; i += 256) ((bf16x8 *)img)[i] = zero8();   // (+ the slack behind the last region)
    if (!DEEP) {
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w1c.landed();
        if (n_conv > 1) w2c.landed();
        if (n_conv > 2) w3c.landed();
    }

    // head: k-steps of this wavefront, and which of them ride in registers
    const int kq = lane >> 4, colh = lane & 15;
    const int per = (K.kflat_steps + 3) >> 2;
    const int k_lo = wave * per, k_hi = (k_lo + per) < K.kflat_steps ? (k_lo + per) : K.kflat_steps;
    const GLOBAL_AS bf16x8 *wa = (const GLOBAL_AS bf16x8 *)K.wh + lane;
    // [k_lo, k_reg): this wavefront's first HF k-steps, fragments resident in registers -- taken from the per-wavefront table K.whw
    // [wavefront][HF][lane], zero-filled behind the wavefront's share, so that neither the loads nor their MFMAs need a condition (a
    // zero fragment times whatever finite bf16 values lie behind the share in LDS adds nothing)
    const bool resident = WRES && K.head_mt == 1;
    const int k_reg = resident ? ((k_lo + HF) < k_hi ? (k_lo + HF) : k_hi) : k_lo;
    bf16x8 hf[HF];
    if (WRES) {
#pragma unroll
        for (int i = 0; i < HF; ++i) hf[i] = ((const GLOBAL_AS bf16x8 *)K.whw)[((size_t)wave * HF + i) * 64 + lane];
    }
    if (WRES) {
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < HF; ++i) {   // (arrived: no wait-count bookkeeping follows these registers into the loops)
            u32x4_t v = __builtin_bit_cast(u32x4_t, hf[i]);
            __asm__ volatile("" : "+v"(v));
            hf[i] = __builtin_bit_cast(bf16x8, v);
        }
    }
    typedef typename ObsRaw<OBS, NCH>::type raw_t;
#ifdef PPG_DIRECT_PROFILE
    long long dp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dp_prev = (long long)clock64();
#endif
    for (int j = 0; j < tpw; ++j) {
        const int tile = (int)blockIdx.x * tpw + j;
        const int n0 = begin + j * K.range_tile;
        if (n0 >= end) break;
        const int nt_samples = (end - n0) < K.range_tile ? (end - n0) : K.range_tile;
        __syncthreads();   // the previous tile's last readers of the table are done
        if (tid < TILE) {  // sample -> (handle, env, row): walk forward from the tile's first env
            unsigned long long src = 0, dst = 0;
            if (tid < nt_samples) {
                const uint32_t n = (uint32_t)(n0 + tid);
                // the last env whose prefix sum is <= n: bisection between this tile's first env and the next tile's (predator tiles span
                // twenty envs: a forward walk is twenty dependent loads)
                int lo = (int)K.tile_env[tile], hi = tile + 1 < n_slots ? (int)K.tile_env[tile + 1] : K.n_envs - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (K.plan[PLAN_HDR + mid] <= n) lo = mid; else hi = mid - 1;
                }
                const int e = lo, row = (int)(n - K.plan[PLAN_HDR + e]);
                const int k = handle_of(K.env_base, K.n_handles, e);
                const int b = e - K.env_base[k];
                src = (unsigned long long)(uintptr_t)(K.obs[k] + ((size_t)b * K.cap + row) * (size_t)K.obs_elems * (OBS == 2 ? 2 : OBS == 1 ? 4 : 8));
                dst = (unsigned long long)(uintptr_t)(K.actions[k] + (size_t)b * K.S + K.slot0 + row);
            }
            tab[2 * tid] = src;
            tab[2 * tid + 1] = dst;
        }
        __syncthreads();
        raw_t pre[NPOS][NCH];
        auto request = [&](int s0) {   // (a thread stages at most NPOS positions: ST * P <= 256 NPOS, ppg_policy_create_spec)
            const int left = nt_samples - s0, ns = left < K.ST ? left : K.ST;
#pragma unroll
            for (int j = 0; j < NPOS; ++j) {
                const int idx = tid + 256 * j;
#pragma unroll
                for (int c = 0; c < NCH; ++c) pre[j][c] = (raw_t)0;
                if (idx < ns * K.P) {
                    const int s = div_small(idx, K.magic_P), p = idx - __mul24(s, K.P);
                    typedef typename ObsRaw<OBS, NCH>::elem elem_t;
                    const GLOBAL_AS elem_t *src = (const GLOBAL_AS elem_t *)(uintptr_t)tab[2 * (s0 + s)] + p * K.p_stride;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) if (c < K.cin) pre[j][c] = (raw_t)src[c * K.c_stride];
                }
            }
        };
        auto stage = [&](int s0) {
            const int left = nt_samples - s0, ns = left < K.ST ? left : K.ST;
#pragma unroll
            for (int j = 0; j < NPOS; ++j) {
                const int idx = tid + 256 * j;
                if (idx < ns * K.P) {
                    const int s = div_small(idx, K.magic_P), p = idx - __mul24(s, K.P);
                    const int y = div_small(p, K.magic_R), x = p - __mul24(y, K.IW);
#pragma unroll
                    for (int cb = 0; cb < CB1; ++cb) {
                        bf16x8 v = zero8();
#pragma unroll
                        for (int c = 0; c < (NCH < 8 ? NCH : 8); ++c) v[c] = ObsRaw<OBS, NCH>::to_bf16(pre[j][8 * cb + c]);
                        *(bf16x8 *)(img + __mul24(s, sample_stride) + K.off_x + (cb * K.Wp2 + __mul24(y + 1, K.Wp) + (x + 1)) * 8) = v;
                    }
                }
            }
        };
        if (!(PPG_DIRECT_ABLATE & 16)) {
            request(0);
            stage(0);
            if (K.ST < nt_samples) request(K.ST);
        }
        __syncthreads();
        PPG_DP(0);
        for (int s0 = 0; s0 < nt_samples; s0 += K.ST) {
            const int ns = (nt_samples - s0) < K.ST ? (nt_samples - s0) : K.ST;
            // ---- convolutions -------------------------------------------------------------------------------------------
            if (!(PPG_DIRECT_ABLATE & 8)) {
                if (DEEP) { w1c.load(K, K.wc1, lane); w1c.landed(); }
                dconv<CB1, 1, B12>(K, w1c, img, sample_stride, K.off_x, n_conv == 1 ? K.off_f : K.off_y, K.cout_blocks[0],
                                 n_conv == 1 ? K.flat_c : 0, ns, wave, 4, lane, 0, dummy);
                PPG_DP(1);
                __syncthreads();
                PPG_DP(2);
            }
            if (n_conv > 1 && !(PPG_DIRECT_ABLATE & 8)) {
                if (DEEP) { w2c.load(K, K.wc2, lane); w2c.landed(); }
                dconv<2, 1, B12>(K, w2c, img, sample_stride, K.off_y, n_conv == 2 ? K.off_f : K.off_x, K.cout_blocks[1],
                               n_conv == 2 ? K.flat_c : 0, ns, wave, 4, lane, 0, dummy);
                PPG_DP(3);
                __syncthreads();
                PPG_DP(4);
            }
            if (n_conv > 2 && !(PPG_DIRECT_ABLATE & 4)) {
                if (DEEP) { w3c.load(K, K.wc3, lane, wave >> 1); w3c.landed(); }
                if (n_conv == 3 && K.flat_c == 64)
                    dconv<4, MT3, DEEP ? 3 : PPG_DIRECT_B3, true, true>(K, w3c, img, sample_stride, K.off_x, K.off_f, K.cout_blocks[2], 64, ns,
                                                                    WRES ? wave : wave & 1, WRES ? 4 : 2, lane, WRES ? 0 : wave >> 1, dummy);
                else
                dconv<4, MT3, DEEP ? 3 : PPG_DIRECT_B3>(K, w3c, img, sample_stride, K.off_x, n_conv == 3 ? K.off_f : K.off_d0, K.cout_blocks[2],
                                            n_conv == 3 ? K.flat_c : 0, ns, WRES ? wave : wave & 1, WRES ? 4 : 2, lane, WRES ? 0 : wave >> 1, dummy);
            }
            if (n_conv > 2 && !(PPG_DIRECT_ABLATE & 4)) {
                PPG_DP(5);
                __syncthreads();
                PPG_DP(6);
            }
            if (DEEP) {
                for (int l = 3; l < n_conv; ++l) {   // 64 -> 64 channels: D0 -> D1 -> D0 ..., the last one into F
                    ConvW<8, 1> wd;
                    wd.load(K, K.wcd[l - 3], lane, wave >> 1);
                    wd.landed();
                    const int in_off = (l & 1) ? K.off_d0 : K.off_d1, mid_off = (l & 1) ? K.off_d1 : K.off_d0;
                    const bool last = l + 1 == n_conv;
                    if (last && K.flat_c == 64)
                        dconv<8, 1, 3, true, true>(K, wd, img, sample_stride, in_off, K.off_f, K.cout_blocks[l], 64, ns, wave & 1, 2, lane, wave >> 1, dummy);
                    else
                    dconv<8, 1, 3>(K, wd, img, sample_stride, in_off, last ? K.off_f : mid_off, K.cout_blocks[l], last ? K.flat_c : 0,
                                   ns, wave & 1, 2, lane, wave >> 1, dummy);
                    __syncthreads();
                }
            }
            // ---- head: partial logits of this wavefront's share of the k-steps; the next sub-group's input goes into X meanwhile ----
            if (s0 + K.ST < nt_samples && !(PPG_DIRECT_ABLATE & 16)) {
                stage(s0 + K.ST);
                PPG_DP(10);
                if (s0 + 2 * K.ST < nt_samples) request(s0 + 2 * K.ST);
                PPG_DP(11);
            }
            {
                const __bf16 *fb = img + __mul24(colh < ns ? colh : 0, sample_stride) + K.off_f;
                f32x4_t hacc[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hacc[m][i] = 0.0f;
                if (WRES && !(PPG_DIRECT_ABLATE & 1)) {
                    if (resident) {
                        bf16x8 fv[HF];
#pragma unroll
                        for (int i = 0; i < HF; ++i) fv[i] = *(const bf16x8 *)(fb + f_koff(K, k_lo + i, kq));
#pragma unroll
                        for (int i = 0; i < HF; ++i) hacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[i], fv[i], hacc[0], 0, 0, 0);
                        // (four independent accumulator chains instead of this one: slower, 1751 vs 1225 cycles -- profiles/r04)
                    }
                }
                if (PPG_DIRECT_ABLATE & 1) {
                } else if (K.head_mt == 1) {
                    // (k-steps beyond the resident fragments, or all of them: from L2, six requested at once, then their six MFMAs)
                    for (int k0 = k_reg; k0 < k_hi; k0 += 6) {
                        bf16x8 a[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) a[i] = wa[(size_t)((k0 + i) < k_hi ? (k0 + i) : k_lo) * 64];
#pragma unroll
                        for (int i = 0; i < 6; ++i)
                            if (k0 + i < k_hi)
                                hacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], *(const bf16x8 *)(fb + f_koff(K, k0 + i, kq)), hacc[0], 0, 0, 0);
                    }
                } else {
                    for (int k0 = k_lo; k0 < k_hi; k0 += 3) {
                        bf16x8 a[2][3];
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const int ks = (k0 + i) < k_hi ? (k0 + i) : k_lo;
                            a[0][i] = wa[(size_t)ks * 64];
                            a[1][i] = wa[((size_t)K.kflat_steps + ks) * 64];
                        }
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            if (k0 + i < k_hi) {
                                const bf16x8 b = *(const bf16x8 *)(fb + f_koff(K, k0 + i, kq));
                                hacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b, hacc[0], 0, 0, 0);
                                hacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b, hacc[1], 0, 0, 0);
                            }
                    }
                }
                PPG_DP(12);
                // D: lane (column = sample colh, rows 4 kq + i) -> red[wave][action tile][row][column]
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (m < K.head_mt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) red[((wave * K.head_mt + m) * 16 + 4 * kq + i) * 16 + colh] = hacc[m][i];
            }
            PPG_DP(7);
            __syncthreads();
            PPG_DP(8);
            // ---- logits of the sub-group: thread (sample tid >> 4, action tid & 15): bias + the four wavefronts' partial sums in
            // wavefront order -> this workgroup's scratch rows (the actions are chosen for the whole tile at once, below) ----
            if (smp < ns && !(PPG_DIRECT_ABLATE & 2)) {
                const int s_local = s0 + smp;
                for (int m = 0; m < K.head_mt; ++m) {
                    float v = bias_r[m];
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[((w * K.head_mt + m) * 16 + a16) * 16 + smp];
                    lgs[s_local * apad + 16 * m + a16] = v;
                    if (K.logits && 16 * m + a16 < K.n_actions) K.logits[(size_t)(n0 + s_local) * K.n_actions + 16 * m + a16] = v;
                }
            }
            PPG_DP(9);
#ifdef PPG_DIRECT_PROFILE
            dp_acc[15] += 1;
#endif
            // (no barrier here: the next conv1 reads X and writes Y; `red` is written again three barriers from now)
        }
        // ---- actions of the tile: one lane per sample (argmax, or Gumbel-max with Philox keyed by (seed, env, row)) ----
        if (!(PPG_DIRECT_ABLATE & 2)) {
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's scratch rows have been written ...
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ... and stale L1 lines of the previous tile's rows are dropped
            if (tid < nt_samples) {
                const float *row = lgs + tid * apad;
                int8_t *dst = (int8_t *)(uintptr_t)tab[2 * tid + 1];
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
                    for (int i = 0; i < 4; ++i) {
                        if (a4 + i >= K.n_actions) continue;
                        float v = q[i];
                        if (K.sample) {   // Gumbel-max: argmax(logit - log(-log u)) ~ softmax(logits)
                            const float u = (float)(rnd[i] >> 9) * (1.0f / 8388608.0f) + (1.0f / 16777216.0f);   // 23 bits: 2^-24 <= u < 1, exactly
                            v -= __logf(-__logf(u));
                        }
                        if (v > bestv) { bestv = v; best = a4 + i; }
                    }
                }
                *dst = (int8_t)best;
            }
        }
        PPG_DP(0);
    }
#ifdef PPG_DIRECT_PROFILE
    if (K.xg && lane == 0)
        for (int i = 0; i < 16; ++i) ((unsigned long long *)K.xg)[((size_t)blockIdx.x * 4 + wave) * 16 + i] = (unsigned long long)dp_acc[i];
#endif
}

#define PPG_POLICY_DIRECT_KERNEL(name, OBS, NCH, DEEP)                                           \
    extern "C" __global__ void __launch_bounds__(256, PPG_DIRECT_W1 ? 1 : 2) name(const PolParams K) { \
        extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                      \
        direct_main<OBS, NCH, DEEP>((KPtr)__builtin_amdgcn_kernarg_segment_ptr(), lds);          \
    }
// NCH = input channel slots staged per position: 8 (R <= 8 channels-last, or any channel-first image with <= 8 channels) or 16
PPG_POLICY_DIRECT_KERNEL(ppg_policy_direct8_f64, 0, 8, false)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_direct8_f32, 1, 8, false)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_direct8_bf16, 2, 8, false)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_direct16_f64, 0, 16, false)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_direct16_f32, 1, 16, false)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_direct16_bf16, 2, 16, false)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_deep8_f64, 0, 8, true)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_deep8_f32, 1, 8, true)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_deep8_bf16, 2, 8, true)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_deep16_f64, 0, 16, true)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_deep16_f32, 1, 16, true)
PPG_POLICY_DIRECT_KERNEL(ppg_policy_deep16_bf16, 2, 16, true)

}  // namespace ppgpol
