// ppg_hip.hip -- libppg_hip.so: the gfx950 kernels and the HIP backend of include/ppg.h.
//
// Build (see __graft_entry__.build_hip): every unit with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c   (ppg_hip.hip, and ppg_kernels.hip once per
//   -DPPG_TU_GEN=1|2 -DPPG_TU_NQ=1|2|4), then hipcc -shared -o libppg_hip.so *.o
// -ffp-contract=off: energies are IEEE float64 sums that must round exactly like CPython's.
#include <hip/hip_runtime.h>

#include "ppg_host.h"

// The kernels are compiled in separate translation units (ppg_kernels.hip, one per generation and prey-register
// count) so that the build runs in parallel; this unit holds the host side and only declares them.
// kernel name: ppg_<mode>_q<prey registers>[g]   (g = generic observation geometry, descriptors in LDS);
// ppg2_* = second generation (two agent types, stochastic reproduction)
#define PPG_K(name, NQ, MODE, FAST) PPG_KERNEL(name, (NQ <= 2 ? 4 : 2))(const ppg::KParams P);
#define PPG_K2(name, NQ, MODE, FAST) PPG_KERNEL(name, (NQ <= 2 ? 4 : 2))(const ppg::KParams P);
#define PPG_K3(name, NQ, MODE) PPG_KERNEL(name, (NQ <= 2 ? 4 : 2))(const ppg::KParams P);
#define PPG_K4(name, NQ, MODE) PPG_KERNEL(name, (NQ <= 2 ? 4 : 2))(const ppg::KParams P);
#define PPG_KW3(name, NQ, NW) PPG_KERNEL_NW(name, (NQ <= 2 ? 4 : 2), NW)(const ppg::KParams P);
#define PPG_KW4(name, NQ, NW) PPG_KERNEL_NW(name, (NQ <= 2 ? 4 : 2), NW)(const ppg::KParams P);
#define PPG_KW(name, NQ, FAST, NW) PPG_KERNEL_NW(name, (NQ <= 2 ? 4 : 2), NW)(const ppg::KParams P);
#define PPG_KW2(name, NQ, FAST, NW) PPG_KERNEL_NW(name, (NQ <= 2 ? 4 : 2), NW)(const ppg::KParams P);
#define PPG_KC(name, NQ, GEN2, NW) PPG_KERNEL_NW(name, 4, NW)(const ppg::KParams P);
#define PPG_KC3(name, NQ) PPG_KERNEL_NW(name, 4, 4)(const ppg::KParams P);
#define PPG_KCM(name, NQ, GEN2) PPG_KERNEL_NW(name, 4, 4)(const ppg::KParams P);
#define PPG_KCH(name, NQ) PPG_KERNEL_NW(name, 8, 4)(const ppg::KParams P);
#define PPG_KCR(name, NQ, GEN2, NW) PPG_KERNEL_NW(name, 4, NW)(const ppg::KParams P);
#include "ppg_kernel_list.h"

PPG_DEFINE_KERNELSC(1)
PPG_DEFINE_KERNELSC(2)
PPG_DEFINE_KERNELS(1)
PPG_DEFINE_KERNELS(2)
PPG_DEFINE_KERNELS(4)
PPG_DEFINE_KERNELS2(1)
PPG_DEFINE_KERNELS2(2)
PPG_DEFINE_KERNELS2(4)
PPG_DEFINE_KERNELS3(1)
PPG_DEFINE_KERNELS3(2)
PPG_DEFINE_KERNELS3(4)
PPG_DEFINE_KERNELS4(1)
PPG_DEFINE_KERNELS4(2)
PPG_DEFINE_KERNELS4(4)
PPG_DEFINE_KERNELSW(1)
PPG_DEFINE_KERNELSW(2)
PPG_DEFINE_KERNELSW(4)
PPG_DEFINE_KERNELSW2(1)
PPG_DEFINE_KERNELSW2(2)
PPG_DEFINE_KERNELSW2(4)

typedef void (*ppg_kernel_fn)(const ppg::KParams);

static ppg_kernel_fn pick_kernel_gen2(int nq, int mode, bool fast) {
    static const ppg_kernel_fn table[2][3][5] = {
        {{ppg2_step_q1g, ppg2_reset_q1g, ppg2_observe_q1g, ppg2_grid_q1g, ppg2_step_ord_q1g},
         {ppg2_step_q2g, ppg2_reset_q2g, ppg2_observe_q2g, ppg2_grid_q2g, ppg2_step_ord_q2g},
         {ppg2_step_q4g, ppg2_reset_q4g, ppg2_observe_q4g, ppg2_grid_q4g, ppg2_step_ord_q4g}},
        {{ppg2_step_q1, ppg2_reset_q1, ppg2_observe_q1, ppg2_grid_q1, ppg2_step_ord_q1},
         {ppg2_step_q2, ppg2_reset_q2, ppg2_observe_q2, ppg2_grid_q2, ppg2_step_ord_q2},
         {ppg2_step_q4, ppg2_reset_q4, ppg2_observe_q4, ppg2_grid_q4, ppg2_step_ord_q4}},
    };
    return table[fast ? 1 : 0][nq == 1 ? 0 : nq == 2 ? 1 : 2][mode];
}

static ppg_kernel_fn pick_kernel_walls(int nq, int mode) {
    if (mode == ppg::MODE_VIS) return nq == 1 ? ppg3_vis_q1 : nq == 2 ? ppg3_vis_q2 : ppg3_vis_q4;
    static const ppg_kernel_fn table[3][5] = {
        {ppg3_step_q1, ppg3_reset_q1, ppg3_observe_q1, ppg3_grid_q1, ppg3_step_ord_q1},
        {ppg3_step_q2, ppg3_reset_q2, ppg3_observe_q2, ppg3_grid_q2, ppg3_step_ord_q2},
        {ppg3_step_q4, ppg3_reset_q4, ppg3_observe_q4, ppg3_grid_q4, ppg3_step_ord_q4},
    };
    return table[nq == 1 ? 0 : nq == 2 ? 1 : 2][mode];
}

static ppg_kernel_fn pick_kernel_drive(int nq, int mode) {
    static const ppg_kernel_fn table[3][5] = {
        {ppg4_step_q1, ppg4_reset_q1, ppg4_observe_q1, ppg4_grid_q1, ppg4_step_ord_q1},
        {ppg4_step_q2, ppg4_reset_q2, ppg4_observe_q2, ppg4_grid_q2, ppg4_step_ord_q2},
        {ppg4_step_q4, ppg4_reset_q4, ppg4_observe_q4, ppg4_grid_q4, ppg4_step_ord_q4},
    };
    return table[nq == 1 ? 0 : nq == 2 ? 1 : 2][mode];
}

static ppg_kernel_fn pick_kernel(int nq, int mode, bool fast) {
    static const ppg_kernel_fn table[2][3][ppg::N_MODES] = {
        {{ppg_step_q1g, ppg_reset_q1g, ppg_observe_q1g, ppg_grid_q1g, ppg_step_ord_q1g, ppg_rollout_q1g, ppg_step_kick_q1g, ppg_step_ord_kick_q1g},
         {ppg_step_q2g, ppg_reset_q2g, ppg_observe_q2g, ppg_grid_q2g, ppg_step_ord_q2g, ppg_rollout_q2g, ppg_step_kick_q2g, ppg_step_ord_kick_q2g},
         {ppg_step_q4g, ppg_reset_q4g, ppg_observe_q4g, ppg_grid_q4g, ppg_step_ord_q4g, ppg_rollout_q4g, ppg_step_kick_q4g, ppg_step_ord_kick_q4g}},
        {{ppg_step_q1, ppg_reset_q1, ppg_observe_q1, ppg_grid_q1, ppg_step_ord_q1, ppg_rollout_q1, ppg_step_kick_q1, ppg_step_ord_kick_q1},
         {ppg_step_q2, ppg_reset_q2, ppg_observe_q2, ppg_grid_q2, ppg_step_ord_q2, ppg_rollout_q2, ppg_step_kick_q2, ppg_step_ord_kick_q2},
         {ppg_step_q4, ppg_reset_q4, ppg_observe_q4, ppg_grid_q4, ppg_step_ord_q4, ppg_rollout_q4, ppg_step_kick_q4, ppg_step_ord_kick_q4}},
    };
    return table[fast ? 1 : 0][nq == 1 ? 0 : nq == 2 ? 1 : 2][mode];
}

#define PPG_HIP_TRY(h, call)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return ppg_fail(h, PPG_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static int backend_init(ppg_handle *h, int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ppg_fail(h, PPG_ENODEV, "no HIP device visible");
    if (device < 0 || device >= n) return ppg_fail(h, PPG_ENODEV, "device %d not in 0..%d", device, n - 1);
    hipDeviceProp_t prop;
    PPG_HIP_TRY(h, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ppg_fail(h, PPG_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    PPG_HIP_TRY(h, hipSetDevice(device));
    PPG_HIP_TRY(h, hipMalloc((void **)&h->lut_dev, h->lut_host.size() * sizeof(uint32_t)));
    PPG_HIP_TRY(h, hipMemcpy(h->lut_dev, h->lut_host.data(), h->lut_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (h->coop_ok) {
        PPG_HIP_TRY(h, hipMalloc((void **)&h->coop_tab_dev, h->coop_tab_host.size() * sizeof(uint32_t)));
        PPG_HIP_TRY(h, hipMemcpy(h->coop_tab_dev, h->coop_tab_host.data(), h->coop_tab_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return PPG_OK;
}

static int backend_alloc(ppg_handle *h, void **out, size_t bytes) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) PPG_HIP_TRY(h, hipSetDevice(h->device));
    PPG_HIP_TRY(h, hipMalloc(out, bytes));
    return PPG_OK;
}

static void backend_release(ppg_handle *h) {
    if (h->vis_dev) (void)hipFree(h->vis_dev);
    h->vis_dev = nullptr;
    if (h->lut_dev) (void)hipFree(h->lut_dev);
    h->lut_dev = nullptr;
    if (h->coop_tab_dev) (void)hipFree(h->coop_tab_dev);
    h->coop_tab_dev = nullptr;
    if (h->order_dev) (void)hipFree(h->order_dev);
    h->order_dev = nullptr;
    if (h->fetch_dev) (void)hipFree(h->fetch_dev);
    h->fetch_dev = nullptr;
    h->fetch_cap = 0;
}

static void backend_free(ppg_handle *, void *p) { (void)hipFree(p); }

// order[] = the envs sorted by descending key (rows weighted by observation size): a counting sort in ONE workgroup, O(B) for any
// batch size.  Keys are small (<= 64 * 8 + 256 * 8), so the histogram lives in LDS; envs with equal keys land in arbitrary order
// (atomics) -- the order only decides which workgroup steps which env, never a result.
#define PPG_RANK_BINS 2568
extern "C" __global__ void __launch_bounds__(1024) ppg_rank_envs(const int32_t *env_state, int batch, int wp, int wq, int32_t *order) {
    __shared__ int32_t hist[PPG_RANK_BINS];
    const int t = (int)threadIdx.x;
    for (int k = t; k < PPG_RANK_BINS; k += 1024) hist[k] = 0;
    __syncthreads();
    for (int i = t; i < batch; i += 1024) {
        int key = env_state[(size_t)i * PPG_ENV_WORDS + PPG_ENV_N_PRED_ROWS] * wp + env_state[(size_t)i * PPG_ENV_WORDS + PPG_ENV_N_PREY_ROWS] * wq;
        key = key < 0 ? 0 : (key >= PPG_RANK_BINS ? PPG_RANK_BINS - 1 : key);
        atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    if (t < 64) {   // exclusive prefix over the bins in DESCENDING key order: one wavefront, 41 bins per lane
        const int per = (PPG_RANK_BINS + 63) / 64;
        const int hi = PPG_RANK_BINS - 1 - t * per;          // this lane's bins: hi, hi-1, ..., hi-per+1
        int sum = 0;
        for (int q = 0; q < per; ++q) if (hi - q >= 0) sum += hist[hi - q];
        int before = 0;
        for (int l = 0; l < 64; ++l) { const int v = __shfl(sum, l, 64); if (l < t) before += v; }
        for (int q = 0; q < per; ++q) if (hi - q >= 0) { const int c = hist[hi - q]; hist[hi - q] = before; before += c; }
    }
    __syncthreads();
    for (int i = t; i < batch; i += 1024) {
        int key = env_state[(size_t)i * PPG_ENV_WORDS + PPG_ENV_N_PRED_ROWS] * wp + env_state[(size_t)i * PPG_ENV_WORDS + PPG_ENV_N_PREY_ROWS] * wq;
        key = key < 0 ? 0 : (key >= PPG_RANK_BINS ? PPG_RANK_BINS - 1 : key);
        order[atomicAdd(&hist[key], 1)] = i;
    }
}

static int backend_rebalance(ppg_handle *h, int weight_pred, int weight_prey, void *stream) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) PPG_HIP_TRY(h, hipSetDevice(h->device));
    if (!h->order_dev) PPG_HIP_TRY(h, hipMalloc((void **)&h->order_dev, (size_t)h->batch * sizeof(int32_t)));
    hipLaunchKernelGGL(ppg_rank_envs, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       (const int32_t *)h->bufs.env_state, (int)h->batch, weight_pred, weight_prey, h->order_dev);
    PPG_HIP_TRY(h, hipGetLastError());
    return PPG_OK;
}

// ---- ppg_pack: the packed observation image (ppg_pack.h) --------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64) ppg_pack_scan(const ppg::PackParams K) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[512];
    ppg::pack_scan_main(*PPG_KERNARG_PTR(ppg::PackParams, K), lds);
}
extern "C" __global__ void __launch_bounds__(64) ppg_pack_rows(const ppg::PackParams K) {
    ppg::pack_rows_main(*PPG_KERNARG_PTR(ppg::PackParams, K));
}

static int backend_pack(ppg_handle *h, const ppg::PackParams &K, void *stream) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) PPG_HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(ppg_pack_scan, dim3(1), dim3(64), 0, (hipStream_t)stream, K);
    hipLaunchKernelGGL(ppg_pack_rows, dim3((unsigned)K.n_envs), dim3(64), 0, (hipStream_t)stream, K);
    PPG_HIP_TRY(h, hipGetLastError());
    return PPG_OK;
}

// ---- ppg_fetch: the host view of a run of envs (ppg_fetch.h) ------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64) ppg_fetch_rows(const ppg::FetchParams K) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    ppg::fetch_main(*PPG_KERNARG_PTR(ppg::FetchParams, K), lds);
}

static int backend_fetch(ppg_handle *h, const ppg::FetchParams &K, void *stream) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) PPG_HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(ppg_fetch_rows, dim3((unsigned)K.n_envs), dim3(64), 0, (hipStream_t)stream, K);
    PPG_HIP_TRY(h, hipGetLastError());
    return PPG_OK;
}

static int backend_copy(ppg_handle *h, void *dst, const void *src, size_t bytes, bool to_device, void *stream) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) PPG_HIP_TRY(h, hipSetDevice(h->device));
    PPG_HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PPG_OK;
}

static int backend_sync(ppg_handle *h, void *stream) {
    PPG_HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    return PPG_OK;
}

static int backend_launch(ppg_handle *h, int mode, const ppg::KParams &P, void *stream) {
    // one workgroup = one wavefront = one environment
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) PPG_HIP_TRY(h, hipSetDevice(h->device));
    const bool fast = P.nch_p <= 2 && P.nch_q <= 3;
    if (h->gen2 && mode > ppg::MODE_STEP_ORDERED && !(mode == ppg::MODE_VIS && h->cfg2.walls) &&
        !(mode == ppg::MODE_ROLLOUT && !h->cfg2.walls && P.coop_e > 0))
        return ppg_fail(h, PPG_EINVAL, "mode %d is not available for second-generation handles", mode);
    if (h->drive && mode > ppg::MODE_STEP_ORDERED) return ppg_fail(h, PPG_EINVAL, "mode %d is not available for the drive-conditioned variant", mode);
    ppg_kernel_fn fn = h->drive ? pick_kernel_drive(h->nq, mode) : !h->gen2 ? pick_kernel(h->nq, mode, fast)
                       : h->cfg2.walls ? pick_kernel_walls(h->nq, mode) : pick_kernel_gen2(h->nq, mode, fast);
    unsigned block = 64, grid = (unsigned)h->batch;
    const ppg_wave_plan_t wp = h->plan;
    if ((mode == ppg::MODE_STEP || mode == ppg::MODE_ROLLOUT) && P.coop_e > 0) {   // cooperative kernels: coop_e envs per workgroup of wp.nw wavefronts
        static const ppg_kernel_fn c[4][2] = {{ppgc_step_q1, ppgc_step_q2}, {ppgc8_step_q1, ppgc8_step_q2}, {ppgc16_step_q1, ppgc16_step_q2},
                                              {ppgc6_step_q1, ppgc6_step_q2}};
        fn = c[wp.nw == 8 ? 1 : wp.nw == 16 ? 2 : wp.nw == 6 ? 3 : 0][h->nq == 1 ? 0 : 1];
        if (!h->gen2 && wp.nw == 4 && P.obs_f32 == 2 && ppg_coop_high_occupancy(h, wp.coop_e))   // bfloat16 rows: the 64-register build, 8 workgroups per CU
            fn = h->nq == 1 ? ppgch_step_q1 : ppgch_step_q2;
        if (h->gen2) fn = h->nq == 1 ? ppgc2_step_q1 : ppgc2_step_q2;   // (second generation: four-wave cooperative kernels)
        if (!P.ch0_map)   // three cell maps per env (ppg_planned_step_params chose that layout: four-wave step launches only)
            fn = h->gen2 ? (h->nq == 1 ? ppgcm2_step_q1 : ppgcm2_step_q2) : (h->nq == 1 ? ppgcm_step_q1 : ppgcm_step_q2);
        if (h->gen2 && h->cfg2.walls) fn = h->nq == 1 ? ppgc3_step_q1 : ppgc3_step_q2;   // (walls: three maps, rows written whole)
        if (mode == ppg::MODE_ROLLOUT)   // (ppg_rollout: the fused form of the four-wave cooperative kernels)
            fn = h->gen2 ? (h->nq == 1 ? ppgc2_rollout_q1 : ppgc2_rollout_q2) : (h->nq == 1 ? ppgc_rollout_q1 : ppgc_rollout_q2);
        block = 64u * (unsigned)wp.nw;
        grid = (unsigned)((h->batch + P.coop_e - 1) / P.coop_e);
        if (P.lds_bytes > 64 * 1024)
            PPG_HIP_TRY(h, hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes));
    } else if (mode == ppg::MODE_STEP && wp.nw > 1) {   // several waves per env: wave 0 steps, all of them write the final observations
        const int qi = h->nq == 1 ? 0 : h->nq == 2 ? 1 : 2;
        if (h->drive) {
            static const ppg_kernel_fn w4[2][3] = {{ppgw4_step_q1, ppgw4_step_q2, ppgw4_step_q4}, {ppgwp4_step_q1, ppgwp4_step_q2, ppgwp4_step_q4}};
            fn = w4[wp.nw == 2 ? 1 : 0][qi];
        } else if (h->gen2 && h->cfg2.walls) {
            static const ppg_kernel_fn w3[2][3] = {{ppgw3_step_q1, ppgw3_step_q2, ppgw3_step_q4}, {ppgwp3_step_q1, ppgwp3_step_q2, ppgwp3_step_q4}};
            fn = w3[wp.nw == 2 ? 1 : 0][qi];
        } else if (wp.nw == 16) {
            static const ppg_kernel_fn w16[3] = {ppgw16_step_q1, ppgw16_step_q2, ppgw16_step_q4};
            fn = w16[qi];
        } else if (wp.nw == 2) {
            static const ppg_kernel_fn wpair[2][3] = {{ppgwp_step_q1g, ppgwp_step_q2g, ppgwp_step_q4g}, {ppgwp_step_q1, ppgwp_step_q2, ppgwp_step_q4}};
            fn = wpair[fast ? 1 : 0][qi];
        } else {
            static const ppg_kernel_fn w[2][2][2][3] = {
                {{{ppgw_step_q1g, ppgw_step_q2g, ppgw_step_q4g}, {ppgw_step_q1, ppgw_step_q2, ppgw_step_q4}},
                 {{ppgw2_step_q1g, ppgw2_step_q2g, ppgw2_step_q4g}, {ppgw2_step_q1, ppgw2_step_q2, ppgw2_step_q4}}},
                {{{ppgw8_step_q1g, ppgw8_step_q2g, ppgw8_step_q4g}, {ppgw8_step_q1, ppgw8_step_q2, ppgw8_step_q4}},
                 {{ppgw28_step_q1g, ppgw28_step_q2g, ppgw28_step_q4g}, {ppgw28_step_q1, ppgw28_step_q2, ppgw28_step_q4}}}};
            fn = w[wp.nw == 8 ? 1 : 0][h->gen2 ? 1 : 0][fast ? 1 : 0][qi];
        }
        block = 64u * (unsigned)wp.nw;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(block), (size_t)P.lds_bytes, (hipStream_t)stream, P);
    PPG_HIP_TRY(h, hipGetLastError());
    return PPG_OK;
}

// device buffers on physical pages spread over device memory (observation tensors; the policy kernels' scratch slots)
#include "ppg_spread.h"

// policy inference next to the env (MFMA kernels + their host side)
#include "ppg_policy.h"
