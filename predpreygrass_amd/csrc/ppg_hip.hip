// ppg_hip.hip -- libppg_hip.so: the gfx950 kernels and the HIP backend of include/ppg.h.
//
// Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -o libppg_hip.so ppg_hip.hip
// -ffp-contract=off: energies are IEEE float64 sums that must round exactly like CPython's.
#include <hip/hip_runtime.h>

#include "ppg_host.h"

// kernel name: ppg_<mode>_q<prey registers>[g]   (g = generic observation geometry, descriptors in LDS)
#define PPG_K(name, NQ, MODE, FAST)                                                         \
    PPG_KERNEL(name, (NQ <= 2 ? 4 : 2))(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, MODE, FAST>(P, lds); }
#define PPG_DEFINE_KERNELS(NQ)                                        \
    PPG_K(ppg_step_q##NQ, NQ, ppg::MODE_STEP, true)                   \
    PPG_K(ppg_reset_q##NQ, NQ, ppg::MODE_RESET, true)                 \
    PPG_K(ppg_observe_q##NQ, NQ, ppg::MODE_OBSERVE, true)             \
    PPG_K(ppg_grid_q##NQ, NQ, ppg::MODE_EXPORT_GRID, true)            \
    PPG_K(ppg_step_ord_q##NQ, NQ, ppg::MODE_STEP_ORDERED, true)       \
    PPG_K(ppg_rollout_q##NQ, NQ, ppg::MODE_ROLLOUT, true)             \
    PPG_K(ppg_step_kick_q##NQ, NQ, ppg::MODE_STEP_KICK, true)         \
    PPG_K(ppg_step_ord_kick_q##NQ, NQ, ppg::MODE_STEP_ORDERED_KICK, true) \
    PPG_K(ppg_step_q##NQ##g, NQ, ppg::MODE_STEP, false)               \
    PPG_K(ppg_reset_q##NQ##g, NQ, ppg::MODE_RESET, false)             \
    PPG_K(ppg_observe_q##NQ##g, NQ, ppg::MODE_OBSERVE, false)         \
    PPG_K(ppg_grid_q##NQ##g, NQ, ppg::MODE_EXPORT_GRID, false)        \
    PPG_K(ppg_step_ord_q##NQ##g, NQ, ppg::MODE_STEP_ORDERED, false)   \
    PPG_K(ppg_rollout_q##NQ##g, NQ, ppg::MODE_ROLLOUT, false)         \
    PPG_K(ppg_step_kick_q##NQ##g, NQ, ppg::MODE_STEP_KICK, false)     \
    PPG_K(ppg_step_ord_kick_q##NQ##g, NQ, ppg::MODE_STEP_ORDERED_KICK, false)

PPG_DEFINE_KERNELS(1)
PPG_DEFINE_KERNELS(2)
PPG_DEFINE_KERNELS(4)

typedef void (*ppg_kernel_fn)(const ppg::KParams);

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
    return PPG_OK;
}

static void backend_release(ppg_handle *h) {
    if (h->lut_dev) (void)hipFree(h->lut_dev);
    h->lut_dev = nullptr;
}

static int backend_launch(ppg_handle *h, int mode, const ppg::KParams &P, void *stream) {
    // one workgroup = one wavefront = one environment
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) PPG_HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(pick_kernel(h->nq, mode, P.nch_p <= 2 && P.nch_q <= 3), dim3((unsigned)h->batch), dim3(64), (size_t)P.lds_bytes,
                       (hipStream_t)stream, P);
    PPG_HIP_TRY(h, hipGetLastError());
    return PPG_OK;
}
