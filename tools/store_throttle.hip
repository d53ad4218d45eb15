// Experiment (not product code): the observation write pattern with at most L storing waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned cu_index() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    return ((xcc & 15u) << 8) | (se << 5) | (sh << 4) | cu;
}
__global__ void __launch_bounds__(64) pattern(double *obs, const int *rows, int cap, int blk, int *tickets, int limit, int delay) {
    const int b = blockIdx.x, ln = threadIdx.x;
    const int n = rows[b];
    // fake "transition" phase: every wave idles for a while first (all waves in phase, like the real kernel)
    for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(127);
    int *t = tickets + cu_index();
    if (limit > 0) {
        for (;;) {
            int old = 0;
            if (ln == 0) old = atomicAdd(t, 1);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old < limit) break;
            if (ln == 0) atomicSub(t, 1);
            __builtin_amdgcn_s_sleep(64);
        }
    }
    double *base = obs + (size_t)b * cap * blk;
    const int nch = (blk + 127) / 128;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < nch; ++c) {
            const int e = c * 128 + 2 * ln;
            if (e < blk) { double2 v; v.x = (double)r; v.y = (double)c; *(double2 *)(base + (size_t)r * blk + e) = v; }
        }
    if (limit > 0) {
        __builtin_amdgcn_s_waitcnt(0);  // stores issued; release when they have left the wave
        if (ln == 0) atomicSub(t, 1);
    }
}
int main(int argc, char **argv) {
    const int B = 4096, mean = 36, iters = 200, cap = 128, blk = 324;
    double *obs; int *rows, *tickets;
    hipMalloc(&obs, (size_t)B * cap * blk * 8); hipMalloc(&rows, B * sizeof(int)); hipMalloc(&tickets, 4096 * sizeof(int));
    hipMemset(tickets, 0, 4096 * sizeof(int));
    std::vector<int> h(B); unsigned s = 12345; size_t tot = 0;
    for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = 10 + (s >> 8) % (2 * mean - 19); tot += h[i]; }
    hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
    for (int limit : {0, 12, 8, 6, 4, 3}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pattern, dim3(B), dim3(64), 0, 0, obs, rows, cap, blk, tickets, limit, 0);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(pattern, dim3(B), dim3(64), 0, 0, obs, rows, cap, blk, tickets, limit, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("limit %2d storing waves per CU: %.1f us per launch, %.2f TB/s\n", limit, ms / iters * 1e3, (double)tot * blk * 8 * iters / (ms * 1e-3) / 1e12);
    }
    int ht[4096]; hipMemcpy(ht, tickets, sizeof ht, hipMemcpyDeviceToHost); int nz = 0; for (int i = 0; i < 4096; ++i) nz += ht[i] != 0;
    printf("tickets left nonzero: %d\n", nz);
    return 0;
}
