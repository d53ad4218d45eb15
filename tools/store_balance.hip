// Experiment (not product code): does the ORDER in which envs of different sizes are assigned to workgroups change the time of
// the observation write pattern?  (load balance between CUs)  Also prints how workgroups map to CUs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
__device__ __forceinline__ unsigned cu_index() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
}
__global__ void __launch_bounds__(64) pattern(double *obs, const int *rows, const int *perm, int cap, int blk, unsigned *where) {
    const int b = perm[blockIdx.x], ln = threadIdx.x;
    const int n = rows[b];
    if (where && ln == 0) where[blockIdx.x] = cu_index();
    double *base = obs + (size_t)b * cap * blk;
    const int nch = (blk + 127) / 128;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < nch; ++c) {
            const int e = c * 128 + 2 * ln;
            if (e < blk) { double2 v; v.x = (double)r; v.y = (double)c; *(double2 *)(base + (size_t)r * blk + e) = v; }
        }
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, mean = 36, iters = 200, cap = 128, blk = 324;
    double *obs; int *rows, *perm; unsigned *where;
    hipMalloc(&obs, (size_t)B * cap * blk * 8); hipMalloc(&rows, B * sizeof(int)); hipMalloc(&perm, B * sizeof(int)); hipMalloc(&where, B * sizeof(unsigned));
    std::vector<int> h(B); unsigned s = 12345; size_t tot = 0;
    for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = 10 + (s >> 8) % (2 * mean - 19); tot += h[i]; }
    hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
    std::vector<int> id(B), sorted(B);
    std::iota(id.begin(), id.end(), 0);
    sorted = id;
    std::sort(sorted.begin(), sorted.end(), [&](int a, int c) { return h[a] > h[c]; });
    const char *names[] = {"identity (random sizes)", "sorted descending", "sorted, dealt in groups of 16", "sorted, heavy/light alternating"};
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<int> p(B);
        if (mode == 0) p = id;
        else if (mode == 1) p = sorted;
        else if (mode == 2) { const int G = B / 16; for (int i = 0; i < B; ++i) { const int g = i / 16, k = i % 16; const int src = k * G + g; p[i] = sorted[src < B ? src : i]; } }
        else { for (int i = 0; i < B; ++i) p[i] = (i & 1) ? sorted[B - 1 - i / 2] : sorted[i / 2]; }
        hipMemcpy(perm, p.data(), B * sizeof(int), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pattern, dim3(B), dim3(64), 0, 0, obs, rows, perm, cap, blk, (unsigned *)nullptr);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(pattern, dim3(B), dim3(64), 0, 0, obs, rows, perm, cap, blk, (unsigned *)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s: %.1f us per launch, %.2f TB/s\n", names[mode], ms / iters * 1e3, (double)tot * blk * 8 * iters / (ms * 1e-3) / 1e12);
    }
    // mapping of workgroups to CUs
    std::vector<int> p = id; hipMemcpy(perm, p.data(), B * sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(pattern, dim3(B), dim3(64), 0, 0, obs, rows, perm, cap, blk, where);
    std::vector<unsigned> w(B); hipMemcpy(w.data(), where, B * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("first 40 workgroups -> (xcc,cu): ");
    for (int i = 0; i < 40; ++i) printf("%u.%u ", w[i] >> 8, w[i] & 255u);
    printf("\n");
    return 0;
}
