// Experiment (not product code): the observation write pattern with normal vs non-temporal 16-byte stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NT>
__global__ void __launch_bounds__(64) pattern(double *obs, const int *rows, int cap, int blk) {
    const int b = blockIdx.x, ln = threadIdx.x;
    const int n = rows[b];
    double *base = obs + (size_t)b * cap * blk;
    const int nch = (blk + 127) / 128;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < nch; ++c) {
            const int e = c * 128 + 2 * ln;
            if (e < blk) {
                d2 v; v.x = (double)r; v.y = (double)c;
                d2 *p = (d2 *)(base + (size_t)r * blk + e);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        }
}
int main() {
    const int B = 4096, mean = 36, iters = 200, cap = 128, blk = 324;
    double *obs; int *rows;
    hipMalloc(&obs, (size_t)B * cap * blk * 8); hipMalloc(&rows, B * sizeof(int));
    std::vector<int> h(B); unsigned s = 12345; size_t tot = 0;
    for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = 10 + (s >> 8) % (2 * mean - 19); tot += h[i]; }
    hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
    for (int nt = 0; nt < 2; ++nt) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto k = nt ? pattern<1> : pattern<0>;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, obs, rows, cap, blk);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, obs, rows, cap, blk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s stores: %.1f us per launch, %.2f TB/s\n", nt ? "non-temporal" : "normal      ", ms / iters * 1e3, (double)tot * blk * 8 * iters / (ms * 1e-3) / 1e12);
    }
    return 0;
}
