// Calibration kernel (not product code): the observation WRITE PATTERN of ppg_step without any compute.
// One 64-lane workgroup per env writes rows[b] blocks of `blk` doubles (16-byte stores, 128 doubles per
// wave instruction) into its own stride-`cap*blk` region.  Usage: ./a.out B rows_mean iters [cap]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
// layout 0: [env][row][blk] (the API layout); layout 1: [row][env][blk] (same row index adjacent across envs)
__global__ void __launch_bounds__(64) pattern(double *obs, const int *rows, int cap, int blk, int layout, int B) {
    const int b = blockIdx.x, ln = threadIdx.x;
    const int n = rows[b];
    double *base = layout ? obs + (size_t)b * blk : obs + (size_t)b * cap * blk;
    const size_t rstride = layout ? (size_t)B * blk : (size_t)blk;
    const int nch = (blk + 127) / 128;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < nch; ++c) {
            const int e = c * 128 + 2 * ln;
            if (e < blk) { double2 v; v.x = (double)r; v.y = (double)c; *(double2 *)(base + (size_t)r * rstride + e) = v; }
        }
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, mean = argc > 2 ? atoi(argv[2]) : 36, iters = argc > 3 ? atoi(argv[3]) : 200;
    const int cap = argc > 4 ? atoi(argv[4]) : 128, blk = 324;   // cap = rows per env slot (stride between envs = cap * blk doubles)
    double *obs; int *rows;
    hipMalloc(&obs, (size_t)B * cap * blk * 8);
    hipMalloc(&rows, B * sizeof(int));
    for (int layout = 0; layout < 2; ++layout)
    for (int mode = 0; mode < 2; ++mode) {  // 0: every env `mean` rows; 1: spread 10..(2*mean-10) like real populations
        std::vector<int> h(B);
        unsigned s = 12345; size_t tot = 0;
        for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = mode ? 10 + (s >> 8) % (2 * mean - 19) : mean; tot += h[i]; }
        hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pattern, dim3(B), dim3(64), 0, 0, obs, rows, cap, blk, layout, B);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(pattern, dim3(B), dim3(64), 0, 0, obs, rows, cap, blk, layout, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)tot * blk * 8;
        printf("layout %s B=%d rows %s (mean %.1f): %.1f us per launch, %.2f TB/s\n", layout ? "[row][env]" : "[env][row]", B, mode ? "spread" : "uniform", (double)tot / B,
               ms / iters * 1e3, bytes * iters / (ms * 1e-3) / 1e12);
    }
    return 0;
}
