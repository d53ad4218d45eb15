// Calibration kernel (not product code): variants of the observation WRITE PATTERN of ppg_step, no compute, footprint beyond the
// 256 MB Infinity Cache.  Which space/time arrangement of the same bytes does HBM take fastest?
//   ./a.out B mean iters [cap=128] [variants: all | base]
// variants:
//  0 base      one wave per env, rows of 324 doubles at stride 324 (the API layout [env][row]), 1 KB per store instruction
//  1 aligned   the same with rows padded to 336 doubles (128-byte aligned rows)
//  2 rowmajor  layout [row][env]
//  3 half      grid B/2, each wave does envs b and b + B/2 one after the other (8 resident waves per CU)
//  4 third     grid B/3, three envs per wave
//  5 quad      4 waves per env (256-thread workgroup), wave w writes rows w, w+4, ... (B workgroups)
//  6 quadchunk 4 waves per env, the 4 waves split EVERY row (adjacent 1 KB chunks at the same time)
//  7 sc1       base with sc1 (write-through) stores
//  8 linear    the same number of bytes as one linear fill (reference point)
//  9 rowwave   ONE SHORT-LIVED WAVE PER ROW, launched in address order (env-major): each writes its 2592-byte row and exits
// 10 rowwave2  one short-lived wave per PAIR of rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(double *p, double a, double b, int nt) {
    d2 v; v.x = a; v.y = b;
    if (nt) __builtin_nontemporal_store(v, (d2 *)p); else *(d2 *)p = v;
}

__global__ void __launch_bounds__(256) pattern(double *obs, const int *rows, int cap, int blk, int stride, int variant, int B, size_t total16) {
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (variant == 8) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total16; i += (size_t)gridDim.x * blockDim.x) st16(obs + 2 * i, 1.0, 2.0, 0);
        return;
    }
    const int nch = (blk + 127) / 128;
    if (variant == 9 || variant == 10) {   // rows[] holds the exclusive prefix sums here: block -> (env, row) by bisection
        const int per = variant == 9 ? 1 : 2;
        const int first = blockIdx.x * per;
        int lo = 0, hi = B - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rows[mid] <= first) lo = mid; else hi = mid - 1; }
        int b = lo, r = first - rows[b];
        for (int k = 0; k < per; ++k) {
            while (b + 1 < B && first + k >= rows[b + 1]) { ++b; }
            r = first + k - rows[b];
            if (first + k >= rows[B]) return;
            double *base = obs + ((size_t)b * cap + r) * stride;
            for (int c = 0; c < nch; ++c) {
                const int e = c * 128 + 2 * ln;
                if (e < blk) st16(base + e, (double)r, (double)c, 0);
            }
        }
        return;
    }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int n = rows[b];
        double *base = variant == 2 ? obs + (size_t)b * stride : obs + (size_t)b * cap * stride;
        const size_t rstride = variant == 2 ? (size_t)B * stride : (size_t)stride;
        if (variant == 6) {
            for (int r = 0; r < n; ++r)
                for (int c = w; c < nch; c += nw) {
                    const int e = c * 128 + 2 * ln;
                    if (e < blk) st16(base + (size_t)r * rstride + e, (double)r, (double)c, 0);
                }
        } else {
            for (int r = w; r < n; r += nw)
                for (int c = 0; c < nch; ++c) {
                    const int e = c * 128 + 2 * ln;
                    if (e < blk) st16(base + (size_t)r * rstride + e, (double)r, (double)c, variant == 7);
                }
        }
    }
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, mean = argc > 2 ? atoi(argv[2]) : 36, iters = argc > 3 ? atoi(argv[3]) : 100;
    const int cap = argc > 4 ? atoi(argv[4]) : 128, blk = 324;   // cap: rows per env slab (the slab stride = cap * row bytes)
    double *obs; int *rows;
    hipMalloc(&obs, (size_t)B * cap * 336 * 8 + 4096);
    hipMalloc(&rows, B * sizeof(int));
    const char *names[] = {"base", "aligned", "rowmajor", "half", "third", "quad", "quadchunk", "nontemporal", "linear", "rowwave", "rowwave2"};
    int *prefix; hipMalloc(&prefix, (B + 1) * sizeof(int));
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<int> h(B);
        unsigned s = 12345; size_t tot = 0;
        for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = mode ? 10 + (s >> 8) % (2 * mean - 19) : mean; tot += h[i]; }
        hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
        std::vector<int> pf(B + 1, 0);
        for (int i = 0; i < B; ++i) pf[i + 1] = pf[i] + h[i];
        hipMemcpy(prefix, pf.data(), (B + 1) * sizeof(int), hipMemcpyHostToDevice);
        const bool only_base = argc > 5;
        for (int variant = 0; variant < 11; ++variant) {
            if (only_base && variant != 0 && variant != 5) continue;
            const int stride = variant == 1 ? 336 : blk;
            const int grid = variant == 3 ? B / 2 : variant == 4 ? (B + 2) / 3 : variant == 8 ? 256 * 8 : variant == 9 ? (int)tot : variant == 10 ? (int)(tot + 1) / 2 : B;
            const int block = (variant == 5 || variant == 6 || variant == 8) ? 256 : 64;
            const size_t total16 = tot * blk / 2;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int *rp = variant >= 9 ? prefix : rows;
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(pattern, dim3(grid), dim3(block), 0, 0, obs, rp, cap, blk, stride, variant, B, total16);
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(pattern, dim3(grid), dim3(block), 0, 0, obs, rp, cap, blk, stride, variant, B, total16);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double bytes = (double)tot * blk * 8;
            printf("%-12s B=%d cap=%d rows %s (mean %.1f): %.1f us per launch, %.2f TB/s\n", names[variant], B, cap, mode ? "spread " : "uniform", (double)tot / B,
                   ms / iters * 1e3, bytes * iters / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
