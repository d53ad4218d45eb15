// Calibration kernel (not product code), round 3: the LAUNCH STRUCTURE of the step with no compute.  Every env's wavefront first
// idles for `delay_us` (the transition: latency-bound, no stores), then the workgroup writes its observation rows; the envs are
// stepped as S sub-batches on S streams, each stream launching back to back (what bench.py / SubBatchedPredPreyGrass do).
//   ./a.out [B=4096] [mean=36] [steps=300] [delay_us=21]
// shapes: pair (2 waves per env, wave 1 waits at the barrier), coop E envs x NW waves (waves < E delay, then all write all)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(double *p, double a, double b) { d2 v; v.x = a; v.y = b; *(d2 *)p = v; }

extern __shared__ unsigned char dyn_lds[];
__global__ void __launch_bounds__(1024) step_like(double *obs, const int *rows, int cap, int blk, int E, int B, int delay_ticks, int spread_ticks) {
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const size_t slab = (size_t)cap * blk;
    const int b0 = blockIdx.x * E;
    if (delay_ticks < 0) dyn_lds[threadIdx.x] = 1;   // (keeps the allocation)
    if (w < E && b0 + w < B) {   // "transition": idle; its length grows with the env's rows like the real one
        const int n = rows[b0 + w];
        const long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        const long long want = delay_ticks + (long long)spread_ticks * n / 36;
        while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
    int at = 0;
    for (int k = 0; k < E; ++k) {
        const int b = b0 + k;
        if (b >= B) break;
        const int tot = rows[b] * blk;
        double *base = obs + (size_t)b * slab;
        int first = w - at; if (first < 0) first += NW;
        for (int e = first * 128 + 2 * ln; e < tot; e += NW * 128) st16(base + e, (double)e, 1.0);
        at = (at + (tot + 127) / 128) % NW;
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, mean = argc > 2 ? atoi(argv[2]) : 36, steps = argc > 3 ? atoi(argv[3]) : 300;
    const double delay_us = argc > 4 ? atof(argv[4]) : 21.0;
    const int cap = 128, blk = 324;
    double *obs; int *rows;
    (void)hipMalloc(&obs, (size_t)B * cap * blk * 8 + 4096);
    (void)hipMalloc(&rows, B * sizeof(int));
    std::vector<int> h(B);
    unsigned s = 12345; size_t tot = 0;
    for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = 10 + (s >> 8) % (2 * mean - 19); tot += h[i]; }
    (void)hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
    hipStream_t st[8];
    for (int i = 0; i < 8; ++i) (void)hipStreamCreate(&st[i]);
    // shapes from the command line: E,NW,ldsKB (dynamic LDS per workgroup: bounds the workgroups per CU like the real kernel's maps)
    struct Shape { char name[32]; int E, NW, lds; };
    std::vector<Shape> shapes;
    for (int a = 5; a < argc; ++a) { Shape sh; sscanf(argv[a], "%d,%d,%d", &sh.E, &sh.NW, &sh.lds); snprintf(sh.name, sizeof sh.name, "E%d_W%d_%dK", sh.E, sh.NW, sh.lds); shapes.push_back(sh); }
    if (shapes.empty()) { shapes.push_back({"pair", 1, 2, 9}); shapes.push_back({"coop42", 2, 4, 18}); shapes.push_back({"coop44", 4, 4, 35}); }
    for (int rep = 0; rep < 2; ++rep)
    for (double d : {delay_us})
    for (int S : {2, 3})
    for (const Shape &sh : shapes) {
        // constant part 60 %, row-proportional part 40 % of the mean delay
        const int ticks = (int)(d * 100 * 0.6), spread = (int)(d * 100 * 0.4);
        auto launch_all = [&]() {
            for (int k = 0; k < S; ++k) {
                const int lo = (int)((long long)B * k / S), hi = (int)((long long)B * (k + 1) / S), nb = hi - lo;
                hipLaunchKernelGGL(step_like, dim3((nb + sh.E - 1) / sh.E), dim3(64 * sh.NW), (size_t)sh.lds * 1024, st[k], obs + (size_t)lo * cap * blk, rows + lo, cap, blk, sh.E, nb, ticks, spread);
            }
        };
        for (int i = 0; i < 20; ++i) launch_all();
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();   // wall clock around ALL streams (an event on one stream sees only that stream)
        for (int i = 0; i < steps; ++i) launch_all();
        (void)hipDeviceSynchronize();
        const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("rep %d delay %4.1f us  streams %d  %-12s: %6.1f us per full step, %.2f TB/s\n", rep, d, S, sh.name, ms / steps * 1e3,
               (double)tot * blk * 8 * steps / (ms * 1e-3) / 1e12);
    }
    return 0;
}
