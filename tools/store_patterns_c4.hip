// Calibration kernel (not product code), round 5: BASELINE config 4's write shape with no compute -- what can the pair kernel's launch
// structure (2 wavefronts per env, 21.6 KB of LDS per env = 7 envs per CU) reach when its 1568-byte rows (4 x 7 x 7 float64) are cut
//   A  per row as ppgwp_step does today: one full 1 KB store + one of 34 lanes (544 B) per row, rows alternate between the two waves
//   B  1 KB pieces that ignore row boundaries (the cooperative kernels' cut), piece p to wave p mod 2
//   C  two 784-byte half rows per row: every store instruction 49 lanes x 16 B, half row h to wave h mod 2
//   D  three rows = 4704 B per five store instructions (294 of 320 lanes), groups of three rows alternate between the waves
// Every env's wave 0 first idles for the "transition" (delay_us, 60 % constant + 40 % proportional to the env's rows), wave 1 waits at
// the barrier, then both write.  Envs as S sub-batches on S streams, launches back to back.
//   ./a.out [B=4096] [mean=39] [steps=300] [delay_us=15]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(double *p, double a, double b) { d2 v; v.x = a; v.y = b; *(d2 *)p = v; }

extern __shared__ unsigned char dyn_lds[];
template <int PATTERN>
__global__ void __launch_bounds__(128) step_like(double *obs, const int *rows, int cap, int B, int delay_ticks, int spread_ticks) {
    constexpr int BLK = 196;   // elements per row
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x;
    if (b >= B) return;
    if (delay_ticks < 0) dyn_lds[threadIdx.x] = 1;   // (keeps the allocation)
    const int n = rows[b];
    if (w == 0) {
        const long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        const long long want = delay_ticks + (long long)spread_ticks * n / 39;
        while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
    double *base = obs + (size_t)b * cap * BLK;
    if (PATTERN == 0) {          // A: per row, 64 + 34 lanes
        for (int r = w; r < n; r += 2) {
            double *row = base + (size_t)r * BLK;
            st16(row + 2 * ln, (double)r, 1.0);
            if (ln < 34) st16(row + 128 + 2 * ln, (double)r, 2.0);
        }
    } else if (PATTERN == 1) {   // B: 1 KB pieces across rows
        const int tot = n * BLK;
        for (int e = w * 128 + 2 * ln; e < tot; e += 256) st16(base + e, (double)e, 1.0);
    } else if (PATTERN == 2) {   // C: half rows of 49 lanes
        for (int h = w; h < 2 * n; h += 2)
            if (ln < 49) st16(base + (size_t)h * 98 + 2 * ln, (double)h, 1.0);
    } else {                     // D: three rows per five instructions
        for (int g = w; 3 * g < n; g += 2) {
            const int left = (n - 3 * g < 3 ? n - 3 * g : 3) * BLK;
            double *grp = base + (size_t)g * 3 * BLK;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int e = i * 128 + 2 * ln;
                if (e < left) st16(grp + e, (double)e, 1.0);
            }
        }
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, mean = argc > 2 ? atoi(argv[2]) : 39, steps = argc > 3 ? atoi(argv[3]) : 300;
    const double delay_us = argc > 4 ? atof(argv[4]) : 15.0;
    const int cap = 192, blk = 196;   // (64 predator + 128 prey rows in the product; one slab here)
    double *obs; int *rows;
    (void)hipMalloc(&obs, (size_t)B * cap * blk * 8 + 4096);
    (void)hipMalloc(&rows, B * sizeof(int));
    std::vector<int> h(B);
    unsigned s = 12345; size_t tot = 0;
    for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = 12 + (s >> 8) % (2 * mean - 23); tot += h[i]; }
    (void)hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
    hipStream_t st[8];
    for (int i = 0; i < 8; ++i) (void)hipStreamCreate(&st[i]);
    typedef void (*kern)(double *, const int *, int, int, int, int);
    const kern ks[4] = {step_like<0>, step_like<1>, step_like<2>, step_like<3>};
    const char *names[4] = {"A rows 64+34", "B 1KB pieces", "C half rows 49", "D 3 rows / 5"};
    const int lds = 21600;
    for (int rep = 0; rep < 2; ++rep)
    for (double d : {delay_us, 0.0})
    for (int S : {1, 3})
    for (int p = 0; p < 4; ++p) {
        const int ticks = (int)(d * 100 * 0.6), spread = (int)(d * 100 * 0.4);
        auto launch_all = [&]() {
            for (int k = 0; k < S; ++k) {
                const int lo = (int)((long long)B * k / S), hi = (int)((long long)B * (k + 1) / S), nb = hi - lo;
                hipLaunchKernelGGL(ks[p], dim3(nb), dim3(128), (size_t)lds, st[k], obs + (size_t)lo * cap * blk, rows + lo, cap, nb, ticks, spread);
            }
        };
        for (int i = 0; i < 20; ++i) launch_all();
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) launch_all();
        (void)hipDeviceSynchronize();
        const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("rep %d delay %4.1f us  streams %d  %-16s: %6.1f us per full step, %.2f TB/s\n", rep, d, S, names[p], ms / steps * 1e3,
               (double)tot * blk * 8 * steps / (ms * 1e-3) / 1e12);
    }
    return 0;
}
