// Calibration kernel (not product code), round 3: who writes which part of an env's observation slab?  No compute, footprint
// beyond the 256 MB Infinity Cache.  An env's rows in use are the first n rows of its slab, i.e. ONE contiguous run of n * 324
// doubles, so the run can be cut into whole 1 KB pieces that ignore the row boundaries ("slab" variants) instead of 2 whole + 1
// partial store per row.
//   ./a.out [B=4096] [mean=36] [iters=200] [cap=128]
// variants (E = waves per workgroup = envs per workgroup):
//   base        one wave per env, per row 2 full + 1 partial (68-element) store: the product's pattern up to round 2
//   pair        two waves per env, rows alternate between the waves (the round-2 headline kernel's pattern)
//   slab1       one wave per env, the env's run in whole 1 KB pieces
//   pairslab    two waves per env, pieces alternate between the two waves
//   coopE_il    E envs per workgroup of E waves; the waves write env 0's run piece-interleaved, then env 1's, ...
//   coopE_bl    the same, every env's run cut into E contiguous ranges (one per wave)
//   coopE_own   E envs per workgroup of E waves, wave w writes env w's run alone (= slab1 in bigger workgroups: control)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(double *p, double a, double b) { d2 v; v.x = a; v.y = b; *(d2 *)p = v; }

enum { V_BASE, V_PAIR, V_SLAB1, V_PAIRSLAB, V_COOP_IL, V_COOP_BL, V_COOP_OWN };

__global__ void __launch_bounds__(1024) pattern(double *obs, const int *rows, int cap, int blk, int variant, int E, int B) {
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t slab = (size_t)cap * blk;
    if (variant == V_BASE || variant == V_PAIR) {
        const int nw = variant == V_PAIR ? 2 : 1;
        const int b = blockIdx.x;
        const int n = rows[b];
        double *base = obs + (size_t)b * slab;
        for (int r = w; r < n; r += nw)
            for (int c = 0; c < 3; ++c) {
                const int e = c * 128 + 2 * ln;
                if (e < blk) st16(base + (size_t)r * blk + e, (double)r, (double)c);
            }
        return;
    }
    if (variant == V_SLAB1 || variant == V_PAIRSLAB) {
        const int nw = variant == V_PAIRSLAB ? 2 : 1;
        const int b = blockIdx.x;
        const int tot = rows[b] * blk;
        double *base = obs + (size_t)b * slab;
        for (int e = w * 128 + 2 * ln; e < tot; e += nw * 128) st16(base + e, (double)e, 1.0);
        return;
    }
    const int b0 = blockIdx.x * E;
    if (variant == V_COOP_OWN) {
        const int b = b0 + w;
        if (b >= B) return;
        const int tot = rows[b] * blk;
        double *base = obs + (size_t)b * slab;
        for (int e = 2 * ln; e < tot; e += 128) st16(base + e, (double)e, 1.0);
        return;
    }
    for (int k = 0; k < E; ++k) {
        const int b = b0 + k;
        if (b >= B) break;
        const int tot = rows[b] * blk;
        double *base = obs + (size_t)b * slab;
        if (variant == V_COOP_IL) {
            for (int e = w * 128 + 2 * ln; e < tot; e += E * 128) st16(base + e, (double)e, 1.0);
        } else {
            const int pieces = (tot + 127) / 128;
            const int lo = (int)((long long)pieces * w / E), hi = (int)((long long)pieces * (w + 1) / E);
            for (int p = lo; p < hi; ++p) {
                const int e = p * 128 + 2 * ln;
                if (e < tot) st16(base + e, (double)e, 1.0);
            }
        }
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, mean = argc > 2 ? atoi(argv[2]) : 36, iters = argc > 3 ? atoi(argv[3]) : 200;
    const int cap = argc > 4 ? atoi(argv[4]) : 128, blk = 324;
    double *obs; int *rows;
    hipMalloc(&obs, (size_t)B * cap * blk * 8 + 4096);
    hipMalloc(&rows, B * sizeof(int));
    struct V { const char *name; int variant, E; };
    const V vs[] = {{"base", V_BASE, 1}, {"pair", V_PAIR, 2}, {"slab1", V_SLAB1, 1}, {"pairslab", V_PAIRSLAB, 2},
                    {"coop2_il", V_COOP_IL, 2}, {"coop4_il", V_COOP_IL, 4}, {"coop8_il", V_COOP_IL, 8}, {"coop16_il", V_COOP_IL, 16},
                    {"coop2_bl", V_COOP_BL, 2}, {"coop4_bl", V_COOP_BL, 4}, {"coop8_bl", V_COOP_BL, 8}, {"coop16_bl", V_COOP_BL, 16},
                    {"coop4_own", V_COOP_OWN, 4}, {"coop16_own", V_COOP_OWN, 16}};
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<int> h(B);
        unsigned s = 12345; size_t tot = 0;
        for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = mode ? 10 + (s >> 8) % (2 * mean - 19) : mean; tot += h[i]; }
        hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep)   // twice: the first pass of a fresh box runs at a higher clock
        for (const V &v : vs) {
            const bool own_wg = v.variant <= V_PAIRSLAB;
            const int grid = own_wg ? B : (B + v.E - 1) / v.E;
            const int block = 64 * v.E;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(pattern, dim3(grid), dim3(block), 0, 0, obs, rows, cap, blk, v.variant, v.E, B);
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(pattern, dim3(grid), dim3(block), 0, 0, obs, rows, cap, blk, v.variant, v.E, B);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)tot * blk * 8;
            printf("%-11s rows %s pass %d: %6.1f us per launch, %.2f TB/s\n", v.name, mode ? "spread " : "uniform", rep, ms / iters * 1e3,
                   bytes * iters / (ms * 1e-3) / 1e12);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    return 0;
}
