// Calibration kernel (not product code), round 4: the cooperative LAUNCH STRUCTURE of the step with no compute (store_patterns4)
// writing its observation rows either into per-env SLABS [B, cap, blk] (what ppg_step does) or into ONE DENSE region in which
// the rows in use of all envs follow each other (the ppg_pack layout).  Question: is the dense pattern's
// speed independent of where the driver put the buffer and of which MI355X of the pool it runs on?
//   ./a.out [B=4096] [mean=36] [steps=300] [delay_us=21] [draws=4]
// modes: slab | dense_oracle (offsets = host-computed exclusive prefix sums: the bare pattern, what any scheme can reach at best)
//        dense_atomic (every transition wave takes its env's offset from one atomic cursor per sub-batch when its transition ends)
//        dense_lookback (env-ordered offsets by a decoupled look-back scan over the envs of the sub-batch: deterministic layout)
// Every draw is a fresh set of hipMalloc'ed buffers (plain allocations; nothing is freed in between, so the draws differ).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(double *p, double a, double b) { d2 v; v.x = a; v.y = b; *(d2 *)p = v; }

enum { SLAB = 0, DENSE_ORACLE = 1, DENSE_ATOMIC = 2, DENSE_LOOKBACK = 3, FRONTIER = 4 };
constexpr unsigned long long ST_AGG = 1ull, ST_INCL = 2ull;
__device__ __forceinline__ unsigned long long pk(unsigned epoch, unsigned long long st, unsigned long long v) {
    return ((unsigned long long)epoch << 40) | (st << 38) | v;
}

extern __shared__ unsigned char dyn_lds[];
__global__ void __launch_bounds__(1024) step_like(double *obs, const int *rows, const unsigned *excl, unsigned long long *cursor,
                                                  unsigned long long *desc, unsigned epoch, int mode, int cap, int blk, int E, int B,
                                                  int delay_ticks, int spread_ticks) {
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const size_t slab = (size_t)cap * blk;
    const int b0 = blockIdx.x * E;
    unsigned long long *offs = (unsigned long long *)dyn_lds;   // [E] element offsets of the workgroup's envs
    if (w < E && b0 + w < B) {   // "transition": idle; its length grows with the env's rows like the real one
        const int b = b0 + w;
        const int n = rows[b];
        const long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        const long long want = delay_ticks + (long long)spread_ticks * n / 36;
        while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(32);
        unsigned long long off = 0;
        if (mode == SLAB) off = (unsigned long long)b * slab;
        else if (mode == DENSE_ORACLE || mode == FRONTIER) off = (unsigned long long)excl[b] * blk;
        else if (mode == DENSE_ATOMIC) {
            unsigned long long o = 0;
            if (ln == 0) o = atomicAdd(&cursor[epoch & 1u], (unsigned long long)n);
            if (b == 0 && ln == 1) cursor[(epoch + 1u) & 1u] = 0;   // (the other parity's launch is long over)
            off = __shfl(o, 0) * (unsigned long long)blk;
        } else {
            // decoupled look-back over env index: publish this env's count, then add up the predecessors' until one has its prefix
            if (ln == 0) __hip_atomic_store(&desc[b], pk(epoch, b == 0 ? ST_INCL : ST_AGG, (unsigned long long)n), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long ex = 0;
            int pos = b - 1;
            while (pos >= 0) {
                const int idx = pos - ln;
                unsigned long long d;
                for (;;) {
                    d = idx >= 0 ? __hip_atomic_load(&desc[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : pk(epoch, ST_INCL, 0);
                    if (__ballot((unsigned)(d >> 40) != epoch) == 0ull) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                const unsigned long long incl = __ballot(((d >> 38) & 3ull) == ST_INCL);
                const int k = incl ? __ffsll((long long)incl) - 1 : 63;
                unsigned long long v = ln <= k ? (d & ((1ull << 38) - 1)) : 0ull;
                for (int s = 32; s; s >>= 1) v += __shfl_xor(v, s);
                ex += v;
                if (incl) break;
                pos -= 64;
            }
            if (ln == 0 && b != 0) __hip_atomic_store(&desc[b], pk(epoch, ST_INCL, ex + (unsigned long long)n), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            off = ex * (unsigned long long)blk;
        }
        if (ln == 0) offs[w] = off;
    }
    __syncthreads();
    if (mode == FRONTIER) {   // compact write frontier: piece p of the sub-batch's dense region to wave p mod (all waves of the launch)
        const size_t total = (size_t)cursor[4] * blk;   // (host-provided: elements of the sub-batch)
        const size_t stride = (size_t)gridDim.x * NW * 128;
        for (size_t e = ((size_t)blockIdx.x * NW + w) * 128 + 2 * ln; e < total; e += stride) st16(obs + e, (double)e, 1.0);
        return;
    }
    int at = 0;
    for (int k = 0; k < E; ++k) {
        const int b = b0 + k;
        if (b >= B) break;
        const int tot = rows[b] * blk;
        double *base = obs + offs[k];
        int first = w - at; if (first < 0) first += NW;
        for (int e = first * 128 + 2 * ln; e < tot; e += NW * 128) st16(base + e, (double)e, 1.0);
        at = (at + (tot + 127) / 128) % NW;
    }
}

__global__ void __launch_bounds__(256) plain_fill(double *obs, size_t n_elems) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 2;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; e < n_elems; e += stride) st16(obs + e, (double)e, 1.0);
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, mean = argc > 2 ? atoi(argv[2]) : 36, steps = argc > 3 ? atoi(argv[3]) : 300;
    const double delay_arg = argc > 4 ? atof(argv[4]) : 21.0;
    const int draws = argc > 5 ? atoi(argv[5]) : 4;
    const bool with_lookback = argc > 6 && atoi(argv[6]) != 0;
    const int cap = 128, blk = 324, E = 2, NW = 4, lds = 18;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    printf("device %s uuid ", prop.name); for (int i = 0; i < 16; ++i) printf("%02x", (unsigned char)prop.uuid.bytes[i]); printf("\n");
    std::vector<int> h(B);
    std::vector<unsigned> hx(B);
    unsigned s = 12345; size_t tot = 0;
    for (int i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = 10 + (s >> 8) % (2 * mean - 19); tot += h[i]; }
    int *rows; unsigned *excl; unsigned long long *cursor, *desc;
    (void)hipMalloc(&rows, B * sizeof(int));
    (void)hipMalloc(&excl, B * sizeof(unsigned));
    (void)hipMalloc(&cursor, 8 * 8 * 8);   // per sub-batch 8 words: [0], [1] the cursors of even / odd launches, [4] rows of the sub-batch
    (void)hipMalloc(&desc, (size_t)B * 8);
    (void)hipMemset(desc, 0, (size_t)B * 8);
    (void)hipMemcpy(rows, h.data(), B * sizeof(int), hipMemcpyHostToDevice);
    hipStream_t st[8];
    for (int i = 0; i < 8; ++i) (void)hipStreamCreate(&st[i]);
    const char *names[5] = {"slab", "dense_oracle", "dense_atomic", "dense_lookback", "frontier"};
    unsigned epoch = 1;
    for (int draw = 0; draw < draws; ++draw) {
        double *obs_slab, *obs_dense;
        if (hipMalloc(&obs_slab, (size_t)B * cap * blk * 8 + 4096) != hipSuccess) break;
        if (hipMalloc(&obs_dense, (size_t)tot * blk * 8 + (size_t)B * 4096) != hipSuccess) break;   // (the dense region: the rows in use + slack)
        {   // a linear fill of the same number of bytes (grid-stride, 16 bytes per lane), on both buffers
            for (double *buf : {obs_slab, obs_dense}) {
                for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(plain_fill, dim3(256 * 8), dim3(256), 0, st[0], buf, tot * blk);
                (void)hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(plain_fill, dim3(256 * 8), dim3(256), 0, st[0], buf, tot * blk);
                (void)hipDeviceSynchronize();
                const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                printf("draw %d linear fill of %s: %6.1f us, %.2f TB/s\n", draw, buf == obs_slab ? "slab buffer " : "dense buffer", ms / 100 * 1e3,
                       (double)tot * blk * 8 * 100 / (ms * 1e-3) / 1e12);
            }
        }
        for (int S : {3, 1}) {
            // per sub-batch: its own exclusive prefix (each sub-batch owns a dense region of its own, like a handle)
            std::vector<size_t> sub_base(S + 1, 0);
            std::vector<unsigned long long> hc(64, 0);
            for (int k = 0; k < S; ++k) {
                const int lo = (int)((long long)B * k / S), hi = (int)((long long)B * (k + 1) / S);
                unsigned acc = 0;
                for (int i = lo; i < hi; ++i) { hx[i] = acc; acc += (unsigned)h[i]; }
                sub_base[k + 1] = sub_base[k] + (size_t)acc * blk + 512;
                hc[8 * k + 4] = acc;
            }
            (void)hipMemcpy(excl, hx.data(), B * sizeof(unsigned), hipMemcpyHostToDevice);
            (void)hipMemcpy(cursor, hc.data(), 64 * 8, hipMemcpyHostToDevice);
            for (double delay_us : {delay_arg, 0.0})
            for (int rep = 0; rep < 2; ++rep)
            for (int mode = 0; mode < 5; ++mode) {
                if (mode == DENSE_LOOKBACK && !(with_lookback && draw == 0 && rep == 0)) continue;
                const int ticks = (int)(delay_us * 100 * 0.6), spread = (int)(delay_us * 100 * 0.4);
                auto launch_all = [&]() {
                    for (int k = 0; k < S; ++k) {
                        const int lo = (int)((long long)B * k / S), hi = (int)((long long)B * (k + 1) / S), nb = hi - lo;
                        double *o = mode == SLAB ? obs_slab + (size_t)lo * cap * blk : obs_dense + sub_base[k];
                        hipLaunchKernelGGL(step_like, dim3((nb + E - 1) / E), dim3(64 * NW), (size_t)lds * 1024, st[k], o, rows + lo, excl + lo,
                                           cursor + 8 * k, desc + lo, epoch, mode, cap, blk, E, nb, ticks, spread);
                    }
                    ++epoch;
                };
                for (int i = 0; i < 20; ++i) launch_all();
                (void)hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();   // wall clock around ALL streams
                for (int i = 0; i < steps; ++i) launch_all();
                (void)hipDeviceSynchronize();
                const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                printf("draw %d streams %d delay %4.1f rep %d %-15s: %6.1f us per full step, %.2f TB/s\n", draw, S, delay_us, rep, names[mode],
                       ms / steps * 1e3, (double)tot * blk * 8 * steps / (ms * 1e-3) / 1e12);
                fflush(stdout);
            }
        }
        // (not freed: the next draw gets other pages)
    }
    return 0;
}
