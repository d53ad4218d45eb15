// Calibration (not product code): which LINEAR write pattern does MI355X's HBM take fastest?  torch's fill_ reaches 6.5-6.7 TB/s,
// a naive grid-stride 16-B-per-lane fill 4.7 TB/s on the same box.   ./a.out MiB iters
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// v0: grid-stride, 16 B per lane   v1: grid-stride, 32 B per lane (two 16-B stores)
// v2: one-shot blocks of 256 threads x 32 B (8 KB per block, torch's shape)   v3: one-shot blocks of 256 threads x 16 B (4 KB)
// v4: one-shot single-wave blocks writing `chunk` bytes contiguous, 16 B per lane per instruction
// v5: 4096 persistent waves, each streams its own contiguous region (total / 4096), 1 KB per instruction (the env pattern)
// v6: like v5 but 32 B per lane (2 KB per instruction pair)
// v7: like v5 (every wave streams `per` contiguous bytes), but the waves' regions lie `chunk16` x 16 B apart: a SPARSE footprint, like
//     the capacity-strided observation slabs of which only the rows in use are written
__global__ void __launch_bounds__(256) fill(double *p, size_t n16, int v, int chunk16) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    d2 val; val.x = 1.0; val.y = 2.0;
    d2 *q = (d2 *)p;
    if (v == 0) { for (size_t i = tid; i < n16; i += nth) q[i] = val; }
    else if (v == 1) { for (size_t i = tid; 2 * i + 1 < n16; i += nth) { q[2 * i] = val; q[2 * i + 1] = val; } }
    else if (v == 2) { const size_t i = tid; if (2 * i + 1 < n16) { q[2 * i] = val; q[2 * i + 1] = val; } }
    else if (v == 3) { if (tid < n16) q[tid] = val; }
    else if (v == 4) {
        const size_t base = (size_t)blockIdx.x * chunk16;
        for (int i = threadIdx.x; i < chunk16; i += 64) if (base + i < n16) q[base + i] = val;
    } else if (v == 5) {
        const size_t per = n16 / gridDim.x, base = (size_t)blockIdx.x * per;
        for (size_t i = threadIdx.x; i < per; i += 64) q[base + i] = val;
    } else if (v == 6) {
        const size_t per = n16 / gridDim.x, base = (size_t)blockIdx.x * per;
        for (size_t i = threadIdx.x; 2 * i + 1 < per; i += 64) { q[base + 2 * i] = val; q[base + 2 * i + 1] = val; }
    } else if (v == 7) {
        const size_t per = n16 / gridDim.x, base = (size_t)blockIdx.x * chunk16;
        for (size_t i = threadIdx.x; i < per; i += 64) q[base + i] = val;
    }
}
int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atoi(argv[1]) : 384; const int iters = argc > 2 ? atoi(argv[2]) : 50;
    const size_t bytes = mib << 20, n16 = bytes / 16;
    double *p; hipMalloc(&p, bytes * 6);   // (v7 spreads its writes over up to 6x the bytes written)
    struct { const char *name; int v; int block; size_t grid; int chunk16; } cases[] = {
        {"grid-stride 16B/lane, 2048x256", 0, 256, 2048, 0},
        {"grid-stride 16B/lane, 1024x256", 0, 256, 1024, 0},
        {"grid-stride 32B/lane, 2048x256", 1, 256, 2048, 0},
        {"one-shot 256 thr x 32B (8 KB/block)", 2, 256, n16 / 512, 0},
        {"one-shot 256 thr x 16B (4 KB/block)", 3, 256, n16 / 256, 0},
        {"one-shot wave, 1 KB", 4, 64, n16 / 64, 64},
        {"one-shot wave, 4 KB", 4, 64, n16 / 256, 256},
        {"one-shot wave, 16 KB", 4, 64, n16 / 1024, 1024},
        {"one-shot wave, 96 KB", 4, 64, n16 / 6144, 6144},
        {"4096 persistent waves, own region, 1 KB/instr", 5, 64, 4096, 0},
        {"4096 persistent waves, own region, 32B/lane", 6, 64, 4096, 0},
        {"1024 persistent waves, own region, 1 KB/instr", 5, 64, 1024, 0},
        {"4096 persistent waves, regions 1.0x apart (dense)", 7, 64, 4096, (int)(n16 / 4096)},
        {"4096 persistent waves, regions 1.5x apart", 7, 64, 4096, (int)(n16 / 4096 * 3 / 2)},
        {"4096 persistent waves, regions 2x apart", 7, 64, 4096, (int)(n16 / 4096 * 2)},
        {"4096 persistent waves, regions 4.5x apart", 7, 64, 4096, (int)(n16 / 4096 * 9 / 2)},
        {"4096 persistent waves, regions 6x apart", 7, 64, 4096, (int)(n16 / 4096 * 6)},
    };
    for (auto &c : cases) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill, dim3((unsigned)c.grid), dim3(c.block), 0, 0, p, n16, c.v, c.chunk16);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(fill, dim3((unsigned)c.grid), dim3(c.block), 0, 0, p, n16, c.v, c.chunk16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-48s %zu MiB: %.1f us, %.2f TB/s\n", c.name, mib, ms / iters * 1e3, (double)bytes * iters / (ms * 1e-3) / 1e12);
    }
    return 0;
}
