// Experiment (not product code), round 4: does the write speed of device memory depend on WHICH 2 MB physical chunk is written?
// ppg_alloc_spread's premise is that where the observation tensors' pages lie decides how fast HBM takes the step's write streams.
// If single chunks differ (a chunk living in a subset of the channels / stacks), buffers could be assembled from measured chunks;
// if every chunk alone takes the full bandwidth, the placement effect is an interaction between MANY pages and no chunk-level
// selection can help.      ./a.out [chunks=96] [passes=200]
//   1. every chunk alone: a kernel of 1024 workgroups writes the 2 MB `passes` times (16 bytes per lane, coalesced)
//   2. groups of 8 chunks written together (16 MB), to see whether some SETS are slower than others
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

// n_chunks chunk base addresses; workgroup g writes 2 KB-strided slices of chunk (g % n_chunks), passes times
__global__ void __launch_bounds__(256) write_chunks(double *const *bases, int n_chunks, size_t chunk_elems, int passes) {
    double *base = bases[blockIdx.x % n_chunks];
    const int part = blockIdx.x / n_chunks, parts = gridDim.x / n_chunks;
    const size_t per = chunk_elems / parts;
    for (int p = 0; p < passes; ++p)
        for (size_t e = (size_t)part * per + 2 * threadIdx.x; e < (size_t)(part + 1) * per; e += 512) {
            d2 v; v.x = (double)p; v.y = 1.0;
            *(d2 *)(base + e) = v;
        }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 96, passes = argc > 2 ? atoi(argv[2]) : 200;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    const size_t chunk = ((size_t)(2u << 20) + gran - 1) / gran * gran;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, chunk * n, chunk, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
    // take the chunks from a wide stretch: create 8x as many, keep every 8th
    std::vector<hipMemGenericAllocationHandle_t> all;
    for (int i = 0; i < n * 8; ++i) { hipMemGenericAllocationHandle_t h; if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) break; all.push_back(h); }
    int mapped = 0;
    for (size_t i = 0; i < all.size() && mapped < n; i += 8) { if (hipMemMap((char *)va + (size_t)mapped * chunk, chunk, 0, all[i], 0) == hipSuccess) ++mapped; }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    (void)hipMemSetAccess(va, chunk * mapped, &acc, 1);
    for (auto h : all) (void)hipMemRelease(h);
    printf("chunk %zu bytes, %d chunks mapped\n", chunk, mapped);
    std::vector<double *> hb(mapped);
    for (int i = 0; i < mapped; ++i) hb[i] = (double *)((char *)va + (size_t)i * chunk);
    double **db; (void)hipMalloc(&db, mapped * sizeof(double *));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const std::vector<double *> &set, int grid_per_chunk, int reps) {
        (void)hipMemcpy(db, set.data(), set.size() * sizeof(double *), hipMemcpyHostToDevice);
        const int grid = (int)set.size() * grid_per_chunk;
        hipLaunchKernelGGL(write_chunks, dim3(grid), dim3(256), 0, 0, db, (int)set.size(), chunk / 8, 2);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(write_chunks, dim3(grid), dim3(256), 0, 0, db, (int)set.size(), chunk / 8, reps);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)set.size() * chunk * reps / (ms * 1e-3) / 1e12;   // TB/s
    };
    std::vector<double> single(mapped);
    for (int i = 0; i < mapped; ++i) single[i] = run({hb[i]}, 1024, passes);
    std::vector<double> s2 = single; std::sort(s2.begin(), s2.end());
    printf("single chunks (1024 workgroups on 2 MB, %d passes): min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f TB/s\n", passes,
           s2.front(), s2[mapped / 10], s2[mapped / 2], s2[mapped * 9 / 10], s2.back());
    for (int rep = 0; rep < 2; ++rep) {
        printf("groups of 8 consecutive chunks (16 MB, 2048 workgroups):");
        for (int g = 0; g + 8 <= mapped; g += 8) {
            std::vector<double *> set(hb.begin() + g, hb.begin() + g + 8);
            printf(" %.2f", run(set, 256, passes / 4));
        }
        printf(" TB/s\n");
    }
    // all chunks together, and the slowest / fastest halves by their single-chunk speed
    std::vector<int> order(mapped);
    for (int i = 0; i < mapped; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return single[a] < single[b]; });
    std::vector<double *> slow, fast, every;
    for (int i = 0; i < mapped; ++i) { (i < mapped / 2 ? slow : fast).push_back(hb[order[i]]); every.push_back(hb[i]); }
    for (int rep = 0; rep < 3; ++rep)
        printf("all %d chunks %.2f TB/s; the half that was slower alone %.2f; the half that was faster alone %.2f\n", mapped,
               run(every, 32, passes / 8), run(slow, 64, passes / 8), run(fast, 64, passes / 8));
    return 0;
}
