// ppg_spread.h -- device buffers whose physical pages are spread over a large stretch of device memory (include/ppg.h: ppg_alloc_spread).
//
// Why (profiles/EXPERIMENTS.md, round 3, profiles/r03/e_placement_experiments.txt): ppg_step writes every env's observation rows as ~2000
// concurrent sequential streams of 1 KB pieces, one per workgroup, into slabs a few hundred KB apart.  How fast HBM takes that
// pattern depends on the PHYSICAL pages behind the tensor: physically contiguous memory is the worst case (105-121 us per 4096-env
// step), what hipMalloc returns draws from 62-91 us, and pages picked at random from a stretch of device memory N times the
// tensor's size get steadily better with N (N = 4: 75 us, 16: 66-68, 32-64: 62-64) -- the memory controllers' address map spreads
// concurrent streams by the upper physical address bits, which a compact allocation does not vary.
// So: reserve a virtual range, create N x as many 2 MB physical chunks as the tensor needs (hipMemCreate), map a random n of
// them in random order (hipMemMap), give the rest back.  HIP virtual memory management API; gfx950 + ROCm 7.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <vector>

namespace ppgspread {

struct Region { size_t size; int device; };
static std::mutex g_mutex;
static std::map<void *, Region> g_regions;
static thread_local char g_error[256] = "";   // (per thread: the allocator has no handle to hang a message on, and no global state to race on)
static uint64_t g_retired_ranges = 0, g_retired_bytes = 0;   // virtual ranges ppg_free_spread has retired (under g_mutex)

static int fail(int code, const char *what, hipError_t e) {
    snprintf(g_error, sizeof g_error, "%s: %s", what, e == hipSuccess ? "failed" : hipGetErrorString(e));
    return code;
}

static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace ppgspread

extern "C" {

int ppg_alloc_spread(int32_t device, uint64_t bytes, int32_t spread, uint64_t seed, void **out) {
    using namespace ppgspread;
    if (!out || bytes == 0 || spread < 1 || spread > 1024) return fail(PPG_EINVAL, "ppg_alloc_spread: bad argument", hipSuccess);
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return fail(PPG_ENODEV, "ppg_alloc_spread: no such device", hipSuccess);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess || gran == 0) return fail(PPG_EHIP, "hipMemGetAllocationGranularity", e);
    const size_t chunk = ((size_t)(2u << 20) + gran - 1) / gran * gran;   // 2 MB pages: smaller ones lose more to address translation than they gain
    const size_t n = ((size_t)bytes + chunk - 1) / chunk, size = n * chunk;
    void *base = nullptr;
    e = hipMemAddressReserve(&base, size, chunk, nullptr, 0);
    if (e != hipSuccess || !base) return fail(PPG_EHIP, "hipMemAddressReserve", e);
    // as many chunks as the spread asks for -- but the transient pool never takes more than half of the memory that is free right now
    // (other processes share the GPU), and fewer if the device runs out all the same (never fewer than the n that are needed)
    size_t want = n * (size_t)spread;
    {
        int cur = -1;
        (void)hipGetDevice(&cur);
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t cap = free_b / 2 / chunk;
            if (want > cap) want = cap > n ? cap : n;
        }
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        (void)hipGetLastError();
    }
    std::vector<hipMemGenericAllocationHandle_t> pool;
    pool.reserve(want);
    for (size_t i = 0; i < want; ++i) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) break;
        pool.push_back(h);
    }
    (void)hipGetLastError();
    if (pool.size() > n + n / 8) {   // leave some of a nearly full device to everybody else
        const size_t give_back = pool.size() < want ? pool.size() / 16 : 0;
        for (size_t i = 0; i < give_back && pool.size() > n; ++i) { (void)hipMemRelease(pool.back()); pool.pop_back(); }
    }
    if (pool.size() < n) {
        for (auto h : pool) (void)hipMemRelease(h);
        (void)hipMemAddressFree(base, size);
        return fail(PPG_ENOMEM, "ppg_alloc_spread: out of device memory", hipErrorOutOfMemory);
    }
    // a random n of the pool (partial Fisher-Yates), mapped in that (random) order
    uint64_t s = seed ^ 0x5DEECE66Dull;
    for (size_t i = 0; i < n; ++i) {
        const size_t j = i + (size_t)(splitmix(s) % (uint64_t)(pool.size() - i));
        std::swap(pool[i], pool[j]);
    }
    int rc = PPG_OK;
    size_t mapped = 0;
    for (; mapped < n; ++mapped) {
        e = hipMemMap((char *)base + mapped * chunk, chunk, 0, pool[mapped], 0);
        if (e != hipSuccess) { rc = fail(PPG_EHIP, "hipMemMap", e); break; }
    }
    if (rc == PPG_OK) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof acc);
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(base, size, &acc, 1);
        if (e != hipSuccess) rc = fail(PPG_EHIP, "hipMemSetAccess", e);
    }
    for (auto h : pool) (void)hipMemRelease(h);   // (a mapped chunk lives on until it is unmapped)
    if (rc != PPG_OK) {
        if (mapped) (void)hipMemUnmap(base, mapped * chunk);
        (void)hipMemAddressFree(base, size);
        return rc;
    }
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        g_regions[base] = Region{size, device};
    }
    *out = base;
    return PPG_OK;
}

int ppg_free_spread(void *ptr) {
    using namespace ppgspread;
    if (!ptr) return PPG_OK;
    Region r;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        auto it = g_regions.find(ptr);
        if (it == g_regions.end()) return fail(PPG_EINVAL, "ppg_free_spread: not a pointer from ppg_alloc_spread", hipSuccess);
        r = it->second;
        g_regions.erase(it);
    }
    // every kernel that may still write the buffer runs on ITS device, which need not be the caller's current one
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != r.device) (void)hipSetDevice(r.device);
    (void)hipDeviceSynchronize();
    hipError_t e = hipMemUnmap(ptr, r.size);   // (gives the physical chunks back: their handles were released after mapping)
    if (cur >= 0 && cur != r.device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return fail(PPG_EHIP, "hipMemUnmap", e);
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        g_retired_ranges += 1;
        g_retired_bytes += r.size;
    }
    // The virtual range is NOT handed back (hipMemAddressFree): a later reservation that got the same addresses read and wrote through
    // stale translations of the old mapping (ROCm 7.2, gfx950: a fill of a re-used range left 13 % of it untouched --
    // tests/test_hip_parity.py::test_spread_allocator_argument_errors).  A dead range costs address space only: 2^47 bytes of it exist,
    // a headline-sized env retires 1.8 GB per create / close cycle -- 70 000 cycles per process; ppg_spread_stats counts them.
    return PPG_OK;
}

/* live and retired ranges of this process: *live_bytes mapped now, *retired_ranges / *retired_bytes virtual ranges given up for good */
int ppg_spread_stats(uint64_t *live_bytes, uint64_t *retired_ranges, uint64_t *retired_bytes) {
    using namespace ppgspread;
    std::lock_guard<std::mutex> lock(g_mutex);
    uint64_t live = 0;
    for (const auto &kv : g_regions) live += kv.second.size;
    if (live_bytes) *live_bytes = live;
    if (retired_ranges) *retired_ranges = g_retired_ranges;
    if (retired_bytes) *retired_bytes = g_retired_bytes;
    return PPG_OK;
}

const char *ppg_spread_last_error(void) { return ppgspread::g_error; }

}  // extern "C"
