// wave_emu.h -- TEST-ONLY lockstep emulator of the wave primitives in
// predpreygrass_amd/csrc/wave.h, so that the *same kernel source* can be compiled with
// g++ and debugged on a machine without a GPU (asan/ubsan/gdb).  It is never part of the
// product: predpreygrass_amd loads libppg_hip.so only and fails loudly without it.
//
// Model: the 64 lanes of each wavefront of a workgroup are cooperative fibers (64 x nwaves of them).  A fiber runs until its
// next wave primitive (a "collective"), publishes its operand and yields; when all 64
// have arrived the scheduler resumes them -- in a RANDOM order each round, so code that
// silently relies on lane execution order or on which lane wins a same-address LDS
// write fails here.  The emulator is stricter than hardware about LDS visibility:
// a value written to LDS by one lane is only guaranteed visible to the others after a
// collective, which is exactly the discipline the kernels document.
// A workgroup of several wavefronts (the multi-wave step kernels) runs as nwaves x 64 fibers: wave collectives involve
// the 64 lanes of one wave only; wg_barrier() parks a wave until every wave that has not finished has arrived.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PPG_WAVE_EMU 1
// AddressSanitizer build (tests/emu_backend.py: build(sanitize="address")): the bytes behind a workgroup's LDS are POISONED, so a
// read past the allocation traps as well (the 0x5C pattern below only catches writes, and only after the fact)
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#define PPG_EMU_POISON(p, n) __asan_poison_memory_region((p), (n))
#define PPG_EMU_UNPOISON(p, n) __asan_unpoison_memory_region((p), (n))
#else
#define PPG_EMU_POISON(p, n) do { } while (0)
#define PPG_EMU_UNPOISON(p, n) do { } while (0)
#endif
#define PPG_DEVICE static inline
#define PPG_MEMBER inline
#define PPG_HOST_DEVICE static inline
#define PPG_KERNEL(name, W) static void name
#define PPG_DYNAMIC_LDS(name) unsigned char *name = wv::emu().lds
#define PPG_BLOCK_INDEX() (wv::emu().block)
#define PPG_CONSTANT_AS
#define PPG_KERNARG_PTR(T, byval) (&(byval))
#define PPG_LAUNDER_S(x) do { } while (0)
#define PPG_LAUNDER_V(x) do { } while (0)
#define __restrict__

namespace wv {

constexpr int W = 64;
constexpr int MAXW = 16;          // wavefronts per workgroup
constexpr int NF = W * MAXW;      // fibers

extern "C" void ppg_emu_ctx_switch(void **save_sp, void *load_sp);

struct Emu {
    void *main_sp = nullptr;
    void *fiber_sp[NF];
    unsigned char *stacks = nullptr;
    bool done[NF];
    bool at_barrier[NF];
    uint64_t phase[NF];
    uint64_t slot[2][NF];
    int cur = 0;
    int nwaves = 1;
    int block = 0;
    unsigned char *lds = nullptr;
    size_t lds_bytes = 0;
    void (*entry)(void *) = nullptr;
    void *arg = nullptr;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    uint64_t n_collectives = 0;
};

inline Emu &emu() {
    static Emu e;
    return e;
}

static void fiber_main() {
    Emu &e = emu();
    e.entry(e.arg);
    e.done[e.cur] = true;
    ppg_emu_ctx_switch(&e.fiber_sp[e.cur], e.main_sp);
    abort();  // never resumed
}

// Publish `mine`, wait for the whole wave, return the 64 operands published by the lanes of this wave.
inline const uint64_t *exchange(uint64_t mine) {
    Emu &e = emu();
    int me = e.cur;
    uint64_t ph = e.phase[me]++;
    e.slot[ph & 1][me] = mine;
    ppg_emu_ctx_switch(&e.fiber_sp[me], e.main_sp);
    e.cur = me;
    return e.slot[ph & 1] + (me & ~(W - 1));
}

constexpr size_t STACK_BYTES = 512 * 1024;

// Run one workgroup (nwaves x 64 lanes) of `entry(arg)` with `lds_bytes` of shared memory.
inline void run_block(void (*entry)(void *), void *arg, int block, size_t lds_bytes, int nwaves = 1) {
    Emu &e = emu();
    if (nwaves < 1 || nwaves > MAXW) { fprintf(stderr, "wave_emu: %d waves per workgroup\n", nwaves); abort(); }
    if (!e.stacks) e.stacks = (unsigned char *)aligned_alloc(64, STACK_BYTES * NF);  // (untouched pages stay virtual)
    constexpr size_t REDZONE = 4096;   // behind the workgroup's LDS: a write there is a kernel writing past its allocation
    if (e.lds) PPG_EMU_UNPOISON(e.lds, e.lds_bytes);
    if (e.lds_bytes < lds_bytes + REDZONE) {
        free(e.lds);
        e.lds = (unsigned char *)aligned_alloc(64, (lds_bytes + REDZONE + 63) / 64 * 64);
        e.lds_bytes = lds_bytes + REDZONE;
    }
    // LDS content is undefined at launch on hardware: poison it.
    memset(e.lds, 0xA5, lds_bytes);
    memset(e.lds + lds_bytes, 0x5C, REDZONE);
    PPG_EMU_POISON(e.lds + lds_bytes, e.lds_bytes - lds_bytes);
    e.entry = entry;
    e.arg = arg;
    e.block = block;
    e.nwaves = nwaves;
    const int nf = nwaves * W;
    for (int l = 0; l < nf; ++l) {
        e.done[l] = false;
        e.at_barrier[l] = false;
        e.phase[l] = 0;
        uint64_t *top = (uint64_t *)(e.stacks + STACK_BYTES * (l + 1));
        top[-1] = 0;                       // fake return address of fiber_main
        top[-2] = (uint64_t)&fiber_main;   // `ret` target of the first switch
        for (int k = 3; k <= 8; ++k) top[-k] = 0;  // rbp rbx r12 r13 r14 r15
        e.fiber_sp[l] = (void *)(top - 8);
    }
    static int order[NF];
    for (;;) {
        int n = 0, live = 0;
        for (int l = 0; l < nf; ++l) {
            if (e.done[l]) continue;
            live++;
            if (!e.at_barrier[l]) order[n++] = l;
        }
        if (live == 0) break;
        if (n == 0) {  // every wave that is still running waits at the workgroup barrier: release them
            for (int l = 0; l < nf; ++l) e.at_barrier[l] = false;
            continue;
        }
        for (int i = n - 1; i > 0; --i) {  // random resume order (across lanes AND waves)
            e.rng ^= e.rng << 13; e.rng ^= e.rng >> 7; e.rng ^= e.rng << 17;
            int j = (int)(e.rng % (uint64_t)(i + 1));
            int t = order[i]; order[i] = order[j]; order[j] = t;
        }
        for (int i = 0; i < n; ++i) {
            e.cur = order[i];
            ppg_emu_ctx_switch(&e.main_sp, e.fiber_sp[order[i]]);
        }
        // the lanes of a wave must now sit at the same collective / the barrier, or all be done
        for (int w = 0; w < nwaves; ++w) {
            int nd = 0, nb = 0;
            uint64_t ph = 0;
            bool have = false, bad = false;
            for (int l = w * W; l < (w + 1) * W; ++l) {
                if (e.done[l]) { nd++; continue; }
                if (e.at_barrier[l]) nb++;
                if (!have) { ph = e.phase[l]; have = true; }
                else if (e.phase[l] != ph) bad = true;
            }
            if (bad || (nd != 0 && nd != W) || (nb != 0 && nb != W - nd)) {
                fprintf(stderr, "wave_emu: divergent collective in block %d wave %d (done=%d, at barrier=%d)\n", block, w, nd, nb);
                abort();
            }
        }
        e.n_collectives++;
    }
    PPG_EMU_UNPOISON(e.lds + lds_bytes, e.lds_bytes - lds_bytes);
    for (size_t i = 0; i < REDZONE; ++i)
        if (e.lds[lds_bytes + i] != 0x5C) {
            fprintf(stderr, "wave_emu: block %d wrote %zu bytes past its %zu bytes of LDS\n", block, i + 1, lds_bytes);
            abort();
        }
}

// ---- the primitives of wave.h ----
inline int lane() { return emu().cur & (W - 1); }

inline uint64_t ballot(bool p) {
    const uint64_t *s = exchange(p ? 1 : 0);
    uint64_t m = 0;
    for (int l = 0; l < W; ++l) m |= (s[l] & 1) << l;
    return m;
}
inline uint32_t readlane(uint32_t v, int k) {
    const uint64_t *s = exchange(((uint64_t)(uint32_t)k << 32) | v);
    for (int l = 0; l < W; ++l)
        if ((int)(s[l] >> 32) != k) { fprintf(stderr, "wave_emu: readlane index not uniform\n"); abort(); }
    if (k < 0 || k >= W) { fprintf(stderr, "wave_emu: readlane index %d\n", k); abort(); }
    return (uint32_t)s[k];
}
inline uint32_t first(uint32_t v) {
    const uint64_t *s = exchange(v);
    // callers use first() on values they claim are uniform: verify the claim
    for (int l = 1; l < W; ++l)
        if ((uint32_t)s[l] != (uint32_t)s[0]) { fprintf(stderr, "wave_emu: first() on non-uniform value\n"); abort(); }
    return (uint32_t)s[0];
}
inline uint32_t writelane(uint32_t v, int k, uint32_t sval) {
    if (k < 0 || k >= W) { fprintf(stderr, "wave_emu: writelane index %d\n", k); abort(); }
    return lane() == k ? sval : v;
}
inline uint32_t prefix(uint64_t mask) {
    int l = lane();
    return (uint32_t)__builtin_popcountll(l ? (mask & (~0ull >> (64 - l))) : 0ull);
}
inline uint32_t shfl_up1(uint32_t v) {
    const uint64_t *s = exchange(v);
    int l = lane();
    return (uint32_t)s[l ? l - 1 : 0];
}
inline double shfl_xor_f64(double v, int mask) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const uint64_t *s = exchange(bits);
    const uint64_t o = s[lane() ^ mask];
    double r;
    memcpy(&r, &o, 8);
    return r;
}
inline int wave_index() { return emu().cur / W; }
// workgroup barrier: the fiber parks until every fiber of the workgroup that has not finished is parked too (a wave that
// has ended no longer takes part, as on hardware)
inline void wg_barrier() {
    Emu &e = emu();
    int me = e.cur;
    e.at_barrier[me] = true;
    ppg_emu_ctx_switch(&e.fiber_sp[me], e.main_sp);
    e.cur = me;
}
inline void wg_barrier_lds() { wg_barrier(); }
inline void sync() { (void)exchange(0); }
// (lanes are fibers that only switch at collectives: a plain increment is atomic here)
inline void lds_count(uint8_t *p) { *p = (uint8_t)(*p + 1); }
inline void lds_count(uint16_t *p) { *p = (uint16_t)(*p + 1); }
// hand-over between the wavefronts of a workgroup through LDS words.  Lanes run one after another here, so lane 0's read-modify-write
// is atomic; the collectives make the result wave-uniform and let the other wavefronts run while this one polls.
inline uint32_t lds_take_issue(uint32_t *p, uint32_t n) { uint32_t old = 0; if (lane() == 0) { old = *p; *p = old + n; } return old; }
inline uint32_t lds_take_value(uint32_t issued) { return readlane(issued, 0); }
inline void lds_or(uint32_t *p, uint32_t bits) { (void)exchange(0); if (lane() == 0) *p |= bits; }   // (every lane's LDS writes first)
inline uint32_t lds_poll(const uint32_t *p) { return readlane(*(const volatile uint32_t *)p, 0); }
inline void poll_sleep() { }
inline void drain_loads() { (void)exchange(0); }  // lanes run one after another here: a collective orders reads before writes
inline uint32_t mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline uint32_t mul24(uint32_t a, uint32_t b) {
    if ((a | b) >> 24) { fprintf(stderr, "wave_emu: mul24 operand of 24 bits or more\n"); abort(); }
    return a * b;
}
inline int popc(uint64_t m) { return __builtin_popcountll(m); }
inline int ctz(uint64_t m) { return __builtin_ctzll(m); }

}  // namespace wv

struct double2 { double x, y; };
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }
