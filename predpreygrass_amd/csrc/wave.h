// wave.h -- the CDNA4 wavefront primitives the PredPreyGrass kernels are written against.
//
// One environment is stepped by ONE 64-lane wavefront (workgroup = 1 wave), so every
// cross-lane exchange is a wave intrinsic and `sync()` is only an LDS ordering point
// (s_waitcnt lgkmcnt(0); the s_barrier of a single-wave workgroup is free).
//
// Rule the kernels follow: these functions are only called from wave-uniform control
// flow (all 64 lanes reach the call).  Values read from LDS that steer control flow are
// made scalar with first() so branches stay on the scalar unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PPG_DEVICE __device__ __forceinline__
#define PPG_MEMBER __device__ __forceinline__
#define PPG_HOST_DEVICE __host__ __device__ inline
// 64 threads = one wavefront per workgroup; W = waves per SIMD the register allocator must allow
// (4 -> <= 128 VGPRs -> 16 waves per CU = 4096 co-resident envs per MI355X)
#define PPG_KERNEL(name, W) extern "C" __global__ void __launch_bounds__(64, W) name
// NW wavefronts per workgroup: wave 0 steps the env, all NW waves write the final observations (small batches, large grids)
#define PPG_KERNEL_NW(name, W, NW) extern "C" __global__ void __launch_bounds__(64 * NW, W) name
#define PPG_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define PPG_BLOCK_INDEX() ((int)blockIdx.x)
// kernel parameters read in place from the kernarg segment (constant address space -> s_load)
#define PPG_CONSTANT_AS __attribute__((address_space(4)))
#define PPG_KERNARG_PTR(T, byval) ((const PPG_CONSTANT_AS T *)__builtin_amdgcn_kernarg_segment_ptr())
// make a value opaque to the optimiser (no instruction is emitted)
#define PPG_LAUNDER_S(x) __asm__ volatile("" : "+s"(x))
#define PPG_LAUNDER_V(x) __asm__ volatile("" : "+v"(x))

namespace wv {

PPG_DEVICE int lane() { return (int)(threadIdx.x & 63u); }
// index of this wavefront within its workgroup (0 unless the kernel is a multi-wave variant), as a scalar
PPG_DEVICE int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// workgroup barrier of the multi-wave variants (LDS writes of every wave visible afterwards)
PPG_DEVICE void wg_barrier() { __syncthreads(); }
// the same without waiting for this wave's outstanding global stores / loads (__syncthreads drains vmcnt too): only LDS traffic is
// ordered across the barrier -- for hand-overs that go through LDS alone
PPG_DEVICE void wg_barrier_lds() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 64-bit mask of lanes whose predicate is true (s_* result: lives in SGPRs).
PPG_DEVICE uint64_t ballot(bool p) { return __ballot(p); }

// v_readlane_b32: value of lane k (k wave-uniform) as a scalar.
PPG_DEVICE uint32_t readlane(uint32_t v, int k) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, k);
}
// v_readfirstlane_b32: makes a value known to be wave-uniform scalar.
PPG_DEVICE uint32_t first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// v with lane k replaced by the scalar s (v_cmp + v_cndmask; this clang has no writelane builtin).
PPG_DEVICE uint32_t writelane(uint32_t v, int k, uint32_t s) { return lane() == k ? s : v; }

// number of set bits of `mask` below this lane (v_mbcnt_lo/hi): exclusive prefix count.
PPG_DEVICE uint32_t prefix(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// value held by lane-1 (lane 0 gets its own).
PPG_DEVICE uint32_t shfl_up1(uint32_t v) { return (uint32_t)__shfl_up((int)v, 1, 64); }

// value held by lane ^ mask, for a float64 (two 32-bit exchanges)
PPG_DEVICE double shfl_xor_f64(double v, int mask) {
    long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, mask, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)b >> 32), mask, 64);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// LDS ordering point between lanes of the wave.  A workgroup is ONE wavefront and the LDS unit
// executes a wave's DS instructions in issue order, so no s_barrier and no counter drain is
// needed -- only the compiler must not move LDS accesses across this point.  (__syncthreads()
// would also drain vmcnt(0), i.e. wait for every outstanding observation store to reach HBM.)
PPG_DEVICE void sync() { __asm__ volatile("" ::: "memory"); }

// Count one more on an 8- or 16-bit LDS counter that other lanes may be counting on too (ds_add_u32 on the word that holds it).
// A counter must never reach its field's width -- the add would carry into the neighbouring cell's counter, which the CPU wave
// emulator (a plain increment of the field) cannot reproduce.  The one caller (Env::move) counts, per species channel, the live
// agents standing on a cell plus the agents that want to move INTO it: disjoint sets of rows of one species, so at most its row
// capacity -- <= 128 on 8-bit maps (Env::MAP8 = at most two prey row registers), <= 256 on 16-bit maps.
PPG_DEVICE void lds_count(uint8_t *p) {
    const uint32_t low = (uint32_t)(uintptr_t)p & 3u;   // (the word's address stays derived from p: the access stays a DS instruction)
    __hip_atomic_fetch_add((uint32_t *)(p - low), 1u << (8u * low), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
PPG_DEVICE void lds_count(uint16_t *p) {
    const uint32_t low = (uint32_t)(uintptr_t)p & 2u;
    __hip_atomic_fetch_add((uint32_t *)((unsigned char *)p - low), 1u << (8u * low), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- hand-over between the wavefronts of ONE workgroup through LDS words (the cooperative kernels' dynamic write phase) ----
// Fetch-add on an LDS word for the whole wavefront: lane 0 adds (ds_add_rtn_u32), lds_take_value() hands its old value to every lane.
// Two calls so that the round trip of the NEXT ticket hides behind the work on the ticket in hand.
PPG_DEVICE uint32_t lds_take_issue(uint32_t *p, uint32_t n) {
    uint32_t old = 0;
    if (lane() == 0) old = __hip_atomic_fetch_add(p, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return old;
}
PPG_DEVICE uint32_t lds_take_value(uint32_t issued) { return readlane(issued, 0); }
// set bits of an LDS word AFTER everything this wavefront wrote to LDS before (DS instructions of a wave execute in issue order; the
// s_waitcnt only keeps the compiler and the counters honest)
PPG_DEVICE void lds_or(uint32_t *p, uint32_t bits) {
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane() == 0) __hip_atomic_fetch_or(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// an LDS word another wavefront may set, as a scalar; LDS reads behind it are not moved in front of it
PPG_DEVICE uint32_t lds_poll(const uint32_t *p) {
    const uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __asm__ volatile("" ::: "memory");
    return first(v);
}
#ifndef PPG_COOP_POLL_SLEEP
#define PPG_COOP_POLL_SLEEP 8
#endif
PPG_DEVICE void poll_sleep() { __builtin_amdgcn_s_sleep(PPG_COOP_POLL_SLEEP); }

// all of this wave's outstanding global loads have returned (used before overwriting memory other lanes just read)
PPG_DEVICE void drain_loads() { __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

PPG_DEVICE uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
// low 32 bits of the product of two values below 2^24 (v_mul_u32_u24: full rate, where v_mul_lo_u32 runs at a quarter of it)
PPG_DEVICE uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
PPG_DEVICE int popc(uint64_t m) { return __popcll(m); }
PPG_DEVICE int ctz(uint64_t m) { return __ffsll((long long)m) - 1; }  // m != 0

}  // namespace wv
