// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets g++ compile the engine's kernel headers (gubernator_amd/csrc/guber_kernels*.h)
// for the host, so that the real kernel source can be run — one workgroup at a time, every "thread" a cooperative fiber — against
// the oracle on a machine without a GPU (tests/hostsim/devsim.cpp, tests/test_kernels_devsim.py).
// What it models: threadIdx / blockIdx, __shared__ (static storage: one workgroup runs at a time), workgroup barriers, the wave64
// cross-lane operations (a rendezvous of the wave's live lanes), atomics (fibers are cooperative, so plain read-modify-write).
// What it cannot model: memory ordering, races between workgroups, performance.  Nothing in the product includes this file.
#pragma once
#include <stdint.h>
#include <string.h>
#include <ucontext.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
// a launch's dynamic LDS (gubernator_amd/csrc/guber_table.h GUBER_DYN_LDS): one workgroup runs at a time, so one static buffer serves
namespace fakehip { inline unsigned char* dyn_lds() { alignas(16) static unsigned char buf[160 * 1024]; return buf; } }
#define GUBER_DYN_LDS(name) unsigned char* const name = fakehip::dyn_lds()
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0

struct alignas(8) uint2 { uint32_t x, y; };
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
struct uint4 { uint32_t x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long a, unsigned long long b) { return ulonglong2{a, b}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
struct dim3 { uint32_t x, y, z; dim3(uint32_t a = 1, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {} };

namespace fakehip {
constexpr int kWave = 64;
#if defined(__x86_64__)
struct Fiber { void* sp; bool done; int waiting; /* 0 run, 1 barrier, 2 wave op */ };       // (fiber_runtime.h: fh_switch)
#else
struct Fiber { ucontext_t ctx; char* stack; bool done; int waiting; /* 0 run, 1 barrier, 2 wave op */ };
#endif
struct State {
    dim3 tidx, bidx, bdim, gdim;
    std::vector<Fiber> fib; int cur = -1; ucontext_t sched;
    int bar_waiting = 0; unsigned long long bar_gen = 0;
    // wave exchange: two alternating buffers per wave (a lane can be at most one operation ahead of its wave)
    unsigned long long wx[16][2][kWave]; int warrived[16][2]; unsigned long long wgen[16]; unsigned long long lane_gen[1024];
    const void* kernarg = nullptr;
    uint32_t chaos = 0;   // != 0: yield inside atomics in a seeded pseudo-random pattern
    uint32_t block_order = 0;   // the order a launch's workgroups run in: 0 ascending, 1 descending, 2 a seeded shuffle
    uint64_t rng = 88172645463325252ull;
    unsigned long long progress = 0;   // bumped whenever a fiber gets past a wait, ends, or yields voluntarily (deadlock detection)
};
extern State S;
void yield();
void barrier();
unsigned long long wave_exchange(unsigned long long v, int src_lane_or_minus1, unsigned long long* all /* [64] or null */, unsigned long long* live_mask);
template <class F> void launch(dim3 grid, dim3 block, const void* kernarg, F body);
inline void maybe_chaos() {
    if (!S.chaos) return;
    S.rng ^= S.rng << 13; S.rng ^= S.rng >> 7; S.rng ^= S.rng << 17;
    if ((S.rng & 7) == 0) { S.progress++; yield(); }
}
}  // namespace fakehip

#define threadIdx (fakehip::S.tidx)
#define blockIdx (fakehip::S.bidx)
#define blockDim (fakehip::S.bdim)
#define gridDim (fakehip::S.gdim)

static inline void __syncthreads() { fakehip::barrier(); }
#define __builtin_amdgcn_s_barrier() fakehip::barrier()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_kernarg_segment_ptr() (fakehip::S.kernarg)
static inline void __threadfence_system() {}
static inline void __threadfence() {}
static inline unsigned long long wall_clock64() { return 0; }

// ---- atomics: cooperative fibers, so a plain read-modify-write is atomic ----
template <class T> static inline T fh_cas(T* p, T cmp, T val) { fakehip::maybe_chaos(); T old = *p; if (old == cmp) *p = val; return old; }
template <class T> static inline T fh_add(T* p, T v) { fakehip::maybe_chaos(); T old = *p; *p = (T)(old + v); return old; }
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long c, unsigned long long v) { return fh_cas(p, c, v); }
static inline unsigned int atomicCAS(unsigned int* p, unsigned int c, unsigned int v) { return fh_cas(p, c, v); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return fh_add(p, v); }
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { return fh_add(p, v); }
static inline int atomicAdd(int* p, int v) { return fh_add(p, v); }
static inline int atomicExch(int* p, int v) { fakehip::maybe_chaos(); auto o = *p; *p = v; return o; }
static inline unsigned int atomicExch(unsigned int* p, unsigned int v) { fakehip::maybe_chaos(); auto o = *p; *p = v; return o; }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { fakehip::maybe_chaos(); auto o = *p; *p = o | v; return o; }
static inline unsigned int atomicOr(unsigned int* p, unsigned int v) { fakehip::maybe_chaos(); auto o = *p; *p = o | v; return o; }
static inline long long atomicMin(long long* p, long long v) { fakehip::maybe_chaos(); auto o = *p; if (v < o) *p = v; return o; }
static inline long long atomicMax(long long* p, long long v) { fakehip::maybe_chaos(); auto o = *p; if (v > o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { fakehip::maybe_chaos(); auto o = *p; if (v < o) *p = v; return o; }
static inline int atomicMax(int* p, int v) { fakehip::maybe_chaos(); auto o = *p; if (v > o) *p = v; return o; }
static inline unsigned int atomicMin(unsigned int* p, unsigned int v) { fakehip::maybe_chaos(); auto o = *p; if (v < o) *p = v; return o; }
static inline unsigned int atomicMax(unsigned int* p, unsigned int v) { fakehip::maybe_chaos(); auto o = *p; if (v > o) *p = v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) fh_add((p), (decltype(*(p) + 0))(v))

// ---- bit tricks ----
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned int x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

// ---- wave64 cross-lane operations: every live lane of the wave must reach the call (a lane that is in another branch deadlocks
// the emulation, which is reported — on the hardware such a call would silently see a partial wave) ----
template <class T> static inline T __shfl(T v, int src, int = 64) {
    unsigned long long all[fakehip::kWave]; unsigned long long live;
    unsigned long long bits = 0; memcpy(&bits, &v, sizeof(T));
    fakehip::wave_exchange(bits, -1, all, &live);
    T out{}; const unsigned long long r = all[src & 63]; memcpy(&out, &r, sizeof(T)); return out;
}
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return __shfl(v, (int)((threadIdx.x & 63) ^ (unsigned)mask)); }
template <class T> static inline T __shfl_up(T v, unsigned d, int = 64) { const int l = (int)(threadIdx.x & 63); return __shfl(v, l >= (int)d ? l - (int)d : l); }
template <class T> static inline T __shfl_down(T v, unsigned d, int = 64) { const int l = (int)(threadIdx.x & 63); return __shfl(v, l + (int)d < 64 ? l + (int)d : l); }
static inline unsigned long long __ballot(int pred) {
    unsigned long long all[fakehip::kWave]; unsigned long long live;
    fakehip::wave_exchange(pred ? 1ull : 0ull, -1, all, &live);
    unsigned long long m = 0; for (int i = 0; i < 64; ++i) if (((live >> i) & 1) && all[i]) m |= 1ull << i;
    return m;
}

#ifdef FAKEHIP_RUNTIME   // the runtime API as well (streams, events, memory, launches): the engine's host code on the CPU
#include "hip_runtime_api_fake.h"
#endif
