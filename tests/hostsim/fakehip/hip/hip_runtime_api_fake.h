// TEST-ONLY stand-in for the HIP RUNTIME API (streams, events, memory, launches), on top of the fiber emulation of the kernel
// language (hip_runtime.h): lets g++ compile gubernator_amd/csrc/guber_engine.hip — the engine's HOST code with its kernels — into a
// library that runs on a machine without a GPU (tests/hostsim/enginesim.cpp, tests/test_enginesim_cpu.py).  Every "device" pointer
// is a host pointer, every copy a memcpy, every launch runs AT ONCE on the calling thread (one launch at a time, process-wide), so
// stream order = call order and every synchronisation returns immediately.  It checks the engine's host logic — which launches, in
// which order, with which arguments — and the kernels' logic; it cannot see races, memory ordering or performance.
// Included by hip_runtime.h when FAKEHIP_RUNTIME is defined.  Nothing in the product includes this file.
#pragma once
#include <chrono>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
struct fh_stream_s { int id; };
struct fh_event_s { double t_ms; };
typedef fh_stream_s* hipStream_t;
typedef fh_event_s* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum : unsigned { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
#define HIP_SYMBOL(x) (&(x))

namespace fakehip {
inline std::recursive_mutex& launch_mutex() { static std::recursive_mutex m; return m; }
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline unsigned long long& launches() { static unsigned long long n = 0; return n; }
}  // namespace fakehip

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fakehip error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t fh_alloc(void** p, size_t n) { *p = nullptr; if (posix_memalign(p, 4096, n ? n : 1)) return hipErrorOutOfMemory; return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return fh_alloc((void**)p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return fh_alloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void* d, size_t pitch, int v, size_t width, size_t height, hipStream_t = nullptr) {
    for (size_t r = 0; r < height; ++r) memset((char*)d + r * pitch, v, width);
    return hipSuccess;
}
static inline hipError_t hipMemcpyToSymbol(void* symbol, const void* src, size_t n) { memcpy(symbol, src, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new fh_stream_s{1}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new fh_event_s{0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { if (e) e->t_ms = fakehip::now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }

// hipLaunchKernelGGL(kernel, grid, block, dynamic LDS, stream, args...): the arguments are converted to the kernel's parameter
// types and kept in one object; a kernel that reads its argument block through __builtin_amdgcn_kernarg_segment_ptr() has exactly
// one (struct) parameter, so the block is that parameter
namespace fakehip {
template <class... KArgs, class... Args> void launch_kernel(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "hipLaunchKernelGGL: argument count");
    std::tuple<std::decay_t<KArgs>...> packed(std::forward<Args>(args)...);
    const void* kernarg = nullptr;
    if constexpr (sizeof...(KArgs) >= 1) kernarg = &std::get<0>(packed);
    std::lock_guard<std::recursive_mutex> lk(launch_mutex());
    launches()++;
    launch(grid, block, kernarg, [&] { std::apply(kernel, packed); });
}
}  // namespace fakehip
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) fakehip::launch_kernel((kernel), dim3(grid), dim3(block), ##__VA_ARGS__)
