// hbm_peak.hip — measured HBM bandwidth of the box ("confirm on the box with a device copy / triad
// kernel and report both spec-peak and measured-peak fractions"), plus the random-access ceiling that actually bounds
// the rate-limit path: 64-byte and 128-byte gathers over a 4 GiB array.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_peak hbm_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_read(const uint4* __restrict__ a, size_t n, uint4* out) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;   // never true for the fill pattern: keeps the loads
}
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_triad(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i] + 3.0 * b[i];
}
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// every thread gathers `per` random lines of `line16` x 16 bytes
template <int LINE16>
__global__ void k_gather(const uint4* __restrict__ a, size_t lines, int per, uint64_t seed, uint4* out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (int j = 0; j < per; ++j) {
        const size_t line = mix(t * 1000003ull + j + seed) % lines;
#pragma unroll
        for (int q = 0; q < LINE16; ++q) { const uint4 v = a[line * LINE16 + q]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;
}

int main() {
    const size_t bytes = 4ull << 30;
    uint4 *a, *b, *c, *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n16 = bytes / 16, n8 = bytes / 8;
    auto timeit = [&](auto&& launch) {
        std::vector<float> ts;
        for (int it = 0; it < 7; ++it) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    };
    const int blocks = 256 * 16;
    float ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n16, out); });
    printf("read   4 GiB sequential : %8.3f ms  %8.1f GB/s\n", ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n16); });
    printf("copy   4 GiB -> 4 GiB   : %8.3f ms  %8.1f GB/s (read + write)\n", ms, 2.0 * bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, 0, (const double*)a, (const double*)b, (double*)c, n8); });
    printf("triad  c = a + 3b       : %8.3f ms  %8.1f GB/s (2 reads + 1 write)\n", ms, 3.0 * bytes / ms / 1e6);
    // random gathers: 16 M threads x 1 line (deep queue) and 65 536 threads x 1 line (the batch shape of the rate-limit path)
    for (int shape = 0; shape < 2; ++shape) {
        const size_t threads = shape == 0 ? (16ull << 20) : 65536;
        const int per = 1;
        ms = timeit([&] { hipLaunchKernelGGL(k_gather<4>, dim3(threads / 256), dim3(256), 0, 0, a, bytes / 64, per, 12345ull, out); });
        printf("gather 64 B  x %9zu  : %8.3f ms  %8.1f GB/s  %8.2f G lines/s\n", threads, ms, threads * 64.0 / ms / 1e6, threads / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_gather<8>, dim3(threads / 256), dim3(256), 0, 0, a, bytes / 128, per, 777ull, out); });
        printf("gather 128 B x %9zu  : %8.3f ms  %8.1f GB/s  %8.2f G lines/s\n", threads, ms, threads * 128.0 / ms / 1e6, threads / ms / 1e6);
    }
    return 0;
}
