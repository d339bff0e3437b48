// fabric_peak.hip — transaction-rate ceilings of one MI355X for the access kinds the rate-limit pipeline issues:
// random 64-byte reads / writes into arrays of different sizes (HBM- vs die-cache-resident), device-scope atomics on random
// and on few addresses, device-scope ("look") loads.  Complements tools/hbm_peak.hip (streaming ceilings).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fabric_peak tools/fabric_peak.hip && /tmp/fabric_peak
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// kind 0: plain 16-byte load of a random 64-B line; 1: device-scope 8-byte load; 2: 16-byte store; 3: atomicAdd without
// return; 4: atomicAdd with return (consumed); 5: atomicCAS with return
template <int KIND>
__global__ void k_rand(unsigned long long* a, size_t lines, int per, uint64_t seed, unsigned long long* out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (int j = 0; j < per; ++j) {
        const size_t line = mix(t * 1000003ull + j + seed) % lines;
        unsigned long long* p = a + line * 8;
        if (KIND == 0) { const ulonglong2 v = *(const ulonglong2*)p; acc ^= v.x ^ v.y; }
        else if (KIND == 1) acc ^= __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (KIND == 2) *(ulonglong2*)p = make_ulonglong2(t, j);
        else if (KIND == 3) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (KIND == 4) acc ^= __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else { unsigned long long exp = 0ull; __hip_atomic_compare_exchange_strong(p, &exp, t, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc ^= exp; }
    }
    if (acc == 0x123456789abcdef0ull) out[0] = acc;
}

int main() {
    const size_t max_bytes = 4ull << 30;
    unsigned long long *a, *out;
    CK(hipMalloc(&a, max_bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, max_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& launch) {
        std::vector<float> ts;
        for (int it = 0; it < 7; ++it) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    };
    const char* names[6] = {"plain 16-B load", "agent-scope 8-B load", "16-B store", "atomicAdd no return", "atomicAdd returning", "atomicCAS returning"};
    const size_t sizes[5] = {2ull << 20, 32ull << 20, 128ull << 20, 1ull << 30, 4ull << 30};
    printf("# random 64-B-line transactions, 4 M threads x 4 accesses (deep queues); G transactions/s by array size\n");
    printf("%-22s", "kind \\ array");
    for (size_t s : sizes) printf(" %8zu MiB", s >> 20);
    printf("\n");
    for (int kind = 0; kind < 6; ++kind) {
        printf("%-22s", names[kind]);
        for (size_t s : sizes) {
            const size_t threads = 4ull << 20; const int per = 4;
            const size_t lines = s / 64;
            float ms = 0;
            auto go = [&](auto kern) { ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, a, lines, per, 4242ull, out); }); };
            switch (kind) {
                case 0: go(k_rand<0>); break; case 1: go(k_rand<1>); break; case 2: go(k_rand<2>); break;
                case 3: go(k_rand<3>); break; case 4: go(k_rand<4>); break; default: go(k_rand<5>); break;
            }
            printf(" %12.2f", threads * per / ms / 1e6);
        }
        printf("\n");
    }
    printf("# the batch shape: 65 536 threads x 1 access, one launch (us per launch = latency of one dependent trip incl. launch)\n");
    printf("%-22s", "kind \\ array");
    for (size_t s : sizes) printf(" %8zu MiB", s >> 20);
    printf("\n");
    for (int kind = 0; kind < 6; ++kind) {
        printf("%-22s", names[kind]);
        for (size_t s : sizes) {
            const size_t threads = 65536; const int per = 1;
            const size_t lines = s / 64;
            float ms = 0;
            auto go = [&](auto kern) { ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, a, lines, per, 999ull, out); }); };
            switch (kind) {
                case 0: go(k_rand<0>); break; case 1: go(k_rand<1>); break; case 2: go(k_rand<2>); break;
                case 3: go(k_rand<3>); break; case 4: go(k_rand<4>); break; default: go(k_rand<5>); break;
            }
            printf(" %9.2f us", ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
