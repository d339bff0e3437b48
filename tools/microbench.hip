// microbench.hip — primitive costs behind k_front on MI355X: random directory loads, dependent bucket
// loads, claim CAS, for different workgroup shapes.  Build: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// mode bits: 1 = dir load (16 B, sc1), 2 = dependent bucket load (128 B), 4 = CAS on dir meta, 8 = atomicOr on a random u64 in a third array
// 16 = plain (L1) dir load instead of sc1, 32 = all lanes same index ("hot key")
__global__ void k(int mode, const ulonglong2* dir, uint64_t dmask, const uint4* bk, unsigned long long* third, uint64_t tmask, uint64_t* out, uint32_t n, uint32_t epoch) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = mix(i * 0x9E3779B97F4A7C15ULL + epoch * 7919ULL);
    if (mode & 32) h = mix((i & 7) + 12345);
    uint64_t pos = h & dmask;
    uint64_t acc = 0;
    unsigned long long t = 0, m = 0;
    if (mode & 1) {
        if (mode & 16) { ulonglong2 e = dir[pos]; t = e.x; m = e.y; }
        else { t = __hip_atomic_load((const unsigned long long*)&dir[pos].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               m = __hip_atomic_load((const unsigned long long*)&dir[pos].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        acc += t + m;
    }
    uint64_t p2 = (mode & 1) ? ((pos + (t & 1)) & dmask) : pos;   // dependent on the loaded value
    if (mode & 2) {
        const uint4* b = bk + p2 * 8;
        uint4 a0 = b[0], a1 = b[1], a2 = b[2], a3 = b[3], a4 = b[4], a5 = b[5], a6 = b[6], a7 = b[7];
        acc += a0.x + a1.y + a2.z + a3.w + a4.x + a5.y + a6.z + a7.w;
    }
    if (mode & 4) {
        unsigned long long want = ((unsigned long long)epoch << 32) | i;
        unsigned long long old = atomicCAS((unsigned long long*)&dir[p2].y, m, want);
        acc += old;
    }
    if (mode & 8) {
        atomicOr(&third[(h >> 20) & tmask], 1ull << (i & 63));
    }
    out[i] = acc;
}

int main() {
    const uint64_t slots = 1ull << 25;
    ulonglong2* dir; uint4* bk; unsigned long long* third; uint64_t* out;
    CK(hipMalloc(&dir, slots * 16)); CK(hipMalloc(&bk, slots * 128)); CK(hipMalloc(&third, (1ull << 17) * 8)); CK(hipMalloc(&out, 65536 * 8));
    CK(hipMemset(dir, 0, slots * 16)); CK(hipMemset(bk, 1, slots * 128)); CK(hipMemset(third, 0, (1ull << 17) * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Cfg { const char* name; int mode; int bs; uint64_t dslots; };
    std::vector<Cfg> cfgs = {
        {"empty kernel                      1024", 0, 1024, slots},
        {"empty kernel                       256", 0, 256, slots},
        {"dir16 sc1                         1024", 1, 1024, slots},
        {"dir16 sc1                          256", 1, 256, slots},
        {"dir16 plain                       1024", 1 | 16, 1024, slots},
        {"dir16 sc1 small table (1M slots)  1024", 1, 1024, 1 << 20},
        {"dir16 + bucket128                 1024", 3, 1024, slots},
        {"dir16 + bucket128                  256", 3, 256, slots},
        {"bucket128 only                    1024", 2, 1024, slots},
        {"dir16 + CAS                       1024", 5, 1024, slots},
        {"dir16 + CAS                        256", 5, 256, slots},
        {"CAS only (no load)                1024", 4, 1024, slots},
        {"dir16 + bucket128 + CAS           1024", 7, 1024, slots},
        {"dir16 + bucket128 + CAS            256", 7, 256, slots},
        {"dir16+bucket+CAS+atomicOr(1MB arr)1024", 15, 1024, slots},
        {"atomicOr only                     1024", 8, 1024, slots},
        {"hot: dir16 sc1 + CAS (8 addrs)    1024", 5 | 32, 1024, slots},
        {"hot: dir16 plain + CAS (8 addrs)  1024", 5 | 16 | 32, 1024, slots},
        {"hot: dir16 sc1 only (8 addrs)     1024", 1 | 32, 1024, slots},
    };
    const uint32_t n = 65536;
    for (auto& c : cfgs) {
        std::vector<float> ts;
        for (int it = 0; it < 30; ++it) {
            CK(hipEventRecord(a, st));
            hipLaunchKernelGGL(k, dim3(n / c.bs), dim3(c.bs), 0, st, c.mode, dir, c.dslots - 1, bk, third, (1ull << 17) - 1, out, n, (uint32_t)(it + 1));
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1000.f);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-42s median %7.2f us  min %7.2f us\n", c.name, ts[ts.size() / 2], ts[0]);
    }
    return 0;
}
