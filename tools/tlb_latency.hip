// tlb_latency.hip — dependent random 64-byte loads (pointer chase) over buffers of growing size: what one dependent table
// access of k_front costs on MI355X as a function of the table's footprint (HBM latency + address translation).
//   hipcc --offload-arch=gfx950 -O3 -o tools/tlb_latency tools/tlb_latency.hip && tools/tlb_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>
__global__ void chase(const uint32_t* next, uint32_t start_stride, int steps, uint32_t* out, unsigned long long* cycles) {
    uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) * start_stride;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) p = next[(size_t)p * 16];       // one 64-byte line per element
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
// independent random loads: every thread issues `k` loads at once, then waits (what a batch of heads does)
__global__ void burst(const uint32_t* buf, uint32_t n_lines, uint32_t seed, uint32_t* out, unsigned long long* cycles) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = (g + 1) * 2654435761u ^ seed;
    const unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    acc += buf[(size_t)(x % n_lines) * 16];
    const unsigned long long t1 = wall_clock64();
    out[g] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    for (double gb : {0.0625, 0.5, 2.0, 4.0, 8.0, 16.0}) {
        const size_t lines = (size_t)(gb * (1ull << 30) / 64);
        uint32_t* d = nullptr;
        if (hipMalloc(&d, lines * 64) != hipSuccess) { printf("alloc %.2f GB failed\n", gb); continue; }
        std::vector<uint32_t> h(lines * 16, 0);
        std::vector<uint32_t> perm(lines);
        for (size_t i = 0; i < lines; ++i) perm[i] = (uint32_t)i;
        std::mt19937_64 r(7);
        for (size_t i = lines - 1; i > 0; --i) std::swap(perm[i], perm[r() % (i + 1)]);
        for (size_t i = 0; i < lines; ++i) h[(size_t)perm[i] * 16] = perm[(i + 1) % lines];   // one big cycle
        hipMemcpy(d, h.data(), lines * 64, hipMemcpyHostToDevice);
        uint32_t* out; unsigned long long* cyc;
        hipMalloc(&out, 65536 * 4); hipMalloc(&cyc, 1024 * 8);
        std::vector<unsigned long long> hc(1024);
        for (int wg : {1, 256}) {
            const int steps = 200;
            hipLaunchKernelGGL(chase, dim3(wg), dim3(64), 0, 0, d, (uint32_t)(lines / (wg * 64)), steps, out, cyc);
            hipDeviceSynchronize();
            hipMemcpy(hc.data(), cyc, wg * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < wg; ++i) s += hc[i];
            printf("%6.2f GB  chase  %3d wave(s): %7.1f ns per dependent load\n", gb, wg, s / wg / steps * 10.0);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(burst, dim3(256), dim3(256), 0, 0, d, (uint32_t)lines, 1234u + rep, out, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0, mx = 0; for (int i = 0; i < 256; ++i) { s += hc[i]; mx = hc[i] > mx ? hc[i] : mx; }
        printf("%6.2f GB  burst  65536 independent loads (256 x 256): avg %7.1f ns, slowest workgroup %7.1f ns\n", gb, s / 256 * 10.0, mx * 10.0);
        hipFree(d); hipFree(out); hipFree(cyc);
    }
    return 0;
}
