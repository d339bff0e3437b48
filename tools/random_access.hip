// random_access.hip — what MI355X sustains for INDEPENDENT random accesses of bucket size (64 / 128 / 192 bytes) over a table of
// bucket-table size: the ceiling of the pipelines' table phase (every distinct key of a batch costs one random bucket read and one
// 64-byte record write), as opposed to the streaming 8 TB/s the bench line's roofline.frac is priced against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/random_access tools/random_access.hip && tools/random_access
// Each thread owns `PER` random buckets per round: all loads are issued before the first use (as k_own / k_front issue a tile's
// bucket loads together), optionally followed by a 64-byte write to the same bucket.  Launched with far more workgroups than CUs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

template <int CHUNKS, bool WRITE>   // CHUNKS x 64 bytes read per bucket
__global__ __launch_bounds__(256) void touch(uint4* table, unsigned long long n_buckets, uint32_t stride16, uint32_t seed, uint32_t rounds, uint32_t* out) {
    constexpr int PER = 4;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    unsigned long long x = ((unsigned long long)(g + 1) * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)seed << 32);
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint4 v[PER][CHUNKS * 4];
        unsigned long long at[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            at[k] = (x % n_buckets) * stride16;
#pragma unroll
            for (int c = 0; c < CHUNKS * 4; ++c) v[k][c] = table[at[k] + c];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
#pragma unroll
            for (int c = 0; c < CHUNKS * 4; ++c) acc += v[k][c].x ^ v[k][c].w;
            if (WRITE) {
#pragma unroll
                for (int c = 0; c < 4; ++c) table[at[k] + (CHUNKS - 1) * 4 + c] = make_uint4(acc, r, k, c);
            }
        }
    }
    out[g] = acc;
}

template <int CHUNKS, bool WRITE>
static void run(uint4* d, size_t bytes, uint32_t bucket_bytes, uint32_t* out) {
    const unsigned long long n_buckets = bytes / bucket_bytes;
    const uint32_t wgs = 256 * 16, rounds = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((touch<CHUNKS, WRITE>), dim3(wgs), dim3(256), 0, 0, d, n_buckets, bucket_bytes / 16, 1u, rounds, out);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((touch<CHUNKS, WRITE>), dim3(wgs), dim3(256), 0, 0, d, n_buckets, bucket_bytes / 16, 2u + i, rounds, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double acc = (double)reps * wgs * 256 * rounds * 4;
    const double rd = acc * CHUNKS * 64, wr = WRITE ? acc * 64 : 0;
    printf("%6.2f GB table, %3u-byte buckets, read %3d B%s: %7.1f M buckets/s, %6.2f TB/s read%s\n", bytes / 1073741824.0, bucket_bytes, CHUNKS * 64,
           WRITE ? " + write 64 B" : "              ", acc / ms / 1e3, rd / ms / 1e9, WRITE ? (" + " + std::to_string(wr / ms / 1e9).substr(0, 4) + " TB/s written").c_str() : "");
    hipEventDestroy(a); hipEventDestroy(b);
}

// what a streaming copy sustains (read + write, 16 bytes per thread and step): the measured counterpart of the 8 TB/s spec figure
__global__ __launch_bounds__(256) void stream_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}
// `random_access quick <table GB>`: ONE table size, the engine's access shape (a 128-byte bucket read, its 64-byte record written back) and
// the 64-byte read + write, plus the streaming copy — one JSON line for bench.py's roofline.measured (about a second)
static double run_quiet(int chunks, uint4* d, size_t bytes, uint32_t bucket_bytes, uint32_t* out) {
    const unsigned long long n_buckets = bytes / bucket_bytes;
    const uint32_t wgs = 256 * 16, rounds = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto launch = [&](uint32_t seed) {
        if (chunks == 1) hipLaunchKernelGGL((touch<1, true>), dim3(wgs), dim3(256), 0, 0, d, n_buckets, bucket_bytes / 16, seed, rounds, out);
        else hipLaunchKernelGGL((touch<2, true>), dim3(wgs), dim3(256), 0, 0, d, n_buckets, bucket_bytes / 16, seed, rounds, out);
    };
    launch(1u);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch(2u + i);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)reps * wgs * 256 * rounds * 4 / ms / 1e6;          // G buckets/s
}
static int quick(double gb) {
    uint32_t* out; hipMalloc(&out, 256 * 16 * 256 * 4);
    const size_t bytes = (size_t)(gb * (1ull << 30));
    uint4* d = nullptr;
    if (hipMalloc(&d, bytes + 4096) != hipSuccess) { printf("{\"error\": \"alloc failed\"}\n"); return 1; }
    hipMemset(d, 1, bytes);
    const double r128w64 = run_quiet(2, d, bytes, 128, out), r64w64 = run_quiet(1, d, bytes, 64, out);
    const size_t half16 = bytes / 32;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(stream_copy, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)d, d + half16, half16);
    hipEventRecord(a, 0);
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(stream_copy, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)d, d + half16, half16);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double copy_gbps = 4.0 * (double)half16 * 32.0 / ms / 1e6;   // read + written bytes
    printf("{\"table_gb\": %.2f, \"random_r128_w64_Gbuckets_s\": %.3f, \"random_r64_w64_Gbuckets_s\": %.3f, \"stream_copy_GBps\": %.1f}\n", gb, r128w64, r64w64, copy_gbps);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "quick") return quick(argc > 2 ? atof(argv[2]) : 4.5);
    uint32_t* out; hipMalloc(&out, 256 * 16 * 256 * 4);
    for (double gb : {0.5, 4.0, 16.0}) {
        const size_t bytes = (size_t)(gb * (1ull << 30));
        uint4* d = nullptr;
        if (hipMalloc(&d, bytes + 4096) != hipSuccess) { printf("alloc %.2f GB failed\n", gb); continue; }
        hipMemset(d, 1, bytes);
        run<1, false>(d, bytes, 64, out);
        run<1, true>(d, bytes, 64, out);
        run<2, false>(d, bytes, 128, out);
        run<3, false>(d, bytes, 192, out);
        run<3, true>(d, bytes, 192, out);
        run<1, false>(d, bytes, 256, out);     // 64 bytes out of a 256-byte bucket (a record without its key cell)
        hipFree(d);
    }
    // streaming reference: every thread reads consecutive 16-byte words
    return 0;
}
