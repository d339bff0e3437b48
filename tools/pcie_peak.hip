// pcie_peak.hip — what the host link of this box gives the end-to-end path: DMA copies (hipMemcpyAsync from / to pinned
// memory) of batch-sized blocks, one at a time and pipelined, both directions at once, and kernels reading / writing
// device-visible host memory in place (what guber_stage_* does today).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pcie_peak tools/pcie_peak.hip && /tmp/pcie_peak
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_read(const uint4* a, size_t n16, uint4* out) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345u && acc.y == 0x6789u) out[0] = acc;
}
__global__ void k_write(uint4* a, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4((unsigned)i, 1, 2, 3);
}

int main() {
    const size_t in_bytes = 65536 * 48, out_bytes = 65536 * 26;     // one batch: request columns + keys in, responses out
    const int NB = 8;
    char *h_in, *h_out, *d_in, *d_out; uint4* sink;
    hipHostMalloc(&h_in, in_bytes * NB, hipHostMallocDefault); hipHostMalloc(&h_out, out_bytes * NB, hipHostMallocDefault);
    hipMalloc(&d_in, in_bytes * NB); hipMalloc(&d_out, out_bytes * NB); hipMalloc(&sink, 64);
    for (size_t i = 0; i < in_bytes * NB; i += 4096) h_in[i] = 1;
    hipStream_t s_in, s_out, s_k;
    hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking); hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking); hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking);
    const int R = 400;
    // 1. one copy at a time (latency per batch-sized copy)
    double t0 = now_s();
    for (int i = 0; i < R; ++i) { hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s_in); hipStreamSynchronize(s_in); }
    double t1 = now_s();
    printf("H2D %zu KB, one at a time     : %7.1f us/copy  %6.1f GB/s\n", in_bytes >> 10, (t1 - t0) / R * 1e6, in_bytes * R / (t1 - t0) / 1e9);
    t0 = now_s();
    for (int i = 0; i < R; ++i) { hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s_out); hipStreamSynchronize(s_out); }
    t1 = now_s();
    printf("D2H %zu KB, one at a time     : %7.1f us/copy  %6.1f GB/s\n", out_bytes >> 10, (t1 - t0) / R * 1e6, out_bytes * R / (t1 - t0) / 1e9);
    // 2. back to back on one stream
    t0 = now_s();
    for (int i = 0; i < R; ++i) hipMemcpyAsync(d_in + (i % NB) * in_bytes, h_in + (i % NB) * in_bytes, in_bytes, hipMemcpyHostToDevice, s_in);
    hipStreamSynchronize(s_in); t1 = now_s();
    printf("H2D %zu KB, back to back      : %7.1f us/copy  %6.1f GB/s\n", in_bytes >> 10, (t1 - t0) / R * 1e6, in_bytes * R / (t1 - t0) / 1e9);
    t0 = now_s();
    for (int i = 0; i < R; ++i) hipMemcpyAsync(h_out + (i % NB) * out_bytes, d_out + (i % NB) * out_bytes, out_bytes, hipMemcpyDeviceToHost, s_out);
    hipStreamSynchronize(s_out); t1 = now_s();
    printf("D2H %zu KB, back to back      : %7.1f us/copy  %6.1f GB/s\n", out_bytes >> 10, (t1 - t0) / R * 1e6, out_bytes * R / (t1 - t0) / 1e9);
    // 3. both directions at once
    t0 = now_s();
    for (int i = 0; i < R; ++i) {
        hipMemcpyAsync(d_in + (i % NB) * in_bytes, h_in + (i % NB) * in_bytes, in_bytes, hipMemcpyHostToDevice, s_in);
        hipMemcpyAsync(h_out + (i % NB) * out_bytes, d_out + (i % NB) * out_bytes, out_bytes, hipMemcpyDeviceToHost, s_out);
    }
    hipStreamSynchronize(s_in); hipStreamSynchronize(s_out); t1 = now_s();
    printf("H2D + D2H concurrently          : %7.1f us/batch %6.1f GB/s in + %6.1f GB/s out -> %6.1f M batches-of-65536-decisions... %6.1f M decisions/s\n",
           (t1 - t0) / R * 1e6, in_bytes * R / (t1 - t0) / 1e9, out_bytes * R / (t1 - t0) / 1e9, R / (t1 - t0) / 1e6, 65536.0 * R / (t1 - t0) / 1e6);
    // 4. ten small copies per batch (one per column) instead of one block
    t0 = now_s();
    for (int i = 0; i < R; ++i)
        for (int c = 0; c < 10; ++c) hipMemcpyAsync(d_in + (i % NB) * in_bytes + c * (in_bytes / 10), h_in + (i % NB) * in_bytes + c * (in_bytes / 10), in_bytes / 10, hipMemcpyHostToDevice, s_in);
    hipStreamSynchronize(s_in); t1 = now_s();
    printf("H2D as 10 column copies/batch   : %7.1f us/batch %6.1f GB/s\n", (t1 - t0) / R * 1e6, in_bytes * R / (t1 - t0) / 1e9);
    // 5. kernels on host memory in place
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 1024, 4096}) {
        float ms;
        hipEventRecord(e0, s_k);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, s_k, (const uint4*)(h_in + (i % NB) * in_bytes), in_bytes / 16, sink);
        hipEventRecord(e1, s_k); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("kernel reads host memory (%4d WGs): %6.1f GB/s", blocks, in_bytes * 50 / (ms * 1e-3) / 1e9);
        hipEventRecord(e0, s_k);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, s_k, (uint4*)(h_out + (i % NB) * out_bytes), out_bytes / 16);
        hipEventRecord(e1, s_k); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("   writes host memory: %6.1f GB/s\n", out_bytes * 50 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
