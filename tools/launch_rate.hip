// launch_rate.hip — how many kernel launches per second the host side of this box sustains: T threads each on its own
// stream, one thread round-robin over several streams, and a captured graph of kernel nodes.  The two-launch pipeline
// needs 2 launches per 65 536-request batch, so launches/s / 2 x 65 536 bounds the decisions/s of any number of shards.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_rate tools/launch_rate.hip -lpthread && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>

struct Big { uint64_t w[60]; };                       // 480 bytes of kernel arguments, like the pipeline's kernels
__global__ void k_small_args(uint64_t* out, uint64_t v) { if (v == 0x1234567ull) out[0] = v; }
__global__ void k_big_args(uint64_t* out, Big b) { if (b.w[59] == 0x1234567ull) out[0] = b.w[0]; }

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    uint64_t* out; hipMalloc(&out, 64);
    const int N = 20000;
    Big big{}; for (int i = 0; i < 60; ++i) big.w[i] = i;
    for (int bigargs = 0; bigargs < 2; ++bigargs) {
        for (int T : {1, 2, 4, 8}) {
            std::vector<hipStream_t> st(T);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            auto body = [&](int t) {
                for (int i = 0; i < N; ++i) {
                    if (bigargs) hipLaunchKernelGGL(k_big_args, dim3(1), dim3(64), 0, st[t], out, big);
                    else hipLaunchKernelGGL(k_small_args, dim3(1), dim3(64), 0, st[t], out, (uint64_t)i);
                }
            };
            body(0); hipDeviceSynchronize();
            const double t0 = now_s();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(body, t);
            for (auto& x : th) x.join();
            const double t1 = now_s();
            hipDeviceSynchronize();
            const double t2 = now_s();
            printf("%-10s %d thread(s) x own stream : enqueue %7.2f us/launch/thread, %8.0f launches/s aggregate (enqueue), %8.0f /s incl. drain\n",
                   bigargs ? "480-B args" : "16-B args", T, (t1 - t0) / N * 1e6, T * N / (t1 - t0), T * N / (t2 - t0));
            for (auto& s : st) hipStreamDestroy(s);
        }
    }
    // one thread, round-robin over S streams
    for (int S : {1, 4, 8}) {
        std::vector<hipStream_t> st(S);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        hipDeviceSynchronize();
        const double t0 = now_s();
        for (int i = 0; i < 4 * N; ++i) hipLaunchKernelGGL(k_big_args, dim3(1), dim3(64), 0, st[i % S], out, big);
        const double t1 = now_s();
        hipDeviceSynchronize();
        const double t2 = now_s();
        printf("480-B args 1 thread round-robin over %d stream(s): enqueue %7.2f us/launch, %8.0f launches/s (enqueue), %8.0f /s incl. drain\n",
               S, (t1 - t0) / (4 * N) * 1e6, 4 * N / (t1 - t0), 4 * N / (t2 - t0));
        for (auto& s : st) hipStreamDestroy(s);
    }
    // a captured graph of G kernel nodes in a chain
    for (int G : {2, 16, 64}) {
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < G; ++i) hipLaunchKernelGGL(k_big_args, dim3(1), dim3(64), 0, s, out, big);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        const int L = 40000 / G;
        const double t0 = now_s();
        for (int i = 0; i < L; ++i) hipGraphLaunch(ge, s);
        const double t1 = now_s();
        hipStreamSynchronize(s);
        const double t2 = now_s();
        printf("graph of %2d kernel nodes: enqueue %7.2f us/graph = %6.2f us/node, %8.0f nodes/s (enqueue), %8.0f /s incl. drain\n",
               G, (t1 - t0) / L * 1e6, (t1 - t0) / L / G * 1e6, (double)L * G / (t1 - t0), (double)L * G / (t2 - t0));
        hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
    }
    // host-side cost of the other calls a stage submit makes (enqueue only; the GPU side drains afterwards)
    {
        hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        char *h, *d; const size_t bytes = 1700000;
        hipHostMalloc(&h, bytes * 4, hipHostMallocDefault); hipMalloc(&d, bytes * 4);
        const int M = 200;
        hipDeviceSynchronize();
        double t0 = now_s();
        for (int i = 0; i < M; ++i) hipMemcpyAsync(d + (i % 4) * bytes, h + (i % 4) * bytes, bytes, hipMemcpyHostToDevice, a);
        double t1 = now_s(); hipDeviceSynchronize();
        printf("hipMemcpyAsync H2D 1.7 MB pinned : %6.2f us host time per call\n", (t1 - t0) / M * 1e6);
        t0 = now_s();
        for (int i = 0; i < M; ++i) hipMemcpyAsync(h + (i % 4) * bytes, d + (i % 4) * bytes, bytes, hipMemcpyDeviceToHost, a);
        t1 = now_s(); hipDeviceSynchronize();
        printf("hipMemcpyAsync D2H 1.7 MB pinned : %6.2f us host time per call\n", (t1 - t0) / M * 1e6);
        t0 = now_s();
        for (int i = 0; i < M; ++i) hipMemcpyAsync(h + (i % 4) * bytes, d + (i % 4) * bytes, 4096, hipMemcpyDeviceToHost, a);
        t1 = now_s(); hipDeviceSynchronize();
        printf("hipMemcpyAsync D2H 4 KB pinned   : %6.2f us host time per call\n", (t1 - t0) / M * 1e6);
        t0 = now_s();
        for (int i = 0; i < 2000; ++i) hipEventRecord(ev, a);
        t1 = now_s(); hipDeviceSynchronize();
        printf("hipEventRecord                   : %6.2f us host time per call\n", (t1 - t0) / 2000 * 1e6);
        t0 = now_s();
        for (int i = 0; i < 2000; ++i) { hipEventRecord(ev, a); hipStreamWaitEvent(b, ev, 0); }
        t1 = now_s(); hipDeviceSynchronize();
        printf("hipEventRecord + StreamWaitEvent : %6.2f us host time per pair\n", (t1 - t0) / 2000 * 1e6);
    }
    return 0;
}
