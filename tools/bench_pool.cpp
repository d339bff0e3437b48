// bench_pool.cpp — throughput of the drop-in surface itself: T caller threads (the gRPC goroutines of a daemon) each
// calling V1Instance::GetRateLimits with RPCs of `items` requests (gubernator.go:183-306, cap 1000), against a
// GPUWorkerPool of S shards (workers.go:54-626).  Everything the Go shim would do per request happens here in C++:
// validation, HashKey, shard routing, queueing, in-place stage filling, submit / wait, response fan-out.
//   make -C gubernator_amd/csrc bench_pool && tools/bench_pool_c [threads] [shards] [items] [keys] [seconds] [batch_wait_us]
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../gubernator_amd/csrc/worker_pool.h"

using namespace gubernator;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 32, S = argc > 2 ? atoi(argv[2]) : 4, items = argc > 3 ? atoi(argv[3]) : 1000;
    const int K = argc > 4 ? atoi(argv[4]) : 1000000;
    const double seconds = argc > 5 ? atof(argv[5]) : 2.0;
    const int wait_us = argc > 6 ? atoi(argv[6]) : 200;
    guber_config_t cfg{};
    cfg.cache_size = (uint64_t)K * 2; cfg.max_batch = 65536; cfg.device = 0;
    GPUWorkerPool pool(cfg, 65536, (uint32_t)wait_us, (uint32_t)S);
    if (!pool.ok()) { printf("pool: error %d\n", pool.create_error()); return 1; }
    V1Instance inst(&pool);
    // Zipf-1.1 ranks over K keys by inverse-CDF on a precomputed table
    std::vector<double> cdf(K);
    double acc = 0;
    for (int i = 0; i < K; ++i) { acc += 1.0 / std::pow((double)(i + 1), 1.1); cdf[i] = acc; }
    std::atomic<uint64_t> done{0}, errors{0};
    std::atomic<bool> stop{false}, go{false};
    std::atomic<int> ready{0};
    auto worker = [&](int t) {
        std::mt19937_64 rng(1234 + t);
        std::uniform_real_distribution<double> U(0.0, acc);
        // the RPCs are drawn before the clock starts: what is timed is the pool, not the generator
        const int NR = std::max(4, 20000 / items);
        std::vector<std::vector<RateLimitReq>> rpcs(NR, std::vector<RateLimitReq>(items));
        char buf[32];
        for (auto& reqs : rpcs)
            for (auto& r : reqs) {
                const int k = (int)(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin());
                r.name = "bench";
                snprintf(buf, sizeof buf, "acct:%08d", k);
                r.unique_key = buf;
                r.hits = 1; r.limit = 100; r.duration = 60000; r.algorithm = 0; r.behavior = 0; r.created_at = 0;
            }
        ready++;
        while (!go.load()) std::this_thread::yield();
        std::vector<RateLimitResp> resps;
        std::string err;
        for (size_t it = 0; !stop.load(std::memory_order_relaxed); ++it) {
            std::vector<RateLimitReq>& reqs = rpcs[it % NR];
            for (auto& r : reqs) r.created_at = 0;
            if (!inst.GetRateLimits(reqs, &resps, &err)) { errors++; continue; }
            for (const auto& o : resps) if (!o.error.empty()) errors++;
            done.fetch_add((uint64_t)items, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(worker, t);
    while (ready.load() < T) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    go.store(true);
    std::this_thread::sleep_for(std::chrono::milliseconds(500));      // warm-up: keys become resident, threads spread out
    const uint64_t d0 = done.load(); const uint64_t b0 = pool.batches_flushed();
    const double t0 = now_s();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const uint64_t d1 = done.load(); const uint64_t b1 = pool.batches_flushed();
    const double t1 = now_s();
    stop.store(true);
    for (auto& x : th) x.join();
    guber_pool_metrics_t m{};
    pool.Metrics(&m);
    printf("pool: %3d caller threads x %4d-item RPCs, %d shard(s), %d keys: %8.2f M decisions/s, %6.0f batches/s, avg batch %6.0f requests, errors %llu\n",
           T, items, S, K, (d1 - d0) / (t1 - t0) / 1e6, (b1 - b0) / (t1 - t0), (b1 - b0) ? (double)(d1 - d0) / (b1 - b0) : 0.0,
           (unsigned long long)errors.load());
    pool.Close();
    return 0;
}
