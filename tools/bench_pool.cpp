// bench_pool.cpp — throughput of the drop-in surface itself: T caller threads (the gRPC goroutines of a daemon) each issuing RPCs
// of `items` requests (gubernator.go:183-306, cap 1000) against a GPUWorkerPool of S shards (workers.go:54-626).  Everything a
// front end does per request happens inside the timed calls: validation, HashKey, XXH64, placement, slot reservation, in-place
// stage filling, completion, response fan-out.
//   api = c   : guber_pool_get_rate_limits — the C ABI a binding calls (include/guber_gpu.h): structure-of-arrays in and out,
//               what the Go shim hands over per RPC (go/gpu_worker_pool.go)
//   api = cpp : V1Instance::GetRateLimits on std::string RateLimitReq objects (the mirror of the reference's Go types)
//   api = wire: guber_wire_pool_get_rate_limits — the payload stage (include/guber_wire.h): the callers hand over the SERIALIZED
//               GetRateLimitsReq of their RPC and get the serialized GetRateLimitsResp back (one compare-and-swap + one memcpy per RPC on
//               the host; decode, HashKey, XXH64, placement, evaluation and the answers' order on the device).  The callers parse
//               the response bytes inside the timed loop: that is what feeds the conservation check.
// The number is GATED: callers run concurrently, so no fixed serial order exists to replay through an oracle — what every
// serialisation of the reference implies for this workload (TOKEN_BUCKET, hits 1, limit 100, one 60 s window: algorithms.go:162-198) is
// checked instead, per key over ALL responses from the pool's creation on: admitted requests <= limit, their `remaining` values are
// exactly {limit-1 .. limit-admitted} (every admitted hit applied exactly once: none lost, none twice — compared by count and sum),
// every refused request says remaining 0, limit echoed, no item error.  Accumulated per caller without atomics, merged after the clock.
//   make -C gubernator_amd/csrc bench_pool && tools/bench_pool_c [threads] [shards] [items] [keys] [seconds] [batch_wait_us] [api]
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../gubernator_amd/csrc/worker_pool.h"
#include "../include/guber_wire.h"
#ifdef GUBER_BENCH_NO_WIRE   // the build against the oracle-backed engine stub (tests/hostsim: no device decoder, no front): api = wire is not available there
extern "C" int guber_wire_pool_create(guber_engine_t* const*, uint32_t, const struct guber_route_rule*, const guber_wire_pool_config_t*, guber_wire_pool_t**) { return GUBER_E_NO_DEVICE; }
extern "C" void guber_wire_pool_destroy(guber_wire_pool_t*) {}
extern "C" int guber_wire_pool_get_rate_limits(guber_wire_pool_t*, const uint8_t*, size_t, int, int, uint8_t*, size_t, size_t*) { return GUBER_E_NO_DEVICE; }
extern "C" size_t guber_wire_pool_response_bound(const uint8_t*, size_t) { return 0; }
extern "C" int guber_wire_pool_stats(guber_wire_pool_t*, guber_wire_pool_stats_t*) { return GUBER_E_NO_DEVICE; }
#define guber_last_error() "api = wire needs the product library"
#endif

using namespace gubernator;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct SoaRpc {                       // one RPC as a binding holds it after decoding the protobuf
    std::vector<uint8_t> name_bytes, ukey_bytes; std::vector<uint32_t> name_off, ukey_off, behavior;
    std::vector<int64_t> hits, limit, duration, burst, created; std::vector<int32_t> algorithm;
};

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 32, S = argc > 2 ? atoi(argv[2]) : 4, items = argc > 3 ? atoi(argv[3]) : 1000;
    const int K = argc > 4 ? atoi(argv[4]) : 1000000;
    const double seconds = argc > 5 ? atof(argv[5]) : 2.0;
    const int wait_us = argc > 6 ? atoi(argv[6]) : 200;
    const bool c_api = !(argc > 7 && strcmp(argv[7], "cpp") == 0);
    const bool wire_api = argc > 7 && strcmp(argv[7], "wire") == 0;
    guber_config_t cfg{};
    cfg.cache_size = (uint64_t)K * 2; cfg.max_batch = 65536; cfg.device = 0;
    guber_pool_t* cp = nullptr;
    // api = wire: S engines (streams shared four by four, as the pool does), the placement's rule, the payload stage over them
    std::vector<guber_engine_t*> w_eng;
    guber_placement_t* w_place = nullptr;
    guber_wire_pool_t* wp = nullptr;
    if (wire_api) {
        // (the HIP runtime maps streams onto four hardware queues: two decode streams + the front's routing stream leave ONE for the engines)
        const int n_streams = getenv("GUBER_BENCH_WIRE_ENGINE_STREAMS") ? atoi(getenv("GUBER_BENCH_WIRE_ENGINE_STREAMS")) : 1;
        std::vector<void*> stream_of(n_streams, nullptr);
        for (int i = 0; i < S; ++i) {
            guber_config_t c = cfg;
            c.cache_size = std::max<uint64_t>(1u << 16, cfg.cache_size / S * 2);
            const int sidx = (int)((int64_t)i * n_streams / S);
            c.stream = stream_of[sidx];
            guber_engine_t* e = nullptr;
            const int rc = guber_engine_create(&c, &e);
            if (rc != GUBER_OK) { printf("pool: guber_engine_create error %d: %s\n", rc, guber_last_error()); return 1; }
            if (!stream_of[sidx]) stream_of[sidx] = guber_engine_stream(e);
            w_eng.push_back(e);
        }
        guber_route_rule_t rule{};
        if (S > 1) {
            if (guber_placement_create((uint32_t)S, 0, &w_place) != GUBER_OK || guber_placement_export(w_place, &rule) != GUBER_OK) { printf("pool: placement error: %s\n", guber_last_error()); return 1; }
        }
        guber_wire_pool_config_t wc{};
        wc.batch_wait_us = (uint32_t)wait_us;
        if (const char* v = getenv("GUBER_BENCH_WIRE_STAGES")) wc.stages = (uint32_t)atoi(v);
        if (const char* v = getenv("GUBER_BENCH_WIRE_ITEMS")) wc.max_items = (uint32_t)atoi(v);
        if (const char* v = getenv("GUBER_BENCH_WIRE_DECODES")) wc.decodes_queued = (uint32_t)atoi(v);
        if (const char* v = getenv("GUBER_BENCH_WIRE_SPIN_US")) wc.spin_us = (uint32_t)atoi(v);
        const int rc = guber_wire_pool_create(w_eng.data(), (uint32_t)S, S > 1 ? &rule : nullptr, &wc, &wp);
        if (rc != GUBER_OK) { printf("pool: guber_wire_pool_create error %d: %s\n", rc, guber_last_error()); return 1; }
    } else {
        const int rc0 = guber_pool_create_sharded(&cfg, (uint32_t)S, 65536, (uint32_t)wait_us, &cp);
        if (rc0 != GUBER_OK) { printf("pool: error %d\n", rc0); return 1; }
    }
    // (the C++ objects behind the handle, for api = cpp: layout of struct guber_pool in worker_pool.cpp)
    struct Handle { GPUWorkerPool* pool; V1Instance* inst; };
    GPUWorkerPool* const pool_p = cp ? ((Handle*)cp)->pool : nullptr;
    V1Instance* const inst_p = cp ? ((Handle*)cp)->inst : nullptr;
    // Zipf-1.1 ranks over K keys by inverse-CDF on a precomputed table
    std::vector<double> cdf(K);
    double acc = 0;
    for (int i = 0; i < K; ++i) { acc += 1.0 / std::pow((double)(i + 1), 1.1); cdf[i] = acc; }
    std::atomic<uint64_t> done{0}, errors{0};
    struct Acc { uint32_t admitted = 0, refused = 0, bad = 0; uint64_t sum_rem = 0; };
    std::vector<std::vector<Acc>> accs(T);                            // per caller: one accumulator per (pre-drawn RPC, item)
    std::vector<std::vector<int>> kids(T);                            // ... and the key it is for
    const double t_created = now_s();
    std::vector<std::vector<float>> lat(T);                           // per-RPC latency samples (us) of the timed window
    std::atomic<bool> timing{false};
    std::atomic<bool> stop{false}, go{false};
    std::atomic<int> ready{0};
    auto worker = [&](int t) {
        std::mt19937_64 rng(1234 + t);
        std::uniform_real_distribution<double> U(0.0, acc);
        // the RPCs are drawn before the clock starts: what is timed is the pool, not the generator
        const int NR = std::max(4, 20000 / items);
        std::vector<std::vector<RateLimitReq>> rpcs;
        std::vector<SoaRpc> soa;
        std::vector<std::vector<uint8_t>> payloads;                    // api = wire: the serialized GetRateLimitsReq of every pre-drawn RPC
        auto put_varint = [](std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); };
        char buf[32];
        accs[t].assign((size_t)NR * items, Acc{}); kids[t].assign((size_t)NR * items, 0);
        Acc* const acc_t = accs[t].data();
        for (int q = 0; q < NR; ++q) {
            std::vector<RateLimitReq> reqs(items);
            SoaRpc a;
            a.name_off.push_back(0); a.ukey_off.push_back(0);
            for (auto& r : reqs) {
                const int k = (int)(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin());
                kids[t][(size_t)q * items + (&r - reqs.data())] = k;
                r.name = "bench";
                snprintf(buf, sizeof buf, "acct:%08d", k);
                r.unique_key = buf;
                r.hits = 1; r.limit = 100; r.duration = 60000; r.algorithm = 0; r.behavior = 0; r.created_at = 0;
                a.name_bytes.insert(a.name_bytes.end(), r.name.begin(), r.name.end()); a.name_off.push_back((uint32_t)a.name_bytes.size());
                a.ukey_bytes.insert(a.ukey_bytes.end(), r.unique_key.begin(), r.unique_key.end()); a.ukey_off.push_back((uint32_t)a.ukey_bytes.size());
                a.hits.push_back(1); a.limit.push_back(100); a.duration.push_back(60000); a.burst.push_back(0); a.created.push_back(0);
                a.algorithm.push_back(0); a.behavior.push_back(0);
            }
            if (wire_api) {                                            // gubernator.proto:137-182: name 1, unique_key 2, hits 3, limit 4, duration 5 (zero fields are not written)
                std::vector<uint8_t> pl;
                for (auto& r : reqs) {
                    std::vector<uint8_t> body;
                    body.push_back(0x0a); put_varint(body, r.name.size()); body.insert(body.end(), r.name.begin(), r.name.end());
                    body.push_back(0x12); put_varint(body, r.unique_key.size()); body.insert(body.end(), r.unique_key.begin(), r.unique_key.end());
                    body.push_back(0x18); put_varint(body, (uint64_t)r.hits);
                    body.push_back(0x20); put_varint(body, (uint64_t)r.limit);
                    body.push_back(0x28); put_varint(body, (uint64_t)r.duration);
                    pl.push_back(0x0a); put_varint(pl, body.size()); pl.insert(pl.end(), body.begin(), body.end());
                }
                payloads.push_back(std::move(pl));
            } else if (c_api) soa.push_back(std::move(a)); else rpcs.push_back(std::move(reqs));
        }
        std::vector<uint8_t> resp_buf(wire_api ? guber_wire_pool_response_bound(payloads[0].data(), payloads[0].size()) + 4096 : 0);
        std::vector<uint8_t> o_status(items), o_err(items); std::vector<int64_t> o_limit(items), o_rem(items), o_reset(items);
        guber_result_t out{};
        out.status = o_status.data(); out.limit = o_limit.data(); out.remaining = o_rem.data(); out.reset_time = o_reset.data(); out.err = o_err.data();
        ready++;
        while (!go.load()) std::this_thread::yield();
        std::vector<RateLimitResp> resps;
        std::string err;
        for (size_t it = 0; !stop.load(std::memory_order_relaxed); ++it) {
            const double c0 = now_s();
            if (wire_api) {
                const std::vector<uint8_t>& pl = payloads[it % NR];
                size_t rl = 0;
                const int rc = guber_wire_pool_get_rate_limits(wp, pl.data(), pl.size(), 1, 1, resp_buf.data(), resp_buf.size(), &rl);
                if (rc != GUBER_OK) { errors++; continue; }
                // the response: repeated RateLimitResp = 1 { status 1, limit 2, remaining 3, reset_time 4, error 5 }
                Acc* a0 = acc_t + (it % NR) * (size_t)items;
                const uint8_t* q = resp_buf.data(); const uint8_t* const qe = q + rl;
                int n_resp = 0;
                auto get_varint = [&](const uint8_t*& z) { uint64_t v = 0; int sh = 0; while (z < qe) { const uint8_t b = *z++; v |= (uint64_t)(b & 0x7f) << sh; sh += 7; if (!(b & 0x80)) break; } return v; };
                while (q < qe && n_resp < items) {
                    if (*q++ != 0x0a) { errors++; break; }
                    const uint64_t bl = get_varint(q);
                    const uint8_t* be = q + bl;
                    uint64_t status = 0, limit = 0, rem = 0; bool err = false;
                    while (q < be) {
                        const uint8_t tag = *q++;
                        if (tag == 0x2a) { const uint64_t l = get_varint(q); q += l; err = true; }
                        else { const uint64_t v = get_varint(q); if (tag == 0x08) status = v; else if (tag == 0x10) limit = v; else if (tag == 0x18) rem = v; }
                    }
                    Acc& x = a0[n_resp++];
                    if (err || limit != 100 || status > 1) { x.bad++; if (err) errors++; }
                    else if (status == 0) { x.admitted++; x.sum_rem += rem; if (rem > 99) x.bad++; }
                    else { x.refused++; if (rem != 0) x.bad++; }
                }
                if (n_resp != items || q != qe) errors++;
            } else if (c_api) {
                const SoaRpc& a = soa[it % NR];
                const int rc = guber_pool_get_rate_limits(cp, (uint32_t)items, a.name_bytes.data(), a.name_off.data(), a.ukey_bytes.data(), a.ukey_off.data(),
                                                          a.hits.data(), a.limit.data(), a.duration.data(), a.burst.data(), a.created.data(), a.algorithm.data(),
                                                          a.behavior.data(), &out, nullptr, 0);
                if (rc != GUBER_OK) { errors++; continue; }
                uint64_t bad = 0;
                Acc* a0 = acc_t + (it % NR) * (size_t)items;
                for (int q = 0; q < items; ++q) {
                    bad += o_err[q] != 0;
                    Acc& x = a0[q];
                    if (o_err[q] != 0 || o_limit[q] != 100 || o_status[q] > 1) x.bad++;
                    else if (o_status[q] == 0) { x.admitted++; x.sum_rem += (uint64_t)o_rem[q]; if (o_rem[q] < 0 || o_rem[q] > 99) x.bad++; }
                    else { x.refused++; if (o_rem[q] != 0) x.bad++; }
                }
                if (bad) errors += bad;
            } else {
                std::vector<RateLimitReq>& reqs = rpcs[it % NR];
                for (auto& r : reqs) r.created_at = 0;
                if (!inst_p->GetRateLimits(reqs, &resps, &err)) { errors++; continue; }
                Acc* a0 = acc_t + (it % NR) * (size_t)items;
                for (size_t q = 0; q < resps.size() && q < (size_t)items; ++q) {
                    const auto& o = resps[q];
                    Acc& x = a0[q];
                    if (!o.error.empty()) { errors++; x.bad++; }
                    else if (o.limit != 100 || (int)o.status > 1) x.bad++;
                    else if ((int)o.status == 0) { x.admitted++; x.sum_rem += (uint64_t)o.remaining; if (o.remaining < 0 || o.remaining > 99) x.bad++; }
                    else { x.refused++; if (o.remaining != 0) x.bad++; }
                }
            }
            if (timing.load(std::memory_order_relaxed) && lat[t].size() < 2000000) lat[t].push_back((float)((now_s() - c0) * 1e6));
            done.fetch_add((uint64_t)items, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(worker, t);
    while (ready.load() < T) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    go.store(true);
    std::this_thread::sleep_for(std::chrono::milliseconds(500));      // warm-up: keys become resident, threads spread out
    timing.store(true);
    auto batches_now = [&]() -> uint64_t { if (pool_p) return pool_p->batches_flushed(); guber_wire_pool_stats_t ws{}; guber_wire_pool_stats(wp, &ws); return ws.stages; };
    const uint64_t d0 = done.load(); const uint64_t b0 = batches_now();
    const double t0 = now_s();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const uint64_t d1 = done.load(); const uint64_t b1 = batches_now();
    const double t1 = now_s();
    timing.store(false);
    stop.store(true);
    for (auto& x : th) x.join();
    const double lifetime = now_s() - t_created;
    // conservation per key over everything the pool ever answered
    uint64_t keys_checked = 0, decisions_checked = 0, violations = 0;
    {
        std::unordered_map<int, Acc> per_key;
        per_key.reserve((size_t)T * 4096);
        for (int t = 0; t < T; ++t)
            for (size_t i = 0; i < accs[t].size(); ++i) {
                const Acc& x = accs[t][i];
                if (!(x.admitted | x.refused | x.bad)) continue;
                Acc& k = per_key[kids[t][i]];
                k.admitted += x.admitted; k.refused += x.refused; k.bad += x.bad; k.sum_rem += x.sum_rem;
            }
        for (const auto& kv : per_key) {
            const Acc& k = kv.second;
            ++keys_checked; decisions_checked += (uint64_t)k.admitted + k.refused + k.bad;
            const uint64_t a = k.admitted;
            bool ok = k.bad == 0 && a <= 100 && k.sum_rem == a * 100 - a * (a + 1) / 2;      // remaining of the i-th admitted hit = 100 - i
            if (k.refused && a != 100) ok = false;                                             // a key refuses only once its window's tokens are gone
            if (!ok && violations++ < 5) fprintf(stderr, "conservation violated for key %d: admitted %u refused %u bad %u sum(remaining) %llu\n", kv.first, k.admitted, k.refused, k.bad, (unsigned long long)k.sum_rem);
        }
        if (lifetime >= 55.0) { fprintf(stderr, "the run outlived the 60 s window: conservation not checked\n"); violations = ~0ull; }
    }
    guber_pool_metrics_t m{};
    if (pool_p) pool_p->Metrics(&m);
    guber_wire_pool_stats_t ws{};
    if (wp) guber_wire_pool_stats(wp, &ws);
    std::vector<float> all;
    for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    const double p50 = all.empty() ? 0 : all[all.size() / 2], p99 = all.empty() ? 0 : all[(size_t)(all.size() * 0.99)];
    printf("pool: %3d caller threads x %4d-item RPCs, %d shard(s), %d keys: %8.2f M decisions/s, %6.0f batches/s, avg batch %6.0f requests, errors %llu, rpc latency p50 %.1f us p99 %.1f us"
           ", conservation: %llu keys %llu decisions %llu violations (api %s; placement passes %llu, hot keys moved %llu; per batch: %.0f us flush->answers; per submission: %.1f us host, %.1f batches)\n",
           T, items, S, K, (d1 - d0) / (t1 - t0) / 1e6, (b1 - b0) / (t1 - t0), (b1 - b0) ? (double)(d1 - d0) / (b1 - b0) : 0.0,
           (unsigned long long)errors.load(), p50, p99, (unsigned long long)keys_checked, (unsigned long long)decisions_checked, (unsigned long long)violations, wire_api ? "wire" : c_api ? "c" : "cpp", (unsigned long long)m.rebalances, (unsigned long long)m.keys_moved,
           m.batches ? (double)m.send_duration_us_sum / m.batches : 0.0, m.submits ? (double)m.submit_us_sum / m.submits : 0.0, m.submits ? (double)m.batches / m.submits : 0.0);
    if (wp) {
        const double ns = ws.stages ? (double)ws.stages : 1.0;
        printf("wire pool: %llu stages (left because: full %llu, BatchWait %llu, decoder idle %llu), callers that waited for a stage %llu; per stage: first payload -> sealed %.0f us, "
               "sealed -> decoded %.0f us, decoded -> answers in host memory %.0f us, %.0f items, %.1f RPCs; the pool threads' own time per stage: decode enqueue %.1f us, routing enqueue %.1f us, evaluation enqueue %.1f us\n", (unsigned long long)ws.stages, (unsigned long long)ws.sealed_full,
               (unsigned long long)ws.sealed_wait, (unsigned long long)ws.sealed_idle, (unsigned long long)ws.open_waits, ws.fill_us_sum / ns, ws.decode_us_sum / ns, ws.eval_us_sum / ns,
               ws.items / ns, ws.rpcs / ns, ws.host_decode_ns / 1e3 / ns, ws.host_route_ns / 1e3 / ns, ws.host_eval_ns / 1e3 / ns);
        guber_wire_pool_destroy(wp);
        for (size_t i = w_eng.size(); i-- > 0;) guber_engine_destroy(w_eng[i]);
        if (w_place) guber_placement_destroy(w_place);
    }
    if (cp) guber_pool_destroy(cp);
    return 0;
}
