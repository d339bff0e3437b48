// wire_pool_tsan.cpp — TEST INFRASTRUCTURE: the payload stage's HOST protocol (gubernator_amd/csrc/guber_wire_pool.h: the callers'
// reservation word, the intake and front threads, the hand-over ring, the wake-ups, shutdown) compiled on its own against stand-ins
// for the device decoder and the front, so that it runs under ThreadSanitizer (the CPU build of the whole engine runs its kernels as
// fibers, which ThreadSanitizer cannot follow; tests/test_enginesim_cpu.py runs the real thing under AddressSanitizer).
// The stand-in decoder is the host transcoder (csrc/wire.cpp), the stand-in evaluation is the ORACLE; both complete "asynchronously":
// a collect call reports GUBER_PENDING a few times first, so every polling path is taken.  Never part of the product library.
//   make -C tests/hostsim wire_pool_tsan && /tmp/guber_wire_pool_tsan [callers] [rpcs per caller] [items per rpc]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/guber_gpu.h"
#include "../../include/guber_wire.h"
#include "../../oracle/guber_oracle.h"

// ---- what guber_wire_pool.h expects from the translation unit it is part of ---------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipStreamNonBlocking = 1 };
static hipError_t hipSetDevice(int) { return hipSuccess; }
static hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)new int(0); return hipSuccess; }
static hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static hipError_t hipStreamDestroy(hipStream_t s) { delete (int*)s; return hipSuccess; }
static thread_local std::string g_last_error;
static int fail(int code, const char* what, hipError_t = hipSuccess) { g_last_error = what; return code; }
#define HIPCHK(call) do { if ((call) != hipSuccess) return fail(GUBER_E_HIP, #call); } while (0)
template <typename T> struct PinBuf {
    T* p = nullptr; size_t cap = 0;
    int ensure(size_t n) { if (n <= cap) return 0; free(p); p = (T*)calloc(std::max<size_t>(n, 16), sizeof(T)); cap = n; return p ? 0 : GUBER_E_NOMEM; }
    void release() { free(p); p = nullptr; cap = 0; }
};
template <typename T> struct CohBuf : PinBuf<T> {};
struct guber_engine { int device = 0; };
extern "C" void* guber_alloc_pinned(size_t n) { return calloc(1, n ? n : 1); }     // (wire.cpp's GUBER_WIRE_PINNED batches: not used here)
extern "C" void guber_free_pinned(void* p) { free(p); }
static oracle_t* g_oracle;
static std::mutex g_oracle_mu;
extern "C" int guber_eval_batch(guber_engine_t*, const guber_batch_t* b, guber_result_t* r) {   // (the pool's direct path of a lone one-request RPC)
    std::lock_guard<std::mutex> lk(g_oracle_mu);
    oracle_eval_batch(g_oracle, b, r);
    return GUBER_OK;
}

// the stand-in decoder: payloads -> ONE host batch (wire.cpp), "decoded" after a few polls; evaluated by the oracle, "answered" after a few more
struct guber_front { int dummy = 0; bool pre_routed = false; };
struct guber_wire_dev {
    uint32_t max_items, max_bytes, max_rpcs;
    std::vector<uint8_t> buf;
    guber_wire_batch_t* wb = nullptr;
    std::vector<int32_t> status; std::vector<uint32_t> first, count;
    uint32_t nrpc = 0, n_items = 0; int polls = 0; bool dec_pending = false, eval_pending = false, routed = false;
    guber_result_t* dst = nullptr;
    uint8_t* enc = nullptr; uint32_t* enc_len = nullptr;
};
extern "C" int guber_front_create(guber_engine_t* const*, uint32_t, const guber_route_rule_t*, uint32_t, uint32_t, guber_front_t** out) { *out = new guber_front(); return 0; }
extern "C" void guber_front_destroy(guber_front_t* f) { delete f; }
extern "C" int guber_wire_dev_create(guber_engine_t*, uint32_t max_items, uint32_t max_payload_bytes, uint32_t max_rpcs, guber_wire_dev_t** out) {
    auto* d = new guber_wire_dev();
    d->max_items = max_items; d->max_bytes = max_payload_bytes + 16 * max_rpcs + 64; d->max_rpcs = max_rpcs;
    d->buf.resize(d->max_bytes);
    d->status.resize(max_rpcs); d->first.resize(max_rpcs); d->count.resize(max_rpcs);
    if (guber_wire_batch_create(max_items, max_payload_bytes + max_items + 64, 0, &d->wb)) { delete d; return GUBER_E_NOMEM; }
    *out = d;
    return 0;
}
extern "C" void guber_wire_dev_destroy(guber_wire_dev_t* d) { if (d) { guber_wire_batch_destroy(d->wb); delete d; } }
extern "C" int guber_wire_dev_set_stream(guber_wire_dev_t*, void*) { return 0; }
extern "C" int guber_wire_dev_buffer(guber_wire_dev_t* d, uint8_t** b, size_t* cap) { *b = d->buf.data(); *cap = d->max_bytes; return 0; }
extern "C" int guber_wire_dev_decode_staged_async(guber_wire_dev_t* d, const uint32_t* offs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                                  uint32_t max_per_rpc, int64_t now_ms) {
    if (d->dec_pending || d->eval_pending) return fail(GUBER_E_INVALID_ARG, "stand-in decoder: previous call not collected");
    guber_wire_batch_reset(d->wb, now_ms);
    d->nrpc = nrpc;
    for (uint32_t r = 0; r < nrpc; ++r) {
        uint32_t f = 0, c = 0;
        const int rc = guber_wire_decode_requests(d->wb, d->buf.data() + offs[r], lens[r], max_per_rpc, is_owner ? is_owner[r] : 1, &f, &c);
        d->status[r] = rc; d->first[r] = f; d->count[r] = rc ? 0 : c;
    }
    d->n_items = guber_wire_batch_size(d->wb);
    d->dec_pending = true; d->polls = 3;
    return 0;
}
extern "C" int guber_wire_dev_decode_collect(guber_wire_dev_t* d, int, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    *n_items = 0;
    if (!d->dec_pending) return 0;
    if (d->polls-- > 0) return GUBER_PENDING;
    d->dec_pending = false;
    for (uint32_t r = 0; r < d->nrpc; ++r) { status[r] = d->status[r]; first[r] = d->first[r]; count[r] = d->count[r]; }
    *n_items = d->n_items;
    return 0;
}
extern "C" int guber_wire_dev_route_front_async(guber_wire_dev_t* d, guber_front_t* f) {
    if (!d->n_items) return 0;
    if (f->pre_routed) return fail(GUBER_E_INVALID_ARG, "stand-in front: a generation is routed ahead already");   // (the protocol must never do this)
    f->pre_routed = true; d->routed = true; d->polls = 2;
    return 0;
}
extern "C" int guber_wire_dev_route_ready(guber_wire_dev_t* d, guber_front_t*) { if (!d->routed) return 0; return d->polls-- > 0 ? GUBER_PENDING : 0; }
extern "C" int guber_wire_dev_eval_front_async(guber_wire_dev_t* d, guber_front_t* f, guber_result_t* r) {
    if (d->routed) { d->routed = false; f->pre_routed = false; }
    if (!d->n_items) return 0;
    {
        std::lock_guard<std::mutex> lk(g_oracle_mu);
        oracle_eval_batch(g_oracle, guber_wire_batch_view(d->wb), guber_wire_batch_result(d->wb));
    }
    d->dst = r; d->eval_pending = true; d->polls = 3;
    return 0;
}
// (k_wire_enc's stand-in: the host transcoder writes every RPC's bytes where the kernel would — or, for an RPC with an item error, nothing but the mark)
namespace guber {
constexpr uint32_t WIRE_ENC_ITEM_MAX = 37, WIRE_ENC_RAW = 0xffffffffu;
static size_t wire_enc_off(uint32_t first, uint32_t r) { return ((size_t)first * WIRE_ENC_ITEM_MAX + (size_t)r * 32u) & ~(size_t)15; }
static size_t wire_enc_bytes(uint32_t max_items, uint32_t max_rpcs) { return (size_t)max_items * WIRE_ENC_ITEM_MAX + (size_t)max_rpcs * 32u + 64u; }
}
static int wire_dev_eval_front_enc_async(guber_wire_dev* d, guber_front* f, uint8_t* enc, uint32_t* enc_len, guber_result_t* raw) {
    d->enc = enc; d->enc_len = enc_len;
    return guber_wire_dev_eval_front_async(d, f, raw);
}
extern "C" int guber_wire_dev_eval_collect(guber_wire_dev_t* d, int) {
    if (!d->eval_pending) return 0;
    if (d->polls-- > 0) return GUBER_PENDING;
    d->eval_pending = false;
    const guber_result_t* s = guber_wire_batch_result(d->wb);
    const size_t n = d->n_items;
    memcpy(d->dst->status, s->status, n); memcpy(d->dst->err, s->err, n);
    memcpy(d->dst->limit, s->limit, n * 8); memcpy(d->dst->remaining, s->remaining, n * 8); memcpy(d->dst->reset_time, s->reset_time, n * 8);
    if (d->enc) for (uint32_t r = 0; r < d->nrpc; ++r) {
        const uint32_t first = d->first[r], count = d->status[r] ? 0 : d->count[r];
        bool bad = false;
        for (uint32_t i = 0; i < count; ++i) bad = bad || s->err[first + i] != 0;
        size_t used = 0;
        if (bad) d->enc_len[r] = guber::WIRE_ENC_RAW;
        else if (count == 0) d->enc_len[r] = 0;
        else if (guber_wire_encode_responses(d->wb, first, count, 1, d->enc + guber::wire_enc_off(first, r), (size_t)count * guber::WIRE_ENC_ITEM_MAX, &used)) return GUBER_E_HIP;
        else d->enc_len[r] = (uint32_t)used;
    }
    return 0;
}

#include "../../gubernator_amd/csrc/guber_wire_pool.h"

// ---- the test ----------------------------------------------------------------------------------------------------------------------
static void put_varint(std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
static std::vector<uint8_t> payload_of(const std::vector<std::string>& keys, int64_t limit) {
    std::vector<uint8_t> pl;
    for (auto& k : keys) {
        std::vector<uint8_t> b;
        b.push_back(0x0a); put_varint(b, 2); b.push_back('n'); b.push_back('s');
        b.push_back(0x12); put_varint(b, k.size()); b.insert(b.end(), k.begin(), k.end());
        b.push_back(0x18); put_varint(b, 1); b.push_back(0x20); put_varint(b, (uint64_t)limit); b.push_back(0x28); put_varint(b, 3600000);
        pl.push_back(0x0a); put_varint(pl, b.size()); pl.insert(pl.end(), b.begin(), b.end());
    }
    return pl;
}
struct Row { uint64_t status = 0, limit = 0, remaining = 0; bool err = false; };
static bool parse_resp(const uint8_t* q, size_t len, std::vector<Row>& rows) {
    const uint8_t* qe = q + len;
    auto gv = [&](const uint8_t*& z) { uint64_t v = 0; int sh = 0; while (z < qe) { const uint8_t b = *z++; v |= (uint64_t)(b & 0x7f) << sh; sh += 7; if (!(b & 0x80)) break; } return v; };
    while (q < qe) {
        if (*q++ != 0x0a) return false;
        const uint64_t bl = gv(q); const uint8_t* be = q + bl;
        Row r;
        while (q < be) { const uint8_t tag = *q++; if (tag == 0x2a) { q += gv(q); r.err = true; } else { const uint64_t v = gv(q); if (tag == 0x08) r.status = v; else if (tag == 0x10) r.limit = v; else if (tag == 0x18) r.remaining = v; } }
        rows.push_back(r);
    }
    return q == qe;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 12, RPCS = argc > 2 ? atoi(argv[2]) : 40, ITEMS = argc > 3 ? atoi(argv[3]) : 60;
    const int KEYS = T * 40, LIMIT = 40;
    int failures = 0;
    for (int round = 0; round < 3; ++round) {
        g_oracle = oracle_create(1 << 20, 1);
        guber_engine eng; guber_engine_t* engs[1] = {&eng};
        guber_wire_pool_config_t cfg{};
        cfg.stages = round == 1 ? 2 : 4; cfg.max_items = 2048; cfg.max_payload_bytes = 1 << 17; cfg.max_rpcs = round == 2 ? 3 : 32; cfg.batch_wait_us = 200;
        cfg.decodes_queued = round == 1 ? 1 : 2; cfg.spin_us = round == 2 ? 1 : 0;
        guber_wire_pool_t* pool = nullptr;
        if (guber_wire_pool_create(engs, 1, nullptr, &cfg, &pool)) { fprintf(stderr, "create failed: %s\n", g_last_error.c_str()); return 1; }
        guber_wire_pool_set_clock(pool, 1700000000000ll);
        std::vector<uint64_t> admitted(KEYS, 0), refused(KEYS, 0), sum_rem(KEYS, 0);
        std::mutex acc_mu;
        std::atomic<int> bad{0};
        auto caller = [&](int t) {
            std::mt19937 rng(1000 * round + t);
            std::vector<uint8_t> resp(1 << 16);
            std::vector<uint64_t> a(KEYS, 0), rf(KEYS, 0), sr(KEYS, 0);
            for (int q = 0; q < RPCS; ++q) {
                std::vector<std::string> keys; std::vector<int> ks;
                const int n = 1 + (int)(rng() % ITEMS);
                const int own = (int)(rng() % n);
                for (int i = 0; i < n; ++i) {
                    if (i == own) { keys.push_back("own" + std::to_string(t)); ks.push_back(-1); }
                    else { const int k = (int)(rng() % KEYS); keys.push_back("k" + std::to_string(k)); ks.push_back(k); }
                }
                std::vector<uint8_t> pl = payload_of(keys, LIMIT);
                size_t rl = 0;
                const int rc = guber_wire_pool_get_rate_limits(pool, pl.data(), pl.size(), 1, 1, resp.data(), resp.size(), &rl);
                std::vector<Row> rows;
                if (rc || !parse_resp(resp.data(), rl, rows) || (int)rows.size() != n) { bad++; continue; }
                for (int i = 0; i < n; ++i) {
                    const Row& r = rows[i];
                    if (r.err || r.limit != (uint64_t)LIMIT || r.status > 1) { bad++; continue; }
                    if (ks[i] < 0) continue;
                    if (r.status == 0) { a[ks[i]]++; sr[ks[i]] += r.remaining; } else { rf[ks[i]]++; if (r.remaining) bad++; }
                }
            }
            std::lock_guard<std::mutex> lk(acc_mu);
            for (int k = 0; k < KEYS; ++k) { admitted[k] += a[k]; refused[k] += rf[k]; sum_rem[k] += sr[k]; }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(caller, t);
        for (auto& x : th) x.join();
        guber_wire_pool_stats_t st{};
        guber_wire_pool_stats(pool, &st);
        guber_wire_pool_destroy(pool);
        // conservation: every admitted hit applied exactly once (the own keys share the limit: each caller's own key is one more key)
        int viol = 0;
        for (int k = 0; k < KEYS; ++k) {
            const uint64_t a = admitted[k];
            if (a > (uint64_t)LIMIT || sum_rem[k] != a * LIMIT - a * (a + 1) / 2 || (refused[k] && a != (uint64_t)LIMIT)) viol++;
        }
        printf("round %d: %d callers x %d RPCs: %llu RPCs in %llu stages (full %llu, BatchWait %llu, decoder idle %llu), bad %d, conservation violations %d\n", round, T, RPCS,
               (unsigned long long)st.rpcs, (unsigned long long)st.stages, (unsigned long long)st.sealed_full, (unsigned long long)st.sealed_wait, (unsigned long long)st.sealed_idle, bad.load(), viol);
        if (bad.load() || viol || st.rpcs != (uint64_t)T * RPCS) failures++;
        oracle_destroy(g_oracle);
    }
    // a pool that is destroyed while idle and one that never saw a caller
    {
        g_oracle = oracle_create(1 << 16, 1);
        guber_engine eng; guber_engine_t* engs[1] = {&eng};
        guber_wire_pool_t* pool = nullptr;
        guber_wire_pool_config_t cfg{}; cfg.stages = 2; cfg.max_items = 256; cfg.max_payload_bytes = 1 << 14; cfg.max_rpcs = 4;
        if (guber_wire_pool_create(engs, 1, nullptr, &cfg, &pool)) return 1;
        guber_wire_pool_destroy(pool);
        oracle_destroy(g_oracle);
    }
    printf(failures ? "WIRE POOL TSAN FAILED\n" : "WIRE POOL TSAN OK\n");
    return failures ? 1 : 0;
}
