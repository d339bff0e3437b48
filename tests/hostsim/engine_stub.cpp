// engine_stub.cpp — TEST INFRASTRUCTURE: the subset of the engine's C ABI that gubernator::GPUWorkerPool calls, answered on
// the CPU by the oracle, so that the pool's host logic (slot reservation by the callers, stage rotation, generations,
// shutdown) can be exercised — also under ThreadSanitizer — on a machine without a GPU.  Linked only into
// tests/hostsim/pool_test; never part of the product library, which has no CPU path at all.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../include/guber_gpu.h"
#include "../../oracle/guber_oracle.h"

// ---- trace ring (builds with -DGUBER_POOL_TRACE: the pool reports reservations / seals / submissions / moves through guber_pool_trace,
// the stub adds what each engine evaluated; pool_test.cpp prints the ring on the first mismatch) ----
#include <atomic>
#include <string>
struct TraceEv { int64_t us; uint64_t tid; const char* what; const void* obj; uint64_t a, b, c; };
static TraceEv g_ring[1 << 15];
static std::atomic<uint64_t> g_ring_n{0};
extern "C" void guber_pool_trace(const char* what, const void* obj, uint64_t a, uint64_t b, uint64_t c) {
    const uint64_t k = g_ring_n.fetch_add(1);
    TraceEv& e = g_ring[k & ((1u << 15) - 1)];
    e.us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    e.tid = (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xffff; e.what = what; e.obj = obj; e.a = a; e.b = b; e.c = c;
}
extern "C" void guber_pool_trace_dump(uint32_t last) {
    const uint64_t n = g_ring_n.load();
    const uint64_t lo = n > last ? n - last : 0;
    for (uint64_t k = lo; k < n; ++k) {
        const TraceEv& e = g_ring[k & ((1u << 15) - 1)];
        fprintf(stderr, "TRACE %lld t%04llx %-14s %p %llx %llx %llx\n", (long long)e.us, (unsigned long long)e.tid, e.what, e.obj, (unsigned long long)e.a, (unsigned long long)e.b, (unsigned long long)e.c);
    }
}
#ifdef GUBER_POOL_TRACE
// what an engine evaluated: per batch, for every "*_hot" key the number of its requests and the first one's answer
static void trace_eval(const void* eng, const void* stage, const guber_batch_t* b, const guber_result_t* r) {
    struct K { uint64_t h; uint32_t n; int64_t first; };
    std::vector<K> ks;
    for (uint32_t i = 0; i < b->n; ++i) {
        const uint32_t off = b->key_off[i], len = b->key_off[i + 1] - off;
        if (len < 4 || memcmp(b->key_bytes + off + len - 4, "_hot", 4) != 0) continue;
        const uint64_t h = guber_xxhash64(b->key_bytes + off, len, 0);
        bool seen = false;
        for (auto& k : ks) if (k.h == h) { k.n++; seen = true; break; }
        if (!seen) ks.push_back({h, 1, r->remaining[i]});
    }
    for (auto& k : ks) { guber_pool_trace("eval", eng, k.h, k.n, (uint64_t)k.first); guber_pool_trace("  in stage", stage, b->n, 0, 0); }
}
#else
static void trace_eval(const void*, const void*, const guber_batch_t*, const guber_result_t*) {}
#endif

struct guber_engine { oracle_t* o; std::mutex mu; uint32_t max_batch; uint64_t cache_size; guber_route_rule_t rule{}; bool have_rule = false;
                      // the rule's arrays are COPIED at the call, as the engine uploads them (guber_engine.hip guber_stage_route): the caller's snapshot may be retired afterwards
                      std::vector<uint16_t> rt_table, rt_exs; std::vector<uint64_t> rt_exh; };
struct guber_stage {
    guber_engine* e; uint32_t max_n, key_cap;
    std::vector<uint32_t> off, beh; std::vector<int64_t> hits, limit, duration, burst, created, rl, rr, rs;
    std::vector<uint8_t> algo, owner, status, err, keys;
    std::vector<uint32_t> dest;
    uint32_t route_counts[16] = {0}, route_engines = 0;
    std::chrono::steady_clock::time_point route_ready{};
    guber_batch_t b{}; guber_result_t r{};
    bool in_flight = false;
    std::chrono::steady_clock::time_point ready_at{};                // guber_stages_submit: when the "GPU" is done with it
    std::mt19937 rng{12345};
};

extern "C" int guber_engine_create(const guber_config_t* cfg, guber_engine_t** out) {
    guber_engine* e = new guber_engine();
    e->o = oracle_create(cfg->cache_size ? cfg->cache_size : 50000, 1);
    e->max_batch = cfg->max_batch; e->cache_size = cfg->cache_size;
    *out = e;
    return GUBER_OK;
}
extern "C" void guber_engine_destroy(guber_engine_t* e) { if (e) { oracle_destroy(e->o); delete e; } }
extern "C" int guber_stage_create(guber_engine_t* e, uint32_t max_n, uint32_t key_bytes_cap, guber_stage_t** out) {
    guber_stage* s = new guber_stage();
    s->e = e; s->max_n = max_n; s->key_cap = key_bytes_cap ? key_bytes_cap : max_n * 64;
    const size_t n = max_n;
    s->off.assign(n + 1, 0); s->dest.assign(n, 0); s->beh.assign(n, 0); s->hits.assign(n, 0); s->limit.assign(n, 0); s->duration.assign(n, 0);
    s->burst.assign(n, 0); s->created.assign(n, 0); s->rl.assign(n, 0); s->rr.assign(n, 0); s->rs.assign(n, 0);
    s->algo.assign(n, 0); s->owner.assign(n, 0); s->status.assign(n, 0); s->err.assign(n, 0); s->keys.assign(s->key_cap + 64, 0);
    s->b.key_bytes = s->keys.data(); s->b.key_off = s->off.data(); s->b.hits = s->hits.data(); s->b.limit = s->limit.data();
    s->b.duration = s->duration.data(); s->b.burst = s->burst.data(); s->b.created_at = s->created.data();
    s->b.algorithm = s->algo.data(); s->b.behavior = s->beh.data(); s->b.is_owner = s->owner.data();
    s->r.status = s->status.data(); s->r.limit = s->rl.data(); s->r.remaining = s->rr.data(); s->r.reset_time = s->rs.data(); s->r.err = s->err.data();
    *out = s;
    return GUBER_OK;
}
extern "C" void guber_stage_destroy(guber_stage_t* s) { delete s; }
extern "C" guber_batch_t* guber_stage_batch(guber_stage_t* s) { return &s->b; }
extern "C" guber_result_t* guber_stage_result(guber_stage_t* s) { return &s->r; }
extern "C" int guber_stage_submit(guber_stage_t* s) {
    if (s->in_flight) return GUBER_E_INVALID_ARG;
    if (s->b.n > s->max_n || (s->b.n && s->b.key_off[s->b.n] > s->key_cap)) return GUBER_E_BATCH_TOO_LARGE;
    {   // evaluation order = submission order, like the engine stream
        std::lock_guard<std::mutex> lk(s->e->mu);
        static const bool null_engine = getenv("GUBER_STUB_NULL") != nullptr;   // measure the pool alone (tools/bench_pool.cpp on a CPU box)
        if (s->b.n && !null_engine) { oracle_eval_batch(s->e->o, &s->b, &s->r); trace_eval(s->e, s, &s->b, &s->r); }
        if (null_engine) memset(s->err.data(), 0, s->b.n);
    }
    s->in_flight = true;
    return GUBER_OK;
}
extern "C" int guber_stage_wait(guber_stage_t* s) {
    static const bool null_engine = getenv("GUBER_STUB_NULL") != nullptr;
    if (s->in_flight && !null_engine) std::this_thread::sleep_for(std::chrono::microseconds(s->rng() % 300));   // the GPU takes a while
    s->in_flight = false;
    return GUBER_OK;
}
// several stages in one submission (the pool's dispatcher): evaluated in array order, "complete" a little later
extern "C" int guber_stages_submit(guber_stage_t* const* stages, uint32_t n, uint32_t, uint32_t* done) {
    static const bool null_engine = getenv("GUBER_STUB_NULL") != nullptr;
    if (done) *done = 0;
    for (uint32_t k = 0; k < n; ++k) {
        guber_stage* s = stages[k];
        for (uint32_t q = 0; q < k; ++q) if (stages[q]->e == s->e) return GUBER_E_INVALID_ARG;
        const int rc = guber_stage_submit(s);
        if (rc) return rc;
        static const long null_lat = getenv("GUBER_STUB_LAT_US") ? atol(getenv("GUBER_STUB_LAT_US")) : 0;   // a fixed device latency under GUBER_STUB_NULL
        s->ready_at = std::chrono::steady_clock::now() + std::chrono::microseconds(null_engine ? null_lat : (long)(s->rng() % 300));
        if (done) *done = k + 1;
    }
    return GUBER_OK;
}
// one stage for several engines: every engine's share is gathered in rank order, evaluated by that engine's oracle, and the
// answers go back to the slots the requests were written at; the ranks must be a permutation of each share (checked)
extern "C" uint32_t* guber_stage_dest(guber_stage_t* s) { return s->dest.data(); }
// guber_stage_route on the host: the same arithmetic as guber_placement_shard over the exported rule (what k_route_count applies)
static uint32_t stub_route(const guber_route_rule_t& R, uint64_t h) {
    if (R.ex_n) {
        for (uint32_t i = (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 56) & (R.ex_cells - 1);; i = (i + 1) & (R.ex_cells - 1)) {
            if (R.ex_hash[i] == h) return R.ex_shard[i];
            if (R.ex_hash[i] == 0) break;
        }
    }
    const uint64_t h63 = h >> 1;
    uint64_t w = (uint64_t)(((unsigned __int128)h63 * R.inv_step) >> 64);
    if ((w + 1) * R.step <= h63) ++w;
    if (w >= R.n_shards) w = R.n_shards - 1;
    uint64_t sub = (uint64_t)(((unsigned __int128)(h63 - w * R.step) * R.inv_sub) >> 64);
    if (sub >= R.per) sub = R.per - 1;
    return R.table[(uint32_t)(w * R.per + sub)];
}
extern "C" int guber_stage_route(guber_stage_t* s, const guber_route_rule_t* rule, uint32_t n_engines) {
    static const long null_lat = getenv("GUBER_STUB_ROUTE_LAT_US") ? atol(getenv("GUBER_STUB_ROUTE_LAT_US")) : 0;
    if (!s || s->in_flight || n_engines == 0 || n_engines > 16) return GUBER_E_INVALID_ARG;
    if (rule) {
        guber_engine* e = s->e;
        if (rule->n_shards == 0 || rule->per == 0 || !rule->table || (rule->ex_n && (!rule->ex_hash || !rule->ex_shard))) return GUBER_E_INVALID_ARG;
        e->rule = *rule;
        e->rt_table.assign(rule->table, rule->table + (size_t)rule->n_shards * rule->per); e->rule.table = e->rt_table.data();
        if (rule->ex_n) {
            e->rt_exh.assign(rule->ex_hash, rule->ex_hash + rule->ex_cells); e->rt_exs.assign(rule->ex_shard, rule->ex_shard + rule->ex_cells);
            e->rule.ex_hash = e->rt_exh.data(); e->rule.ex_shard = e->rt_exs.data();
        }
        e->have_rule = true;
    }
    if (!s->e->have_rule) return GUBER_E_INVALID_ARG;
    const guber_route_rule_t& R = s->e->rule;
    const uint32_t n = s->b.n;
    if (n > s->max_n || n > 65536) return GUBER_E_BATCH_TOO_LARGE;
    memset(s->route_counts, 0, sizeof s->route_counts);
    static const bool null_engine = getenv("GUBER_STUB_NULL") != nullptr;   // the pool measured alone: the "device" routes at no host cost
    if (null_engine) {
        for (uint32_t i = 0; i < n; ++i) { const uint32_t e = i % n_engines; s->dest[i] = e << 24 | s->route_counts[e]++; }
        s->route_engines = n_engines;
        s->route_ready = std::chrono::steady_clock::now() + std::chrono::microseconds(null_lat);
        return GUBER_OK;
    }
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t off = s->b.key_off[i], len = s->b.key_off[i + 1] - off;
        uint32_t e = 0;
        if (R.global_engine >= 0 && (s->b.behavior[i] & 2u)) e = (uint32_t)R.global_engine;
        else if (len && R.n_shards > 1) e = stub_route(R, guber_xxhash64(s->b.key_bytes + off, len, 0));
        if (e >= n_engines) e = 0;
        s->dest[i] = e << 24 | s->route_counts[e]++;
    }
    s->route_engines = n_engines;
    s->route_ready = std::chrono::steady_clock::now() + std::chrono::microseconds(null_lat);
    return GUBER_OK;
}
extern "C" int guber_stage_route_poll(guber_stage_t* s, uint32_t* counts) {
    if (!s || !counts) return GUBER_E_INVALID_ARG;
    if (std::chrono::steady_clock::now() < s->route_ready) return 0;
    for (uint32_t j = 0; j < s->route_engines; ++j) counts[j] = s->route_counts[j];
    return 1;
}
extern "C" int guber_stage_submit_routed(guber_stage_t* s, guber_engine_t* const* engines, uint32_t n_engines, const uint32_t* counts) {
    static const bool null_engine = getenv("GUBER_STUB_NULL") != nullptr;
    static const long null_lat = getenv("GUBER_STUB_LAT_US") ? atol(getenv("GUBER_STUB_LAT_US")) : 0;
    if (s->in_flight || n_engines == 0 || n_engines > 16) return GUBER_E_INVALID_ARG;
    const uint32_t n = s->b.n;
    if (n > s->max_n || (n && s->b.key_off[n] > s->key_cap)) return GUBER_E_BATCH_TOO_LARGE;
    uint64_t total = 0;
    for (uint32_t j = 0; j < n_engines; ++j) total += counts[j];
    if (total != n) return GUBER_E_INVALID_ARG;
    if (null_engine) memset(s->err.data(), 0, n);
    for (uint32_t j = 0; j < n_engines && !null_engine; ++j) {
        const uint32_t nj = counts[j];
        if (!nj) continue;
        std::vector<uint32_t> at(nj, 0xffffffffu);
        for (uint32_t i = 0; i < n; ++i) {
            if ((s->dest[i] >> 24) != j) continue;
            const uint32_t r = s->dest[i] & 0xffffffu;
            if (r >= nj || at[r] != 0xffffffffu) { fprintf(stderr, "engine stub: ranks of engine %u are no permutation (request %u, rank %u of %u)\n", j, i, r, nj); abort(); }
            at[r] = i;
        }
        for (uint32_t r = 0; r < nj; ++r) if (at[r] == 0xffffffffu) { fprintf(stderr, "engine stub: rank %u of engine %u is missing\n", r, j); abort(); }
        std::vector<uint8_t> keys; std::vector<uint32_t> off(nj + 1, 0), beh(nj); std::vector<int64_t> hits(nj), limit(nj), duration(nj), burst(nj), created(nj), rl(nj), rr(nj), rs(nj);
        std::vector<uint8_t> algo(nj), owner(nj), status(nj), err(nj);
        for (uint32_t r = 0; r < nj; ++r) {
            const uint32_t i = at[r];
            keys.insert(keys.end(), s->b.key_bytes + s->b.key_off[i], s->b.key_bytes + s->b.key_off[i + 1]);
            off[r + 1] = (uint32_t)keys.size();
            hits[r] = s->b.hits[i]; limit[r] = s->b.limit[i]; duration[r] = s->b.duration[i]; burst[r] = s->b.burst[i]; created[r] = s->b.created_at[i];
            algo[r] = s->b.algorithm[i]; beh[r] = s->b.behavior[i]; owner[r] = s->b.is_owner[i];
        }
        keys.resize(keys.size() + 16, 0);
        guber_batch_t b{}; guber_result_t res{};
        b.n = nj; b.key_bytes = keys.data(); b.key_off = off.data(); b.hits = hits.data(); b.limit = limit.data(); b.duration = duration.data(); b.burst = burst.data();
        b.created_at = created.data(); b.algorithm = algo.data(); b.behavior = beh.data(); b.is_owner = owner.data(); b.now_ms = s->b.now_ms;
        res.status = status.data(); res.limit = rl.data(); res.remaining = rr.data(); res.reset_time = rs.data(); res.err = err.data();
        {
            std::lock_guard<std::mutex> lk(engines[j]->mu);
            oracle_eval_batch(engines[j]->o, &b, &res);
            trace_eval(engines[j], s, &b, &res);
        }
        for (uint32_t r = 0; r < nj; ++r) {
            const uint32_t i = at[r];
            s->status[i] = status[r]; s->rl[i] = rl[r]; s->rr[i] = rr[r]; s->rs[i] = rs[r]; s->err[i] = err[r];
        }
    }
    s->in_flight = true;
    s->ready_at = std::chrono::steady_clock::now() + std::chrono::microseconds(null_engine ? null_lat : (long)(s->rng() % 300));
    return GUBER_OK;
}
extern "C" int guber_stage_poll(guber_stage_t* s) { return !s->in_flight || std::chrono::steady_clock::now() >= s->ready_at ? 1 : 0; }
extern "C" void* guber_engine_stream(guber_engine_t*) { return nullptr; }
// a hot key changes its logical shard: found by its XXH64 among the oracle's items (test-only: a scan)
extern "C" int guber_move_items_by_hash(guber_engine_t* from, guber_engine_t* to, const uint64_t* hashes, uint32_t n, uint32_t* moved) {
    if (moved) *moved = 0;
    if (!from || !to || from == to) return GUBER_E_INVALID_ARG;
    guber_engine* a = from < to ? from : to; guber_engine* b = from < to ? to : from;
    std::lock_guard<std::mutex> la(a->mu); std::lock_guard<std::mutex> lb(b->mu);
    std::vector<guber_item_t> items((size_t)oracle_size(from->o) + 16);
    const uint64_t m = oracle_each(from->o, items.data(), items.size());
    for (uint32_t i = 0; i < n; ++i)
        for (uint64_t q = 0; q < m; ++q) {
            if (oracle_xxhash64(items[q].key, items[q].key_len, 0) != hashes[i]) continue;
            guber_item_t it = items[q];
            std::vector<uint8_t> key(it.key, it.key + it.key_len);
            it.key = key.data();
            int ex = 0;
            oracle_add_item(to->o, &it, 0, &ex);
            oracle_remove_item(from->o, key.data(), (uint32_t)key.size());
            if (moved) ++*moved;
            break;
        }
    return GUBER_OK;
}
// the GLOBAL exchange needs devices: not part of the CPU checks
extern "C" int guber_comm_create_local(guber_engine_t* const*, uint32_t, const guber_ring_t*, int, guber_comm_t**) { return GUBER_E_INVALID_ARG; }
extern "C" void guber_comm_destroy(guber_comm_t*) {}
extern "C" int guber_global_sync(guber_comm_t*, int64_t, guber_global_sync_stats_t*) { return GUBER_E_INVALID_ARG; }
extern "C" int guber_eval_batch(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r) {
    std::lock_guard<std::mutex> lk(e->mu);
    static const bool null_engine = getenv("GUBER_STUB_NULL") != nullptr;
    if (b->n && !null_engine) { oracle_eval_batch(e->o, b, r); trace_eval(e, nullptr, b, r); }
    if (null_engine) memset(r->err, 0, b->n);
    else { static thread_local std::mt19937 rng{99}; std::this_thread::sleep_for(std::chrono::microseconds(rng() % 40)); }   // the launch + the GPU take a while
    return GUBER_OK;
}
extern "C" int guber_add_items(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed) {
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t i = 0; i < n; ++i) { int ex = 0; oracle_add_item(e->o, &items[i], 0, &ex); if (existed) existed[i] = (uint8_t)ex; }
    return GUBER_OK;
}
extern "C" int guber_get_item(guber_engine_t* e, const uint8_t* key, uint32_t key_len, int64_t now_ms, guber_item_t* out, int* found) {
    std::lock_guard<std::mutex> lk(e->mu);
    return oracle_get_item(e->o, key, key_len, now_ms, out, found);
}
extern "C" int64_t guber_size(guber_engine_t* e) { std::lock_guard<std::mutex> lk(e->mu); return oracle_size(e->o); }
// not exercised on the CPU (the Store / Loader paths are covered by the GPU tests)
extern "C" int guber_dump(guber_engine_t*, guber_item_t*, uint64_t, uint8_t*, uint64_t, uint64_t*, uint64_t*) { return GUBER_E_INVALID_ARG; }
extern "C" int guber_probe_missing(guber_engine_t*, const guber_batch_t*, uint8_t*) { return GUBER_E_INVALID_ARG; }
extern "C" int guber_eval_batch_store(guber_engine_t*, const guber_batch_t*, guber_result_t*, guber_store_events_t*) { return GUBER_E_INVALID_ARG; }
