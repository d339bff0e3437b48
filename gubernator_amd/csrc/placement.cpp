// placement.cpp — which logical shard of a GPU holds a key (include/guber_gpu.h guber_placement_*).
//
// The reference splits a node's cache over Config.Workers goroutines by XXH64 range (workers.go:125-151: hashRingStep =
// 2^63 / Workers; getWorker :180-184: idx = hash63 / hashRingStep).  Which worker holds a key never shows in a response, so
// the engine is free to place keys where the load is even — and it has to: a shard is a serial chain of batches, and with a
// skewed stream the shard of the hottest key is the chain everybody waits for (Zipf-1.1 over 10 M keys: one key carries
// 11.6 % of the requests).
//
// Keys map to `n_slots` hash slots by the reference's own rule (slot = hash63 / (2^63 / n_slots): the worker rule with
// n_slots virtual workers), slots map to shards through a table whose initial content — contiguous runs of slots — IS the
// reference's getWorker.  Keys that alone weigh more than a fraction of a shard's fair share are placed individually
// (an exception list keyed by the 64-bit key hash).  Table and list come from observed traffic: per-slot request counts
// and a two-row Misra-Gries sketch of heavy hitters, both updated with relaxed atomics (callers may observe
// concurrently; the estimates are approximate by design), assigned longest-processing-time-first.
//
// Readers (guber_placement_shard) are wait-free: the exception list is an immutable snapshot behind an atomic pointer,
// the slot table an array of atomics.  Moving a RESIDENT key is the caller's business (the pool migrates the bucket at a
// batch boundary, worker_pool.cpp); guber_placement_rebalance only reports the moves it made.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/guber_gpu.h"

#include "guber_placement_impl.h"
using namespace guber_placement_detail;

static uint32_t sketch_cell(uint64_t h, uint32_t row) {
    const uint64_t m = row ? 0xC2B2AE3D27D4EB4Full : 0x9E3779B97F4A7C15ull;
    return (uint32_t)(((h ^ (h >> 29)) * m) >> (64 - kSketchBits));
}

extern "C" int guber_placement_create(uint32_t n_shards, uint32_t n_slots, guber_placement_t** out) {
    if (!out || n_shards == 0 || n_shards > 4096) return GUBER_E_INVALID_ARG;
    if (n_slots == 0) n_slots = 4096;
    if (n_slots > (1u << 20)) return GUBER_E_INVALID_ARG;
    guber_placement* p = new guber_placement();
    p->n_shards = n_shards; p->per = std::max(1u, n_slots / n_shards); p->n_slots = n_slots = p->per * n_shards;
    p->step = (1ull << 63) / n_shards;
    p->inv_step = (uint64_t)((((unsigned __int128)1) << 64) / p->step);
    p->inv_sub = (uint64_t)(((((unsigned __int128)1) << 64) * p->per) / p->step);
    p->table.reset(new std::atomic<uint16_t>[n_slots]);
    p->slot_w.reset(new std::atomic<uint64_t>[n_slots]);
    for (uint32_t s = 0; s < n_slots; ++s) {
        p->table[s].store((uint16_t)(s / p->per));                             // contiguous runs: the reference's getWorker
        p->slot_w[s].store(0);
    }
    p->sketch.reset(new Cell[(size_t)kSketchRows << kSketchBits]);
    *out = p;
    return GUBER_OK;
}
extern "C" void guber_placement_destroy(guber_placement_t* p) { delete p; }

extern "C" uint32_t guber_placement_shard(const guber_placement_t* p, uint64_t key_hash) {
    return p ? guber_placement_shard_inl(p, key_hash) : 0;
}
extern "C" uint32_t guber_placement_version(const guber_placement_t* p) { return p ? p->version.load(std::memory_order_acquire) : 0; }

extern "C" void guber_placement_observe(guber_placement_t* p, uint64_t h, uint32_t weight) {
    if (!p || weight == 0) return;
    p->slot_w[p->slot_of(h)].fetch_add(weight, std::memory_order_relaxed);   // (the total is the sum of these: no shared hot word)
    for (uint32_t row = 0; row < kSketchRows; ++row) {       // Misra-Gries, one counter per cell: the cell's majority key survives
        Cell& c = p->sketch[((size_t)row << kSketchBits) + sketch_cell(h, row)];
        const uint64_t cur = c.h.load(std::memory_order_relaxed);
        const uint32_t cnt = c.c.load(std::memory_order_relaxed);
        if (cur == h) c.c.store(cnt + weight, std::memory_order_relaxed);
        else if (cnt <= weight) { c.h.store(h, std::memory_order_relaxed); c.c.store(weight - cnt, std::memory_order_relaxed); }
        else c.c.store(cnt - weight, std::memory_order_relaxed);
    }
}

extern "C" int guber_placement_route_keys(const guber_placement_t* p, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t n,
                                          uint32_t* shard_out, uint64_t* hash_out) {
    if (!p || (n && (!key_bytes || !key_off || (!shard_out && !hash_out)))) return GUBER_E_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t h = guber_xxhash64(key_bytes + key_off[i], key_off[i + 1] - key_off[i], 0);   // workers.go:153-155
        if (hash_out) hash_out[i] = h;
        if (shard_out) shard_out[i] = guber_placement_shard(p, h);
    }
    return GUBER_OK;
}
extern "C" int guber_placement_observe_keys(guber_placement_t* p, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t n) {
    if (!p || (n && (!key_bytes || !key_off))) return GUBER_E_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) guber_placement_observe(p, guber_xxhash64(key_bytes + key_off[i], key_off[i + 1] - key_off[i], 0), 1);
    return GUBER_OK;
}

// Longest-processing-time-first over what was observed since the last call.  move_slots != 0: slots and hot keys are all
// placed afresh (only legal while no key of this placement is resident anywhere: before the first request, or offline);
// move_slots == 0: the slot table stays, keys that became heavy are given a shard of their own choice (the least loaded one)
// and pins of keys that stopped being heavy kColdRounds passes ago are taken back — those are the moves reported, for the
// caller to migrate (at most `cap` of them per pass: what does not fit waits for the next).
// plan (p->mu held): the new exception list (and, with move_slots, the new slot table written in place); nothing published
static uint32_t plan_locked(guber_placement* p, double heavy_fraction, int move_slots, guber_placement_move_t* moves, uint32_t cap,
                            std::unique_ptr<Exceptions>& ne) {
    if (heavy_fraction <= 0) heavy_fraction = 0.125;
    uint64_t total = 0;
    for (uint32_t s = 0; s < p->n_slots; ++s) total += p->slot_w[s].load();
    ne.reset();
    if (total == 0) return 0;
    const double fair = (double)total / p->n_shards;
    const Exceptions* old = p->ex.load();
    // heavy hitters: the larger of the two rows' estimates per key
    struct Hot { uint64_t h; double w; };
    std::vector<Hot> hot;
    for (size_t k = 0; k < ((size_t)kSketchRows << kSketchBits); ++k) {
        const uint64_t h = p->sketch[k].h.load(); const uint32_t c = p->sketch[k].c.load();
        if (h == 0 || (double)c <= fair * heavy_fraction) continue;
        bool seen = false;
        for (auto& x : hot) if (x.h == h) { x.w = std::max(x.w, (double)c); seen = true; break; }
        if (!seen) hot.push_back({h, (double)c});
    }
    std::sort(hot.begin(), hot.end(), [](const Hot& a, const Hot& b) { return a.w != b.w ? a.w > b.w : a.h < b.h; });
    if (hot.size() > kMaxHot) hot.resize(kMaxHot);
    // slot weights without their heavy keys (+ a little per slot so that silent slots spread too)
    std::vector<double> sw(p->n_slots);
    for (uint32_t s = 0; s < p->n_slots; ++s) sw[s] = (double)p->slot_w[s].load();
    for (auto& x : hot) { double& w = sw[p->slot_of(x.h)]; w = std::max(0.0, w - x.w); }
    std::vector<double> load(p->n_shards, 0.0);
    ne.reset(new Exceptions());
    if (old) *ne = *old;
    uint32_t nm = 0;
    auto least = [&]() { return (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin()); };
    if (move_slots) {
        *ne = Exceptions();
        struct Item { double w; int kind; uint64_t id; };
        std::vector<Item> items;
        for (auto& x : hot) items.push_back({x.w, 0, x.h});
        const double eps = 0.05 * (double)total / p->n_slots;
        for (uint32_t s = 0; s < p->n_slots; ++s) items.push_back({sw[s] + eps, 1, s});
        std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.w != b.w ? a.w > b.w : (a.kind != b.kind ? a.kind < b.kind : a.id < b.id); });
        for (auto& it : items) {
            const uint32_t j = least();
            load[j] += it.w;
            if (it.kind == 0) {
                const uint32_t from = guber_placement_shard(p, it.id);
                ne->put(it.id, (uint16_t)j);
                if (moves && nm < cap) moves[nm] = guber_placement_move_t{it.id, from, j};
                nm++;
            } else p->table[it.id].store((uint16_t)j, std::memory_order_relaxed);
        }
    } else {
        // pins AGE: a key that has not been heavy for kColdRounds passes in a row (each over a window that saw real traffic) follows
        // its slot again — reported as a move back to the slot's shard, so that the caller migrates the bucket — and makes room for
        // the keys that are heavy now (without this a drifting hot set ends in kMaxHot stale pins and no isolation for new ones)
        const uint32_t room = moves ? cap : 0xffffffffu;                          // never more moves than the caller can take
        *ne = Exceptions();
        if (old) {
            for (uint32_t i = 0; i < kExCells; ++i) {
                const uint64_t h = old->h[i];
                if (h == 0) continue;
                bool heavy = false;
                for (auto& x : hot) if (x.h == h) { heavy = true; break; }
                const uint32_t c = heavy ? 0u : (total >= kMinWindow ? (uint32_t)old->cold[i] + 1u : (uint32_t)old->cold[i]);
                const uint32_t slot_shard = p->table[p->slot_of(h)].load();
                if (c >= kColdRounds && (slot_shard == old->s[i] || nm < room)) {
                    if (slot_shard != old->s[i]) { if (moves && nm < cap) moves[nm] = guber_placement_move_t{h, old->s[i], slot_shard}; nm++; }
                    continue;                                                      // the pin is gone
                }
                ne->put(h, old->s[i]);
                ne->set_cold(h, (uint8_t)std::min(c, 255u));
            }
        }
        for (uint32_t s = 0; s < p->n_slots; ++s) load[p->table[s].load()] += sw[s];
        for (auto& x : hot) { const int s = ne->get(x.h); if (s >= 0) load[s] += x.w; }     // already isolated: stays where it is
        for (auto& x : hot) {
            if (ne->get(x.h) >= 0 || ne->n >= kMaxHot) continue;
            if (nm >= room) break;
            const uint32_t from = p->table[p->slot_of(x.h)].load();
            const uint32_t j = least();
            load[j] += x.w;
            ne->put(x.h, (uint16_t)j);
            if (j == from) continue;                                                     // pinned where it is: no migration
            if (moves && nm < cap) moves[nm] = guber_placement_move_t{x.h, from, j};
            nm++;
        }
    }
    return nm;
}
// publish a planned list (p->mu held); the next round observes afresh
static void publish_locked(guber_placement* p, std::unique_ptr<Exceptions>& ne) {
    if (ne) {
        p->retired.push_back(std::move(ne));
        p->ex.store(p->retired.back().get(), std::memory_order_release);
        p->version.fetch_add(1, std::memory_order_acq_rel);
        if (p->retired.size() > kKeepRetired) p->retired.erase(p->retired.begin());   // (a reader holds a snapshot for a handful of instructions, not for 64 publishes)
    }
    for (uint32_t s = 0; s < p->n_slots; ++s) p->slot_w[s].store(0);
    for (size_t k = 0; k < ((size_t)kSketchRows << kSketchBits); ++k) { p->sketch[k].h.store(0); p->sketch[k].c.store(0); }
}

extern "C" int guber_placement_rebalance(guber_placement_t* p, double heavy_fraction, int move_slots, guber_placement_move_t* moves,
                                         uint32_t cap, uint32_t* n_moves) {
    if (n_moves) *n_moves = 0;
    if (!p) return GUBER_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    std::unique_ptr<Exceptions> ne;
    const uint32_t nm = plan_locked(p, heavy_fraction, move_slots, moves, cap, ne);
    p->pending.reset();
    publish_locked(p, ne);
    if (n_moves) *n_moves = nm;
    return (moves && nm > cap) ? GUBER_E_NOMEM : GUBER_OK;
}

// The same in two steps, for a caller that has to quiesce its shards between learning which resident keys move and letting
// requests follow the new placement (GPUWorkerPool): plan (move_slots = 0 semantics; nothing a reader sees changes), then
// — after the buckets have been migrated — commit.
extern "C" int guber_placement_plan(guber_placement_t* p, double heavy_fraction, guber_placement_move_t* moves, uint32_t cap, uint32_t* n_moves) {
    if (n_moves) *n_moves = 0;
    if (!p) return GUBER_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    const uint32_t nm = plan_locked(p, heavy_fraction, 0, moves, cap, p->pending);
    if (n_moves) *n_moves = nm;
    return (moves && nm > cap) ? GUBER_E_NOMEM : GUBER_OK;
}
// a planned move whose bucket could not be migrated: the key stays where the published placement has it — following its slot (a
// new pin is dropped) or pinned (a pin that had aged out is kept) — nothing a reader sees changes for it
extern "C" int guber_placement_cancel(guber_placement_t* p, uint64_t key_hash) {
    if (!p) return GUBER_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->pending) return GUBER_OK;
    const Exceptions* pub = p->ex.load();
    const int was = pub ? pub->get(key_hash) : -1;                              // where the published placement pins the key (-1: it follows its slot)
    if (p->pending->get(key_hash) == was) return GUBER_OK;                      // nothing planned for this key
    std::unique_ptr<guber_placement::Exceptions> ne(new guber_placement::Exceptions());
    for (uint32_t i = 0; i < guber_placement_detail::kExCells; ++i)
        if (p->pending->h[i] != 0 && p->pending->h[i] != key_hash) { ne->put(p->pending->h[i], p->pending->s[i]); ne->set_cold(p->pending->h[i], p->pending->cold[i]); }
    if (was >= 0) ne->put(key_hash, (uint16_t)was);                              // a pin that was to go stays (its bucket could not be moved back)
    p->pending = std::move(ne);
    return GUBER_OK;
}
extern "C" int guber_placement_commit(guber_placement_t* p) {
    if (!p) return GUBER_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    publish_locked(p, p->pending);
    return GUBER_OK;
}

// the published state as plain arrays + numbers: what a device needs to apply slot_of / Exceptions::get itself (guber_stage_route)
extern "C" int guber_placement_export(const guber_placement_t* p, guber_route_rule_t* out) {
    if (!p || !out) return GUBER_E_INVALID_ARG;
    static_assert(sizeof(std::atomic<uint16_t>) == sizeof(uint16_t), "the slot table is read as plain 16-bit words");
    *out = guber_route_rule_t{};
    out->n_shards = p->n_shards; out->per = p->per; out->step = p->step; out->inv_step = p->inv_step; out->inv_sub = p->inv_sub;
    out->table = reinterpret_cast<const uint16_t*>(p->table.get());
    out->ex_cells = kExCells; out->global_engine = -1;
    if (const Exceptions* e = p->ex.load(std::memory_order_acquire)) { out->ex_n = e->n; out->ex_hash = e->h; out->ex_shard = e->s; }
    return GUBER_OK;
}

extern "C" int guber_placement_info(const guber_placement_t* p, uint32_t* n_shards, uint32_t* n_slots, uint32_t* n_hot) {
    if (!p) return GUBER_E_INVALID_ARG;
    if (n_shards) *n_shards = p->n_shards;
    if (n_slots) *n_slots = p->n_slots;
    const Exceptions* e = p->ex.load();
    if (n_hot) *n_hot = e ? e->n : 0;
    return GUBER_OK;
}
