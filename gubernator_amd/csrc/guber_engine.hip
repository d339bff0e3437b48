// guber_engine.hip — host side of the MI355X rate-limit engine and the C ABI of include/guber_gpu.h.
// It owns the HBM-resident table, sequences the kernels of guber_kernels.h on one HIP stream and
// stages host batches through pinned memory.  It is the replacement for the reference's WorkerPool
// (workers.go:125-626): same operations, one call per batch instead of one channel hop per request.
// There is no CPU fallback: without a HIP device every entry point fails with GUBER_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/guber_gpu.h"
#include "guber_host.h"
#include "guber_kernels.h"
#include "guber_kernels_lru.h"
#include <hipcub/hipcub.hpp>

using namespace guber;

static thread_local std::string g_last_error;
static int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_last_error = buf;
    return code;
}
#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return fail(GUBER_E_HIP, #call, _e);          \
    } while (0)

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = std::max(n, (size_t)16);
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess) return fail(GUBER_E_NOMEM, "hipMalloc", e);
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <typename T>
struct PinBuf {
    T* p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = std::max(n, (size_t)16);
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return fail(GUBER_E_NOMEM, "hipHostMalloc", e);
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// host memory the kernels read and write in place (device-visible, coherent): no copy launches on the host-pointer path
template <typename T>
struct CohBuf {
    T* p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = std::max(n, (size_t)64);
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) return fail(GUBER_E_NOMEM, "hipHostMalloc(coherent)", e);
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct guber_engine;
static void ep_flush_held(const guber_engine* e);
struct guber_engine {
    int device = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    hipStream_t copy_in = nullptr; bool stage_dma = true;        // stages: DMA copies beside the kernels (guber_stage_submit)
    uint32_t stage_copy_min = 257;                                // guber_stages_submit: batches from this size on reach HBM through the copy kernel
    uint64_t slots = 0, cache_size = 0;
    uint32_t max_batch = 0, max_key = 0;
    Table T{};
    Work W{};
    // table + work storage
    DevBuf<DirEntry> dir; DevBuf<Bucket> buckets; DevBuf<uint8_t> arena; DevBuf<DevCounters> ctr;
    DevBuf<uint32_t> w_u32;    // all u32 work arrays carved from one allocation
    DevBuf<uint8_t> w_rflags; DevBuf<Rec> w_snap; DevBuf<uint32_t> w_hist; DevBuf<BlockCounters> bctr;
    CohBuf<BlockCounters> h_bctr; uint32_t n_bctr = 0;   // (device-visible: written by k_ctr_snapshot)
    // tile-bitmap grouping path (batches <= 65536)
    DevBuf<unsigned long long> w_tilemask; DevBuf<SegRec> w_srec; DevBuf<int64_t> w_sinv; DevBuf<uint16_t> w_tilerow;
    DevBuf<uint32_t> w_did2;
    // owner-partitioned pipeline (guber_kernels_part.h): messages tile -> owner, records owner -> tile, runs per (tile, owner),
    // tile maps of walked segments, the per-request words.  cap256 = fast_cap rounded up to whole tiles.
    DevBuf<GMsg> w_gmsg; DevBuf<GRec> w_grec; DevBuf<uint32_t> w_gse, w_did3, w_pmode; DevBuf<unsigned long long> w_segtiles;
    uint32_t cap256 = 0, part_min = 1024; bool use_part = true, part_single = false, force_part = false, fuse_ep = true; uint64_t part_batches = 0, ep_launches = 0;
    uint32_t fast_cap = 0;      // entries of the arrays above
    uint32_t fast_batches = 0, fast_prev_n = 0;
    bool force_radix = false;
    bool careful = false;       // retry rounds run without speculative claims
    bool always_careful = false;
    // GLOBAL pending queues
#ifdef GUBER_PHASE_TIMING
    DevBuf<unsigned long long> dbg; double dbg_avg[5][8] = {{0}}, dbg_max[5][8] = {{0}}; uint64_t dbg_n = 0, dbg_pn = 0;
#endif
    DevBuf<uint64_t> d_ring_h; DevBuf<uint32_t> d_ring_o; uint64_t ring_cached_id = 0; uint32_t ring_npts = 0;   // ring image for the *_dev routers
    DevBuf<ItemIn> d_items; DevBuf<uint32_t> d_islots; DevBuf<uint8_t> d_iflags, d_ikeys, d_ires;   // guber_add_items[_dev] scratch (persistent)
    DevBuf<uint8_t> d_lkey; DevBuf<Rec> d_lrec; DevBuf<int> d_lfound;                 // guber_get_item / guber_remove_item scratch
    DevBuf<uint64_t> d_mvh;                                                           // guber_move_items_by_hash: the hashes
    DevBuf<unsigned long long> w_claims; uint32_t claims_cells = 0; uint32_t fast_epoch16 = 0;   // k_front's per-batch claim table
    DevBuf<uint8_t> d_sflags; DevBuf<Rec> d_safter;   // Store side channel (guber_eval_batch_store), allocated on first use
    DevBuf<GPend> gpend; DevBuf<uint32_t> gdirty, gdirty2, gtake_ctr; DevBuf<uint8_t> d_take; PinBuf<uint8_t> h_take;
    // staging for the host-pointer entry points
    DevBuf<uint8_t> d_keys; DevBuf<uint32_t> d_off; DevBuf<int64_t> d_i64; DevBuf<uint32_t> d_beh; DevBuf<uint8_t> d_u8;
    DevBuf<int64_t> d_out64; DevBuf<uint8_t> d_out8;
    PinBuf<uint8_t> h_stage;   // one pinned arena for inputs and outputs (copy path)
    CohBuf<uint8_t> z_stage;   // device-visible arena of the zero-copy path: inputs, outputs, SmallOut
    DevBuf<int64_t> d_stash64; DevBuf<uint32_t> d_stash32; DevBuf<uint8_t> d_stash8;   // k_front's HBM copy of host-resident request columns
    hipEvent_t z_event = nullptr; uint32_t small_seq = 0; bool zero_copy = true, no_small = false;
    bool fuse = true; uint64_t fused_batches = 0;                 // guber_eval_batches_routed_dev: several engines per launch
    struct guber_stage* small_pending = nullptr;                  // a <= 256-request stage launched by guber_stages_submit whose outcome has not been looked at yet
    // guber_stages_submit, stages of several engines in one pair of launches: ONE completion event per group (a ring, owned by the
    // group's first engine; a slot is reused only after its previous use has completed, so "the slot has moved on" means
    // "complete"), and the device copy of the launches' argument blocks when these do not fit the kernel-argument segment
    struct GroupEv { hipEvent_t ev = nullptr; std::atomic<uint32_t> seq{0}; };
    static constexpr uint32_t kGroupEvs = 16;
    GroupEv gev[kGroupEvs]; uint32_t gev_next = 0;
    DevBuf<uint8_t> d_margs;
    uint64_t small_batches = 0, small_fallbacks = 0;
    DevBuf<uint16_t> d_rt_table, d_rt_exs; DevBuf<uint64_t> d_rt_exh; RouteRule rule{}; bool have_rule = false;   // guber_stage_route: the placement rule on the device
    CohBuf<DevCounters> h_ctr; CohBuf<uint32_t> h_rb_seq; uint32_t rb_seq = 0;   // counter snapshot + its completion stamp
    DevCounters last_ctr{};
    uint32_t epoch = 0;
    uint64_t batches = 0;
    uint64_t tags_upper = 0;   // host-side upper bound of ctr.tags_used
    uint64_t size_upper = 0;   // host-side upper bound of the live items (ctr.size)
    int64_t clock_ms = 0;      // latest `now` seen (guber_set_clock / batches / lookups): classifies evictions as expired or not
    uint64_t evict_passes = 0;
    // The recency order of LRUCache's list (lrucache.go:88-128) as request sequence numbers: a batch of n requests takes n stamps
    // (request i the i-th), Add one per item, GetItem one; every bucket carries the stamp of its last touch (rec_stamp, 53 bits).
    uint64_t seq_next = 1;     // the next stamp to hand out
    // The exact victim order when the cache binds (guber_kernels_lru.h): the tail list (live items sorted by stamp), the pre-pass's
    // control block and scratch.  Allocated by the first call that may overflow the cache.
    DevBuf<LruCtl> lru_ctl; PinBuf<LruCtl> lru_hctl;
    DevBuf<unsigned long long> lru_tstamp, lru_tstamp_in, lru_cnt; DevBuf<uint32_t> lru_tslot, lru_tslot_in; DevBuf<uint8_t> lru_sort_tmp;
    DevBuf<unsigned long long> lru_u64; DevBuf<uint32_t> lru_u32; DevBuf<uint8_t> lru_u8;
    bool lru_tail_ok = false;  // false: the slots' numbering or the table changed under the list (rebuild before use)
    uint64_t lru_admits = 0, lru_applied = 0, lru_rebuilds = 0, lru_cuts = 0, lru_passes = 0; uint32_t lru_split_at = 0;
    uint64_t touch = 0;        // first stamp of the call in progress (take_stamps)
    // asynchronous counter read-back (maintain): enqueued when an upper bound crosses its soft limit, folded when its event
    // has completed — the hot path never waits for it
    // a ring of snapshot slots (+ one reserved for the synchronous refresh): every slot remembers how many requests had been
    // enqueued when it was armed, so a completed snapshot gives  exact count as of then + requests enqueued since  as the bound
    static constexpr uint32_t kRb = 32;
    struct RbSlot { uint32_t seq = 0; uint64_t mark = 0; bool armed = false; };
    RbSlot rb[kRb + 1]; int rb_ride = -1;      // rb_ride: the slot that waits for a batch to ride on (k_front / k_part carry it)
    uint64_t added_total = 0;                  // requests (items) ever enqueued: each might have created an item and a directory entry
    uint64_t settle_waits = 0;
    uint64_t compactions = 0;
    std::mutex mu;
    // optional per-kernel timing (guber_profile_*)
    bool profiling = false;
    struct Span { int kernel; hipEvent_t a, b; hipStream_t st; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[24] = {0}; uint64_t prof_n[24] = {0}, prof_units[24] = {0};
    std::vector<float> group_us;   // per pipeline pass (the launches of one batch, or of one fused group): first kernel's start -> last kernel's end

    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t ev = event_pool.back(); event_pool.pop_back(); return ev; }
        hipEvent_t ev = nullptr; (void)hipEventCreate(&ev); return ev;
    }
    // (st: the stream the launch goes to when it is not the engine's own — the front's routing stream, guber_front.h)
    void span_begin(int k, uint64_t units = 0, hipStream_t st = nullptr) { if (profiling) { Span s{k, get_event(), get_event(), st ? st : stream}; (void)hipEventRecord(s.a, s.st); spans.push_back(s); prof_units[k] += units; } }
    void span_end() { if (profiling) (void)hipEventRecord(spans.back().b, spans.back().st); }

    // GUBER_FUSE_EP: the k_eval3 a routed call is holding back for a group this engine belongs to (guarded by mu).  Whoever is about to
    // enqueue on this engine, or to read what that launch writes, launches it first — every entry point comes through set_device()
    // after taking the lock; entry points that lock several engines call ep_flush_held on each (guber_engine.hip "held back")
    mutable struct PendingEval* held = nullptr;
    int set_device() const { if (held) ep_flush_held(this); return hipSetDevice(device) == hipSuccess ? 0 : -1; }
};

enum { KT_FRONT = 0, KT_EVAL2, KT_RESOLVE, KT_HIST, KT_SCATTER0, KT_SCATTER, KT_HEADS, KT_EVAL, KT_FRONT_MULTI, KT_EVAL2_MULTI,
       KT_PART, KT_OWN, KT_EVAL3, KT_PART_MULTI, KT_OWN_MULTI, KT_EVAL3_MULTI, KT_EVALPART_MULTI, KT_FR_COUNT, KT_FR_SCAN, KT_FR_SCATTER, KT_FR_OUT, KT_COUNT };
static_assert(KT_COUNT <= 24, "guber_engine::prof_* hold 24 kernels");
static const char* const kKernelNames[KT_COUNT] = {"k_front", "k_eval2", "k_resolve", "k_hist", "k_scatter(first)",
                                                   "k_scatter", "k_heads", "k_eval", "k_front_multi", "k_eval2_multi",
                                                   "k_part", "k_own", "k_eval3", "k_part_multi", "k_own_multi", "k_eval3_multi", "k_evalpart_multi",
                                                   "k_fr_count", "k_fr_scan", "k_fr_scatter", "k_fr_out"};

static uint64_t take_stamps(guber_engine* e, uint64_t n) {
    const uint64_t b = e->seq_next;
    e->seq_next += n ? n : 1;
    e->touch = b;
    return b;
}
static uint64_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

static void quiesce_all(guber_engine* e) { (void)hipStreamSynchronize(e->stream); }

// ---- the engine's counters on the host -----------------------------------------------------------------------------------------
// Snapshots of the device counters travel to device-visible host memory by a small launch of their own (k_ctr_snapshot) or riding
// on a batch's first kernel (Work::snap_*), each into its own slot with a sequence number stamped when it is complete: no copy
// engine, no event — the host just looks at the stamps.
static void note_enqueued(guber_engine* e, uint64_t n) { e->size_upper += n; e->tags_upper += n; e->added_total += n; }
static bool rb_slot_done(const guber_engine* e, uint32_t i) {
    return __atomic_load_n((volatile uint32_t*)&e->h_rb_seq.p[i], __ATOMIC_ACQUIRE) == e->rb[i].seq;
}
static bool rb_any_armed(const guber_engine* e) { for (uint32_t i = 0; i < guber_engine::kRb; ++i) if (e->rb[i].armed) return true; return false; }
// armed and already launched (a riding slot that has no batch yet is not on its way)
static bool rb_any_launched(const guber_engine* e) {
    for (uint32_t i = 0; i < guber_engine::kRb; ++i) if (e->rb[i].armed && (int)i != e->rb_ride) return true;
    return false;
}
static void rb_disarm_all(guber_engine* e) { for (auto& s : e->rb) s.armed = false; e->rb_ride = -1; }
// fold slot i (complete) into the host's image: the counters as of the snapshot, the bounds = that + what was enqueued since
static void rb_fold_slot(guber_engine* e, uint32_t i) {
    DevCounters c = e->h_ctr.p[i];
    const BlockCounters* hb = e->h_bctr.p + (size_t)i * e->n_bctr;
    for (uint32_t b = 0; b < e->n_bctr; ++b) { c.over += hb[b].over; c.hits += hb[b].hits; c.misses += hb[b].misses; c.size += hb[b].size_delta; }
    const uint64_t since = e->added_total - e->rb[i].mark;
    e->last_ctr = c;
    e->tags_upper = c.tags_used + since;
    e->size_upper = (uint64_t)std::max<long long>(c.size, 0) + since;
}
// the newest completed snapshot (if any) becomes the host's knowledge; every completed slot is free again
static bool rb_fold_newest(guber_engine* e) {
    int best = -1;
    for (uint32_t i = 0; i < guber_engine::kRb; ++i) {
        if (!e->rb[i].armed || (int)i == e->rb_ride || !rb_slot_done(e, i)) continue;
        if (best < 0 || e->rb[i].mark > e->rb[best].mark) best = (int)i;
    }
    if (best < 0) return false;
    rb_fold_slot(e, (uint32_t)best);
    for (uint32_t i = 0; i < guber_engine::kRb; ++i) if (e->rb[i].armed && (int)i != e->rb_ride && e->rb[i].mark <= e->rb[best].mark) e->rb[i].armed = false;
    return true;
}
static void rb_launch(guber_engine* e, uint32_t i) {
    hipLaunchKernelGGL(k_ctr_snapshot, dim3(1), dim3(256), 0, e->stream, e->ctr.p, e->bctr.p, e->n_bctr, e->h_ctr.p + i, e->h_bctr.p + (size_t)i * e->n_bctr,
                       e->h_rb_seq.p + i, e->rb[i].seq);
}
// arm a free slot: ride = the next batch's first kernel carries it (free of charge), else a launch of its own.  -1 = every slot is on its way.
static int rb_arm(guber_engine* e, bool ride) {
    if (ride && e->rb_ride >= 0) return e->rb_ride;
    for (uint32_t i = 0; i < guber_engine::kRb; ++i) {
        if (e->rb[i].armed) continue;
        e->rb[i].seq = ++e->rb_seq ? e->rb_seq : ++e->rb_seq;
        e->rb[i].mark = e->added_total; e->rb[i].armed = true;
        if (ride) e->rb_ride = (int)i; else rb_launch(e, i);
        return (int)i;
    }
    return -1;
}
// the riding snapshot goes with this batch (Work::snap_*)
static void attach_counter_readback(guber_engine* e, Work& W) {
    const uint32_t i = (uint32_t)e->rb_ride;
    W.snap_seq = e->rb[i].seq; W.snap_n = e->n_bctr; W.snap_c = e->h_ctr.p + i; W.snap_b = e->h_bctr.p + (size_t)i * e->n_bctr; W.snap_stamp = e->h_rb_seq.p + i;
    e->rb_ride = -1;
}
// synchronous: a snapshot at the tail of the stream into the reserved slot; after the stream has drained fold_counters makes it
// the host's knowledge (exact: nothing is in flight) and forgets every older snapshot
static int enqueue_counter_readback(guber_engine* e) {
    const uint32_t i = guber_engine::kRb;
    e->rb[i].seq = ++e->rb_seq ? e->rb_seq : ++e->rb_seq; e->rb[i].mark = e->added_total;
    rb_launch(e, i);
    HIPCHK(hipGetLastError());
    return 0;
}
static void fold_counters(guber_engine* e) {
    rb_fold_slot(e, guber_engine::kRb);
    rb_disarm_all(e);
}
static int engine_refresh_counters(guber_engine* e) {
    int rc = enqueue_counter_readback(e);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    fold_counters(e);
    return 0;
}

extern "C" int guber_engine_create(const guber_config_t* cfg, guber_engine_t** out) {
    if (!cfg || !out) return fail(GUBER_E_INVALID_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GUBER_E_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(GUBER_E_INVALID_ARG, "device ordinal out of range");
    guber_engine* e = new guber_engine();
    e->device = cfg->device;
    if (hipSetDevice(e->device) != hipSuccess) { delete e; return fail(GUBER_E_HIP, "hipSetDevice"); }
    e->cache_size = cfg->cache_size ? cfg->cache_size : 50000;  // workers.go:126
    e->max_batch = cfg->max_batch ? cfg->max_batch : 65536;
    // the cache may hold cache_size items when a batch of max_batch new keys arrives (eviction runs between batches): both fit
    // under the directory's load limit
    // ... with room to spare where it is cheap: below a directory load of 0.25 nine keys in ten sit at their home position (one
    // trip); measured on MI355X, 12 tables of 0.83 M keys: 2^22 slots each (load 0.2) 7.3, 2^23 7.6, 2^24 7.8 G decisions/s, 2^21 6.9.
    // Tables up to 4 GB take the factor 4 (144 bytes per slot), larger ones stay at 2.
    {
        const uint64_t need = e->cache_size + std::min<uint64_t>(e->max_batch, 1u << 20);
        uint64_t slots = next_pow2(std::max<uint64_t>(4 * need, 1024));
        if (slots * (sizeof(Bucket) + sizeof(DirEntry)) > (4ull << 30)) slots = next_pow2(std::max<uint64_t>(2 * need, 1024));
        e->slots = cfg->table_slots ? next_pow2(cfg->table_slots) : slots;
    }
    if (e->slots > (1ull << 32)) { delete e; return fail(GUBER_E_INVALID_ARG, "table_slots above 2^32"); }
    if (e->max_batch > (1u << 24)) { delete e; return fail(GUBER_E_INVALID_ARG, "max_batch above 2^24"); }
    e->max_key = cfg->max_key_bytes ? cfg->max_key_bytes : 1024;
    if (e->max_key > 65000) e->max_key = 65000;
    if (cfg->stream) { e->stream = (hipStream_t)cfg->stream; e->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return fail(GUBER_E_HIP, "hipStreamCreate"); }
        e->own_stream = true;
    }
    int rc = 0;
    const uint64_t arena_cap = std::max<uint64_t>(e->slots * 16, 1 << 20);   // long-key overflow arena
    rc |= e->dir.ensure(e->slots); rc |= e->buckets.ensure(e->slots);
    rc |= e->arena.ensure(arena_cap + 64); rc |= e->ctr.ensure(1); rc |= e->h_ctr.ensure(1);
    const uint32_t M = e->max_batch;
    const uint32_t tiles = (M + TILE - 1) / TILE;
    rc |= e->w_u32.ensure((size_t)M * 14);
    rc |= e->w_rflags.ensure(M); rc |= e->w_snap.ensure(M);
    rc |= e->w_hist.ensure((size_t)MAX_PASSES * RADIX * tiles);
    e->fast_cap = std::min<uint32_t>(M, FT * FT_MAX_TILES);
    e->force_radix = (cfg->flags & GUBER_FLAG_TEST_FORCE_RADIX) != 0;
    e->always_careful = (cfg->flags & GUBER_FLAG_TEST_CAREFUL) != 0;
    e->no_small = (cfg->flags & (GUBER_FLAG_TEST_NO_SMALL | GUBER_FLAG_TEST_FORCE_PART)) != 0 || e->force_radix || e->always_careful;
    e->zero_copy = getenv("GUBER_NO_ZEROCOPY") == nullptr;
    e->fuse = getenv("GUBER_NO_FUSE") == nullptr;
    e->stage_dma = getenv("GUBER_NO_STAGE_DMA") == nullptr;
    if (const char* v = getenv("GUBER_STAGE_COPY_MIN")) e->stage_copy_min = (uint32_t)atoi(v);
    e->cap256 = (e->fast_cap + FT - 1) / FT * FT;
    rc |= e->w_tilemask.ensure((size_t)2 * e->fast_cap * FT_WORDS); rc |= e->w_srec.ensure(e->fast_cap); rc |= e->w_sinv.ensure(e->cap256);
    rc |= e->w_tilerow.ensure((size_t)e->cap256 * FT_MAX_TILES);
    e->force_part = (cfg->flags & GUBER_FLAG_TEST_FORCE_PART) != 0;
    e->use_part = !(cfg->flags & GUBER_FLAG_NO_PART) && !e->force_radix && !e->always_careful;
    // GUBER_PIPELINE: "claims" = never the owner-partitioned pipeline, "part" = also for a batch launched on its own; default: the
    // owner-partitioned pipeline when several tables share the launches (where it is faster: profiles/r04_*), claims otherwise
    if (const char* v = getenv("GUBER_PIPELINE")) { if (!strcmp(v, "claims")) e->use_part = false; else if (!strcmp(v, "part")) e->part_single = true; }
    if (const char* v = getenv("GUBER_PART_MIN")) e->part_min = (uint32_t)std::max(257, atoi(v));
    // Inside one guber_eval_batches_routed_dev call a group's k_eval3 shares a launch with the k_part of the same tables' next batches
    // (k_evalpart_multi, guber_kernels_part.h: two launches per pass instead of three; measured in round 5 on the headline, same box,
    // alternating: 8.69 -> 8.99 G decisions/s, profiles/r05_a_fuse_ep_ab.txt).  Another thread's call on one of the group's engines
    // launches the held-back k_eval3 first (guber_engine::held).  GUBER_FUSE_EP=0 is the switch for an A/B on another box.
    if (const char* v = getenv("GUBER_FUSE_EP")) e->fuse_ep = atoi(v) != 0;
    if (e->force_part) e->part_min = 1;
    rc |= e->w_gmsg.ensure(e->cap256); rc |= e->w_grec.ensure((size_t)e->cap256 + e->cap256 / 2); rc |= e->w_gse.ensure((size_t)FT_MAX_TILES * PT_PARTS);   // (grec: 32-byte records first, then the 64-byte form)
    rc |= e->w_did3.ensure((size_t)e->cap256 * (e->fuse_ep ? 2 : 1)); rc |= e->w_segtiles.ensure((size_t)e->cap256 * 4); rc |= e->w_pmode.ensure(16);
    rc |= e->w_did2.ensure((size_t)2 * e->fast_cap);
    e->claims_cells = 1024;
    while (e->claims_cells < 4 * e->fast_cap) e->claims_cells <<= 1;   // load <= 0.25: short probe chains, 2 MB at 65 536
    rc |= e->w_claims.ensure(e->claims_cells);
    uint32_t gdirty_cap = 0;
    if (cfg->flags & GUBER_FLAG_GLOBAL) {
        gdirty_cap = (uint32_t)std::min<uint64_t>(e->slots, 1u << 24);
        rc |= e->gpend.ensure(e->slots); rc |= e->gdirty.ensure(gdirty_cap); rc |= e->gdirty2.ensure(gdirty_cap);
        rc |= e->gtake_ctr.ensure(4);
    }
    e->n_bctr = (M + 255) / 256;
    rc |= e->bctr.ensure(e->n_bctr); rc |= e->h_bctr.ensure((size_t)e->n_bctr * (guber_engine::kRb + 1));
    rc |= e->h_ctr.ensure(guber_engine::kRb + 1); rc |= e->h_rb_seq.ensure(guber_engine::kRb + 1);
    if (!rc) memset(e->h_rb_seq.p, 0, (guber_engine::kRb + 1) * sizeof(uint32_t));
    if (rc) { guber_engine_destroy(e); return GUBER_E_NOMEM; }
    hipError_t he = hipSuccess;
    // owners per batch of the owner-partitioned pipeline: starts at 128 and follows the traffic on the device (guber_kernels_part.h
    // "HOW MANY OWNERS"); GUBER_PT_BITS=7|8 pins it (measurements, tests)
    uint32_t pm0[8] = {7u, 0u, 0u, 0u, 7u, 7u, 0u, 0u};              // ([4..5]: the bits per batch parity of a GUBER_FUSE_EP engine, Work::pmslot)
    if (const char* v = getenv("GUBER_PT_BITS")) { const int b = atoi(v); if (b == 7 || b == 8) { pm0[0] = pm0[4] = pm0[5] = (uint32_t)b; pm0[3] = 1u; } }
    if ((he = hipMemsetAsync(e->dir.p, 0, e->slots * sizeof(DirEntry), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->buckets.p, 0, e->slots * sizeof(Bucket), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->ctr.p, 0, sizeof(DevCounters), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->bctr.p, 0, e->n_bctr * sizeof(BlockCounters), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->w_tilemask.p, 0, (size_t)2 * e->fast_cap * FT_WORDS * 8, e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->w_srec.p, 0, (size_t)e->fast_cap * sizeof(SegRec), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->w_did2.p, 0, (size_t)2 * e->fast_cap * 4, e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->w_segtiles.p, 0, (size_t)e->cap256 * 4 * 8, e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->w_gse.p, 0, (size_t)FT_MAX_TILES * PT_PARTS * 4, e->stream)) != hipSuccess ||
        (he = hipMemcpyAsync(e->w_pmode.p, pm0, sizeof(pm0), hipMemcpyHostToDevice, e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->w_claims.p, 0, (size_t)e->claims_cells * 8, e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(e->w_u32.p, 0, (size_t)M * 14 * 4, e->stream)) != hipSuccess ||
        (e->gpend.p && (he = hipMemsetAsync(e->gpend.p, 0, e->slots * sizeof(GPend), e->stream)) != hipSuccess) ||
        (he = hipStreamSynchronize(e->stream)) != hipSuccess) {
        guber_engine_destroy(e);
        return fail(GUBER_E_HIP, "table initialisation", he);
    }
    e->T.gpend = e->gpend.p; e->T.gdirty = e->gdirty.p; e->T.gdirty_cap = gdirty_cap;
    e->T.dir = e->dir.p; e->T.buckets = e->buckets.p; e->T.arena = e->arena.p;
    e->T.mask = e->slots - 1; e->T.arena_cap = arena_cap; e->T.ctr = e->ctr.p; e->T.bctr = e->bctr.p;
    e->T.max_probe = (uint32_t)std::min<uint64_t>(e->slots, 1u << 12); e->T.max_key = e->max_key;
    e->T.hash_mask = (cfg->flags & GUBER_FLAG_TEST_WEAK_HASH) ? 0x1f80ull : ~0ull;   // 6 significant bits
    uint32_t* u = e->w_u32.p;
    uint32_t** fields[] = {&e->W.slot, &e->W.did, &e->W.keyA, &e->W.valA, &e->W.keyB, &e->W.valB, &e->W.pos,
                           &e->W.order, &e->W.sdid, &e->W.seg_first, &e->W.seg_last, &e->W.seg_flags, &e->W.seg_rep,
                           &e->W.seg_slot};
    for (auto f : fields) { *f = u; u += M; }
    e->W.rflags = e->w_rflags.p; e->W.snap = e->w_snap.p;
    e->W.hist = e->w_hist.p;
    e->W.tiles = tiles; e->W.epoch = 0;
    e->W.seg_tilemask = e->w_tilemask.p; e->W.srec = e->w_srec.p; e->W.sinv = e->w_sinv.p; e->W.tilerow = e->w_tilerow.p;

    e->W.snap_seq = 0; e->W.snap_n = 0; e->W.snap_c = nullptr; e->W.snap_b = nullptr; e->W.snap_stamp = nullptr;
    e->W.parity = 0; e->W.clear_n = 0; e->W.store_flags = nullptr; e->W.store_after = nullptr;
    e->W.claims = e->w_claims.p; e->W.cmask = e->claims_cells - 1; e->W.epoch16 = 0;
    e->W.gmsg = e->w_gmsg.p;
    e->W.grs = (GRecS*)e->w_grec.p; e->W.grec = e->w_grec.p + e->cap256 / 2; e->W.gse = e->w_gse.p; e->W.segtiles = e->w_segtiles.p; e->W.pmode = e->w_pmode.p;
    { uint32_t lg = 0; while ((1ull << lg) < e->slots) ++lg; e->W.pshift = lg - 8; e->W.pmslot = 0; }   // (slots >= 1024)
#ifdef GUBER_PHASE_TIMING
    (void)e->dbg.ensure(4096 + 3 * 2048);
#endif
    *out = e;
    return GUBER_OK;
}

extern "C" void guber_engine_destroy(guber_engine_t* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (getenv("GUBER_ENGINE_STATS"))
        fprintf(stderr, "[engine %p] batches %llu (small %llu fused %llu part %llu, k_evalpart launches %llu) cache_size %llu size_upper %llu last size %lld | eviction pre-passes: calls %llu launches %llu applied %llu cuts %llu tail rebuilds %llu | waits for a snapshot %llu | compactions %llu\n",
                (void*)e, (unsigned long long)e->batches, (unsigned long long)e->small_batches, (unsigned long long)e->fused_batches, (unsigned long long)e->part_batches, (unsigned long long)e->ep_launches,
                (unsigned long long)e->cache_size, (unsigned long long)e->size_upper, (long long)e->last_ctr.size, (unsigned long long)e->lru_admits, (unsigned long long)e->lru_passes,
                (unsigned long long)e->lru_applied, (unsigned long long)e->lru_cuts, (unsigned long long)e->lru_rebuilds, (unsigned long long)e->settle_waits, (unsigned long long)e->compactions);
#ifdef GUBER_PHASE_TIMING
    if (e->dbg_n) {
        static const char* names[2][8] = {{"entry", "grouped in LDS", "claimed + published", "requests compared", "probe + verify done", "end (all drained)", "key_off loaded", "key hashed"},
                                          {"entry", "loads issued+prepass", "after barrier", "eval done", "end", "", "", ""}};
        for (int kern = 0; kern < 2; ++kern) {
            fprintf(stderr, "[phase timing] %s over %llu full batches (us since the first workgroup's entry: avg over workgroups / last workgroup)\n",
                    kern ? "k_eval2" : "k_front", (unsigned long long)e->dbg_n);
            for (int k = 0; k < (kern ? 5 : 8); ++k)
                fprintf(stderr, "    %-24s %7.2f / %7.2f\n", names[kern][k], e->dbg_avg[kern][k] / e->dbg_n, e->dbg_max[kern][k] / e->dbg_n);
        }
    }
    if (e->dbg_pn) {
        static const char* kn[3] = {"k_part", "k_own", "k_eval3"};
        static const char* names[3][8] = {{"entry", "fields + key hashed", "grouped in LDS", "members compared", "owner runs scanned", "end (all drained)", "", ""},
                                          {"runs read (gse)", "scanned", "gathered + keys installed", "table loads issued", "messages compared", "bases done", "table done (records in LDS)", "end (all drained)"},
                                          {"entry", "word + record + fields loaded", "evaluated", "end (all drained)", "", "", "", ""}};
        static const int ns[3] = {6, 8, 4};
        for (int kern = 0; kern < 3; ++kern) {
            fprintf(stderr, "[phase timing] %s over %llu full batches (us since the first workgroup's entry: avg over workgroups / last workgroup)\n", kn[kern], (unsigned long long)e->dbg_pn);
            for (int k = 0; k < ns[kern]; ++k)
                fprintf(stderr, "    %-30s %7.2f / %7.2f\n", names[kern][k], e->dbg_avg[2 + kern][k] / e->dbg_pn, e->dbg_max[2 + kern][k] / e->dbg_pn);
        }
    }
    e->dbg.release();
#endif
    e->d_ring_h.release(); e->d_ring_o.release(); e->d_items.release(); e->d_islots.release(); e->d_iflags.release(); e->d_ikeys.release(); e->d_ires.release();
    e->d_lkey.release(); e->d_lrec.release(); e->d_lfound.release(); e->d_mvh.release();
    e->w_claims.release();
    e->d_sflags.release(); e->d_safter.release();
    e->gpend.release(); e->gdirty.release(); e->gdirty2.release(); e->gtake_ctr.release();
    e->d_take.release(); e->h_take.release();
    e->dir.release(); e->buckets.release(); e->arena.release(); e->ctr.release();
    e->w_u32.release(); e->w_rflags.release(); e->w_snap.release(); e->w_hist.release();
    e->bctr.release(); e->h_bctr.release();
    e->w_tilemask.release(); e->w_srec.release(); e->w_sinv.release(); e->w_tilerow.release();
    e->w_did2.release();
    e->w_gmsg.release(); e->w_grec.release(); e->w_gse.release(); e->w_did3.release(); e->w_pmode.release(); e->w_segtiles.release();
    e->d_keys.release(); e->d_off.release(); e->d_i64.release(); e->d_beh.release(); e->d_u8.release();
    e->d_out64.release(); e->d_out8.release(); e->h_stage.release(); e->h_ctr.release(); e->h_rb_seq.release(); e->z_stage.release();
    e->d_stash64.release(); e->d_stash32.release(); e->d_stash8.release();
    if (e->z_event) (void)hipEventDestroy(e->z_event);
    for (auto& g : e->gev) if (g.ev) (void)hipEventDestroy(g.ev);
    e->d_margs.release();
    e->lru_ctl.release(); e->lru_hctl.release(); e->lru_tstamp.release(); e->lru_tstamp_in.release(); e->lru_cnt.release(); e->lru_tslot.release();
    e->lru_tslot_in.release(); e->lru_sort_tmp.release(); e->lru_u64.release(); e->lru_u32.release(); e->lru_u8.release();
    if (e->copy_in) (void)hipStreamDestroy(e->copy_in);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// Enqueue the kernel sequence for one batch whose arrays are all in HBM.
static int compact_table(guber_engine* e, int64_t now_ms);
static int maintain(guber_engine* e, uint64_t incoming, int64_t now_ms, bool batch_follows = false, bool* defer_hard = nullptr);
// What a batch needs before its kernels can be enqueued, shared by the single-engine and the fused multi-engine launch:
// cache maintenance, the engine's epochs, and (two-launch pipeline) the views / work arrays of this batch.
struct FastPlan { BatchView B2, B3; Work W; uint32_t ftiles; };
static bool takes_fast_path(const guber_engine* e, uint32_t n) { return n != 0 && n <= e->fast_cap && !e->force_radix; }
// the owner-partitioned pipeline (three launches, guber_kernels_part.h): batches the coordination between tiles is worth it for,
// in HBM (k_part and k_eval3 both read the request columns), outside retry rounds (those verify before they group)
static bool takes_part_path(const guber_engine* e, uint32_t n, bool host_resident, bool fused) {
    return takes_fast_path(e, n) && e->use_part && !e->careful && n >= e->part_min && (!host_resident || e->force_part) &&
           (fused || e->part_single || e->force_part);
}

// ---- the bounded cache's exact victim order (guber_kernels_lru.h) --------------------------------------------------------------
// May this call make the cache longer than cache_size?  size_upper is the host's upper bound of the live items (every request
// might create one); lru_admit reads the exact figure when it matters.
// When the bound says yes, the bound is first brought up to date: it counts every request in flight as a new item, so the host
// looks at the counter snapshots that ride on the batches (maintain) and, as long as one is on its way, waits for the GPU to get
// there — a wait for PROGRESS, not a drain: the queue stays as deep as the cache's headroom allows.  Only when nothing is left to
// wait for is the answer yes (the pre-pass then synchronises and sees the exact figure).  Engine mutex held.
static bool lru_may_bind(guber_engine* e, uint64_t n) {
    if (e->size_upper + n <= e->cache_size) return false;
    bool waited = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (rb_fold_newest(e) && e->size_upper + n <= e->cache_size) { e->settle_waits += waited; return false; }
        if (!rb_any_launched(e)) break;
        waited = true;
        if ((spins & 0xff) == 0xff) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
            std::this_thread::yield();                           // (the engine mutex is held: whoever else needs this CPU gets it)
        }
        __builtin_ia32_pause();
    }
    e->settle_waits += waited;
    return true;
}
static bool lru_may_bind_unlocked(guber_engine* e, uint64_t n) { std::lock_guard<std::mutex> lk(e->mu); return lru_may_bind(e, n); }
static LruKeys lru_keys_of(const BatchView& B) {
    LruKeys K{};
    K.bytes = B.key_bytes; K.algorithm = B.algorithm; K.key_stride = B.key_stride;
    K.behavior = B.behavior; K.duration = B.duration; K.greg_duration = (B.greg_expire && B.greg_duration) ? B.greg_duration : nullptr;
    if (!B.key_stride) { K.off_p = (const uint8_t*)B.key_off; K.off_stride = 4; }
    if (B.key_stride || B.key_len) { K.len_p = (const uint8_t*)B.key_len; K.len_stride = 4; }
    return K;
}
// The tail list: every live item's (stamp, slot), sorted by stamp — one table scan and one radix sort, then good for as many
// batches as it has valid entries left (an entry is valid while its bucket still carries that stamp).
static int lru_rebuild(guber_engine* e) {
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    const uint64_t live = (uint64_t)std::max<long long>(e->last_ctr.size, 0);
    const uint64_t cap = live + 64;
    if (e->lru_ctl.ensure(1) || e->lru_hctl.ensure(1) || e->lru_cnt.ensure(1) || e->lru_tstamp.ensure(cap) || e->lru_tstamp_in.ensure(cap) ||
        e->lru_tslot.ensure(cap) || e->lru_tslot_in.ensure(cap)) return GUBER_E_NOMEM;
    hipStream_t st = e->stream;
    HIPCHK(hipMemsetAsync(e->lru_cnt.p, 0, 8, st));
    hipLaunchKernelGGL(k_lru_gather, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, st, e->T, e->slots, e->lru_tstamp_in.p, e->lru_tslot_in.p, cap, e->lru_cnt.p);
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, e->lru_cnt.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (cnt > cap) return fail(GUBER_E_HIP, "the table holds more live items than its counters say");
    if (cnt) {
        size_t tmp = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, e->lru_tstamp_in.p, e->lru_tstamp.p, e->lru_tslot_in.p, e->lru_tslot.p, (int)cnt, 0, 53, st));
        if (e->lru_sort_tmp.ensure(tmp + 16)) return GUBER_E_NOMEM;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(e->lru_sort_tmp.p, tmp, e->lru_tstamp_in.p, e->lru_tstamp.p, e->lru_tslot_in.p, e->lru_tslot.p, (int)cnt, 0, 53, st));
    }
    LruCtl c{}; c.cursor = 0; c.tail_n = cnt;
    HIPCHK(hipMemcpyAsync(e->lru_ctl.p, &c, sizeof(c), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    e->lru_tail_ok = true; e->lru_rebuilds++;
    return 0;
}
// The pre-pass of a call that may overflow the cache: n requests (or n = 0: only bring the cache down to cache_size).  On return
// *status is LRU_NONE / LRU_APPLIED (the buckets that leave are absent, the counters adjusted: evaluate the batch) or LRU_CUT
// (nothing done: the batch is larger than the cache and evictions are due — the caller evaluates it in pieces of cache_size).
static int lru_admit(guber_engine* e, const LruKeys& K, uint32_t n, int64_t now_ms, uint32_t* status) {
    hipStream_t st = e->stream;
    if (e->lru_ctl.ensure(1) || e->lru_hctl.ensure(1)) return GUBER_E_NOMEM;
    if (!e->lru_tstamp.p) {                                          // first use: an empty list (the first check asks for a real one)
        LruCtl c{};
        HIPCHK(hipMemcpyAsync(e->lru_ctl.p, &c, sizeof(c), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        if (e->lru_tstamp.ensure(16) || e->lru_tslot.ensure(16)) return GUBER_E_NOMEM;
        e->lru_tail_ok = true;                                       // (valid and empty)
    }
    uint32_t cells = 1024; while (cells < 2 * (uint64_t)n) cells <<= 1;
    e->lru_admits++;
    uint64_t w_len = std::max<uint64_t>(2 * (uint64_t)n, 4096);
    for (int round = 0; round < 64; ++round) {
        if (!e->lru_tail_ok) { const int rc = lru_rebuild(e); if (rc) return rc; }
        const uint64_t live_cap = e->lru_tstamp.cap;
        if (w_len > live_cap) w_len = live_cap;
        const uint32_t W = (uint32_t)std::min<uint64_t>(w_len, 1u << 30), wblocks = (W + 255) / 256;
        // scratch: u64 [cells gid | n rstamp | W zstamp], u32 [cells gfirst | cells gfirst_ok | cells gfirst_reset | n rfirst | n rslot | W zslot | W zwidx | n qfirst | n qrank | n qslot |
        //               wblocks + 1 blockcnt | n + 1 new_before | W + 1 touched_before | 4 n_risk], u8 [n + 1 isnew_at | W wflag | W ztouched]
        const size_t nn = (size_t)n + 1;
        if (e->lru_u64.ensure((size_t)cells + nn + W + 8) || e->lru_u32.ensure(3 * (size_t)cells + 6 * nn + 3 * ((size_t)W + 1) + wblocks + 16) ||
            e->lru_u8.ensure(nn + 2 * ((size_t)W + 1) + 64)) return GUBER_E_NOMEM;
        unsigned long long* p64 = e->lru_u64.p; uint32_t* p32 = e->lru_u32.p; uint8_t* p8 = e->lru_u8.p;
        LruGroups G{p64, p32, p32 + cells, p32 + 2 * (size_t)cells, cells - 1}; p64 += cells; p32 += 3 * (size_t)cells;
        LruRes R{p32, p32 + nn, p64}; p32 += 2 * nn; p64 += nn;
        LruWin Z{p64, p32, p32 + W + 1}; p64 += W; p32 += 2 * ((size_t)W + 1);
        LruRisk Q{p32, p32 + nn, p32 + 2 * nn}; p32 += 3 * nn;
        uint32_t* blockcnt = p32; p32 += wblocks + 1;
        uint32_t* new_before = p32; p32 += nn;
        uint32_t* touched_before = p32; p32 += (size_t)W + 1;
        uint32_t* n_risk = p32;
        uint8_t* isnew_at = p8; uint8_t* wflag = p8 + nn; uint8_t* ztouched = wflag + W + 1;
        LruCtl* C = e->lru_ctl.p;
        hipLaunchKernelGGL(k_lru_begin, dim3(1), dim3(256), 0, st, e->T, C, e->n_bctr);
        if (n) {
            HIPCHK(hipMemsetAsync(G.id, 0xff, (size_t)cells * 8, st));
            HIPCHK(hipMemsetAsync(G.first, 0xff, (size_t)cells * 12, st));                  // (first, first_ok and first_reset)
            HIPCHK(hipMemsetAsync(isnew_at, 0, nn, st));
            hipLaunchKernelGGL(k_lru_probe, dim3((n + 255) / 256), dim3(256), 0, st, e->T, K, n, G);
            hipLaunchKernelGGL(k_lru_keys, dim3(cells / 256), dim3(256), 0, st, e->T, G, C, isnew_at, R);
        }
        HIPCHK(hipMemsetAsync(ztouched, 0, (size_t)W + 1, st));
        HIPCHK(hipMemsetAsync(n_risk, 0, 4, st));
        if (W) {
            hipLaunchKernelGGL(k_lru_win_flag, dim3(wblocks), dim3(256), 0, st, e->T, e->lru_tstamp.p, e->lru_tslot.p, C, W, wflag, blockcnt);
            hipLaunchKernelGGL(k_lru_scan_u32, dim3(1), dim3(1024), 0, st, blockcnt, wblocks, &C->win_valid);
            hipLaunchKernelGGL(k_lru_win_emit, dim3(wblocks), dim3(256), 0, st, e->lru_tstamp.p, e->lru_tslot.p, C, W, wflag, blockcnt, Z);
        }
        hipLaunchKernelGGL(k_lru_check, dim3(1), dim3(1), 0, st, C, W, n, e->cache_size);
        if (W) {
            if (n) {
                hipLaunchKernelGGL(k_lru_risk, dim3((n + 255) / 256), dim3(256), 0, st, C, R, Z, ztouched, Q, n_risk);
                hipLaunchKernelGGL(k_lru_scan_u8, dim3(1), dim3(1024), 0, st, isnew_at, n, new_before);
            }
            hipLaunchKernelGGL(k_lru_scan_u8, dim3(1), dim3(1024), 0, st, ztouched, W, touched_before);
            if (n) hipLaunchKernelGGL(k_lru_decide, dim3((n + 255) / 256), dim3(256), 0, st, e->T, C, e->cache_size, Q, n_risk, new_before, now_ms);
            hipLaunchKernelGGL(k_lru_evict, dim3(wblocks), dim3(256), 0, st, e->T, C, Z, ztouched, touched_before, now_ms);
            hipLaunchKernelGGL(k_lru_end, dim3(1), dim3(1), 0, st, e->T, C, Z);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(e->lru_hctl.p, C, sizeof(LruCtl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const LruCtl& c = *e->lru_hctl.p;
        e->lru_passes++;
        if (c.status == LRU_NONE || c.status == LRU_APPLIED) {
            const long long left = c.len0 - (long long)c.evicted;
            e->size_upper = (uint64_t)std::max<long long>(left, 0);
            rb_disarm_all(e);                                        // (the stream has drained: every snapshot on its way is older news)
            e->last_ctr.size = left; e->last_ctr.evictions += c.unexpired;
            if (c.status == LRU_APPLIED) e->lru_applied++;
            *status = c.status;
            return 0;
        }
        if (c.status == LRU_CUT) { e->lru_cuts++; *status = LRU_CUT; return 0; }
        if (c.status == LRU_SPLIT) { e->lru_cuts++; e->lru_split_at = c.split_at; *status = LRU_SPLIT; return 0; }
        if (c.status == LRU_MORE) { w_len *= 4; continue; }
        if (c.status == LRU_REBUILD) { e->lru_tail_ok = false; w_len = std::max<uint64_t>(w_len, 2 * (uint64_t)n + c.zone); continue; }
        return fail(GUBER_E_HIP, "the eviction pre-pass left no verdict");
    }
    return fail(GUBER_E_HIP, "the eviction pre-pass did not converge");
}

static int batch_prelude(guber_engine* e, const BatchView& B, Work& W, bool* defer_hard = nullptr) {
    const uint32_t n = B.n;
    if (n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    if (defer_hard && e->epoch + 1 >= 0x7fffffffu) { *defer_hard = true; return 0; }   // (the wrap below enqueues a launch)
    if (B.now_ms > e->clock_ms) e->clock_ms = B.now_ms;
    // Bounded cache and directory load.  size_upper / tags_upper are host-side upper bounds (every request might create a
    // new item); only when one crosses its limit are the real counters read back, the least recently used items evicted
    // (lrucache.go:98-100) and, if the directory is above its load limit, the table rebuilt without its dead entries.  A
    // batch that still finds no room gets per-item GUBER_ITEM_E_TABLE_FULL from the bounded probe, for NEW keys only.
    {
        const int rc = maintain(e, n, B.now_ms, takes_fast_path(e, n), defer_hard);
        if (rc) return rc;
        if (defer_hard && *defer_hard) return 0;             // (nothing has been done: the caller comes back)
    }
    note_enqueued(e, n);
    if (++e->epoch >= 0x7fffffffu) {   // 31-bit epoch wrapped: drop all dense-id claims
        hipLaunchKernelGGL(k_clear_claims, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, e->stream, e->T, e->slots);
        e->epoch = 1;
    }
    W = e->W;
    W.epoch = e->epoch;
    W.touch = take_stamps(e, n);
    W.tiles = (n + TILE - 1) / TILE;
    return 0;
}

static int plan_fast(guber_engine* e, const BatchView& B, bool host_resident, Work& W, FastPlan& P) {
    const uint32_t n = B.n;
    BatchView B2 = B;
    B2.n_cap = e->fast_cap;
    W.careful = (e->careful || e->always_careful) ? 1u : 0u;
    W.snap_seq = 0;
    if (e->rb_ride >= 0) attach_counter_readback(e, W);
    if (++e->fast_epoch16 > 0xffffu) {   // 16-bit claim epoch wrapped: forget every cell
        HIPCHK(hipMemsetAsync(e->w_claims.p, 0, (size_t)e->claims_cells * 8, e->stream));
        HIPCHK(hipMemset2DAsync(&e->w_srec.p[0].flags, sizeof(SegRec), 0, sizeof(unsigned long long), e->fast_cap, e->stream));   // epoch-tagged flag words
        e->fast_epoch16 = 1;
    }
    W.epoch16 = e->fast_epoch16;
    {   // the batch's share of the claim table: 4 cells per request (k_eval2 zeroes exactly that part again)
        uint32_t cells = 1024;
        while (cells < 4 * n && cells < e->claims_cells) cells <<= 1;
        W.cmask = cells - 1;
    }
    W.parity = e->fast_batches & 1u;
    W.did = e->w_did2.p + (size_t)W.parity * e->fast_cap;
    W.did_prev = e->w_did2.p + (size_t)(W.parity ^ 1u) * e->fast_cap;
    W.clear_n = e->fast_prev_n;
#ifdef GUBER_PHASE_TIMING
    W.dbg = e->dbg.p;
#endif
    BatchView B3 = B2;                     // what k_eval2 reads
    W.st_hits = nullptr;
    if (host_resident) {
        const size_t c = e->fast_cap;
        if (e->d_stash64.ensure(5 * c) || e->d_stash32.ensure(c) || e->d_stash8.ensure(2 * c)) return GUBER_E_NOMEM;
        int64_t* q = e->d_stash64.p;
        W.st_hits = q; W.st_limit = q + c; W.st_duration = q + 2 * c; W.st_burst = q + 3 * c; W.st_created = q + 4 * c;
        W.st_behavior = e->d_stash32.p; W.st_algorithm = e->d_stash8.p; W.st_owner = e->d_stash8.p + c;
        B3.hits = W.st_hits; B3.limit = W.st_limit; B3.duration = W.st_duration; B3.burst = W.st_burst; B3.created_at = W.st_created;
        B3.behavior = W.st_behavior; B3.algorithm = W.st_algorithm; B3.is_owner = W.st_owner;
    }
    P.B2 = B2; P.B3 = B3; P.W = W; P.ftiles = (n + FT - 1) / FT;
    return 0;
}
static int plan_part(guber_engine* e, const BatchView& B, Work& W, FastPlan& P) {
    BatchView B2 = B;
    B2.n_cap = e->cap256;
    W.careful = 0u;
    W.snap_seq = 0;
    if (e->rb_ride >= 0) attach_counter_readback(e, W);
    // (a GUBER_FUSE_EP engine: packed words and owner count per batch parity — this batch's k_part may run beside the previous
    // batch's k_eval3, k_evalpart_multi)
    W.did = e->w_did3.p + (e->fuse_ep ? (size_t)(e->part_batches & 1) * e->cap256 : 0);
    W.pmslot = e->fuse_ep ? 1u + (uint32_t)(e->part_batches & 1) : 0u;
    W.st_hits = nullptr;
#ifdef GUBER_PHASE_TIMING
    W.dbg = e->dbg.p;
#endif
    P.B2 = B2; P.B3 = B2; P.W = W; P.ftiles = (B.n + FT - 1) / FT;
    return 0;
}
static void finish_fast(guber_engine* e, uint32_t n) {
    e->fast_batches++;
    e->fast_prev_n = n;
    e->batches++;
}

static int launch_batch_inner(guber_engine* e, const BatchView& B, const ResultView& R, bool host_resident);
// requests [pos, pos + len) of a batch as a batch of their own
static BatchView batch_slice(const BatchView& B, uint32_t pos, uint32_t len) {
    BatchView S = B;
    S.n = len;
    if (B.key_stride) S.key_bytes = B.key_bytes + (size_t)pos * B.key_stride; else S.key_off = B.key_off + pos;
    if (B.key_len) S.key_len = B.key_len + pos;
    S.hits = B.hits + pos; S.limit = B.limit + pos; S.duration = B.duration + pos;
    if (B.burst) S.burst = B.burst + pos;
    if (B.created_at) S.created_at = B.created_at + pos;
    if (B.algorithm) S.algorithm = B.algorithm + pos;
    if (B.behavior) S.behavior = B.behavior + pos;
    if (B.is_owner) S.is_owner = B.is_owner + pos;
    if (B.greg_expire) S.greg_expire = B.greg_expire + pos;
    if (B.greg_duration) S.greg_duration = B.greg_duration + pos;
    return S;
}
// One batch through the engine.  A batch that may overflow the cache first goes through the eviction pre-pass (lru_admit: the
// reference evicts in the middle of a stream of requests, lrucache.go:98-100, and the pre-pass reproduces exactly that); a batch
// larger than the cache is then evaluated in pieces of cache_size requests, each with its own pre-pass.
static int launch_batch(guber_engine* e, const BatchView& B, const ResultView& R, bool host_resident = false) {
    if (B.n == 0) return 0;
    if (!lru_may_bind(e, B.n)) return launch_batch_inner(e, B, R, host_resident);
    if (B.n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    uint8_t* const sf0 = e->W.store_flags; Rec* const sa0 = e->W.store_after;
    int rc = 0;
    for (uint32_t pos = 0; pos < B.n && !rc;) {
        uint32_t len = std::min<uint32_t>(B.n - pos, 1u << 20);
        uint32_t st = 0;
        rc = lru_admit(e, lru_keys_of(batch_slice(B, pos, len)), len, B.now_ms, &st);
        if (!rc && st == LRU_CUT) {
            len = (uint32_t)std::min<uint64_t>(len, std::max<uint64_t>(e->cache_size, 1));
            rc = lru_admit(e, lru_keys_of(batch_slice(B, pos, len)), len, B.now_ms, &st);
        }
        // a resident key whose first request cannot insert (guber_kernels_lru.h "ISOLATED"): the requests before it, then it alone, then the rest
        if (!rc && st == LRU_SPLIT) {
            len = e->lru_split_at ? std::min(len, e->lru_split_at) : 1u;
            rc = lru_admit(e, lru_keys_of(batch_slice(B, pos, len)), len, B.now_ms, &st);
            if (!rc && st != LRU_NONE && st != LRU_APPLIED) rc = fail(GUBER_E_HIP, "the eviction pre-pass split a piece twice");
        }
        if (rc) break;
        if (sf0) { e->W.store_flags = sf0 + pos; e->W.store_after = sa0 + pos; }
        rc = launch_batch_inner(e, batch_slice(B, pos, len), ResultView{R.status + pos, R.limit + pos, R.remaining + pos, R.reset_time + pos, R.err + pos}, host_resident);
        pos += len;
    }
    e->W.store_flags = sf0; e->W.store_after = sa0;
    return rc;
}
static int launch_batch_inner(guber_engine* e, const BatchView& B, const ResultView& R, bool host_resident) {
    const uint32_t n = B.n;
    if (n == 0) return 0;
    Work W;
    {
        const int rc = batch_prelude(e, B, W);
        if (rc) return rc;
    }
    const uint32_t tiles = W.tiles;
    if (takes_part_path(e, n, host_resident, false)) {
        FastPlan P;
        {
            const int rc = plan_part(e, B, W, P);
            if (rc) return rc;
        }
        e->span_begin(KT_PART, n);
        hipLaunchKernelGGL(k_part, dim3(P.ftiles), dim3(FT), 0, e->stream, e->T, P.B2, P.W);
        e->span_end();
        e->span_begin(KT_OWN, n);
        hipLaunchKernelGGL(k_own, dim3(PT_PARTS), dim3(256), 0, e->stream, e->T, P.B2, P.W, P.ftiles);
        e->span_end();
        e->span_begin(KT_EVAL3, n);
        hipLaunchKernelGGL(k_eval3, dim3(P.ftiles), dim3(256), 0, e->stream, EvalArgs{e->T, P.B3, R, P.W});
        e->span_end();
        HIPCHK(hipGetLastError());
#ifdef GUBER_PHASE_TIMING
        if (n == e->fast_cap) {   // fold the stamps of full batches (as for the two-launch pipeline below)
            static unsigned long long hb[3 * 2048];
            (void)hipStreamSynchronize(e->stream);
            (void)hipMemcpy(hb, e->dbg.p + 4096, sizeof(hb), hipMemcpyDeviceToHost);
            static const int nst[3] = {6, 8, 4};
            for (int kern = 0; kern < 3; ++kern) {
                const unsigned long long* b = hb + kern * 2048;
                const uint32_t wgs = kern == 1 ? (uint32_t)PT_PARTS : P.ftiles;
                unsigned long long t0 = ~0ull;
                uint32_t ran = 0;                                        // (k_own workgroups beyond the batch's owner count return at once and stamp 0)
                for (uint32_t t = 0; t < wgs; ++t) if (b[t * 8]) { t0 = b[t * 8] < t0 ? b[t * 8] : t0; ran++; }
                for (int k = 0; k < nst[kern] && ran; ++k) {
                    double sum = 0, mx = 0;
                    for (uint32_t t = 0; t < wgs; ++t) { if (!b[t * 8]) continue; const double v = (double)(b[t * 8 + k] - t0) * 0.01; sum += v; mx = v > mx ? v : mx; }
                    e->dbg_avg[2 + kern][k] += sum / ran; e->dbg_max[2 + kern][k] += mx;
                }
            }
            e->dbg_pn++;
        }
#endif
        e->batches++; e->part_batches++;
        return 0;
    }
    if (takes_fast_path(e, n)) {
        // two launches: resolve + in-tile grouping, then evaluation
        FastPlan P;
        {
            const int rc = plan_fast(e, B, host_resident, W, P);
            if (rc) return rc;
        }
        const uint32_t ftiles = P.ftiles;
        e->span_begin(KT_FRONT, n);
        hipLaunchKernelGGL(k_front, dim3(ftiles), dim3(FT), 0, e->stream, e->T, P.B2, P.W);
        e->span_end();
        e->span_begin(KT_EVAL2, n);
        hipLaunchKernelGGL(k_eval2, dim3((n + 255) / 256), dim3(256), 0, e->stream, EvalArgs{e->T, P.B3, R, P.W});
        e->span_end();
        HIPCHK(hipGetLastError());
#ifdef GUBER_PHASE_TIMING
        if (n == e->fast_cap) {   // fold the stamps of full batches: avg and max over workgroups, relative to the first workgroup's entry
            static unsigned long long hb[4096];
            (void)hipStreamSynchronize(e->stream);
            (void)hipMemcpy(hb, e->dbg.p, sizeof(hb), hipMemcpyDeviceToHost);
            for (int kern = 0; kern < 2; ++kern) {
                const int ns = kern ? 5 : 8;
                const unsigned long long* b = hb + kern * 2048;
                unsigned long long t0 = ~0ull;
                for (uint32_t t = 0; t < ftiles; ++t) t0 = b[t * 8] < t0 ? b[t * 8] : t0;
                for (int k = 0; k < ns; ++k) {
                    double sum = 0, mx = 0;
                    for (uint32_t t = 0; t < ftiles; ++t) { const double v = (double)(b[t * 8 + k] - t0) * 0.01; sum += v; mx = v > mx ? v : mx; }
                    e->dbg_avg[kern][k] += sum / ftiles; e->dbg_max[kern][k] += mx;
                }
            }
            e->dbg_n++;
        }
#endif
        finish_fast(e, n);
        return 0;
    }
    int passes = 1;
    while (passes < MAX_PASSES && (1ull << (RADIX_BITS * passes)) < n) passes++;
    e->span_begin(KT_RESOLVE);
    hipLaunchKernelGGL(k_resolve, dim3(tiles), dim3(TILE), 0, e->stream, e->T, B, W);
    e->span_end();
    const uint32_t* kin = nullptr; const uint32_t* vin = nullptr;
    uint32_t* kout = W.keyA; uint32_t* vout = W.valA;
    for (int p = 0; p < passes; ++p) {
        if (p > 0) {
            e->span_begin(KT_HIST);
            hipLaunchKernelGGL(k_hist, dim3(tiles), dim3(TILE), 0, e->stream, W, n, p, kin);
            e->span_end();
        }
        e->span_begin(p == 0 ? KT_SCATTER0 : KT_SCATTER);
        hipLaunchKernelGGL(k_scatter, dim3(tiles), dim3(TILE), 0, e->stream, e->T, B, W, p, p == 0 ? 1 : 0,
                           p == passes - 1 ? 1 : 0, kin, vin, kout, vout);
        e->span_end();
        kin = kout; vin = vout;
        kout = (kout == W.keyA) ? W.keyB : W.keyA; vout = (vout == W.valA) ? W.valB : W.valA;
    }
    e->span_begin(KT_HEADS);
    hipLaunchKernelGGL(k_heads, dim3((n + 255) / 256), dim3(256), 0, e->stream, W, n);
    e->span_end();
    e->span_begin(KT_EVAL);
    hipLaunchKernelGGL(k_eval, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, B, R, W);
    e->span_end();
    HIPCHK(hipGetLastError());
    e->batches++;
    return 0;
}

static int check_batch_args(const guber_batch_t* b, const guber_result_t* r) {
    if (!b || !r) return fail(GUBER_E_INVALID_ARG, "null batch/result");
    if (b->n == 0) return 0;
    if (!b->key_bytes || !b->key_off || !b->hits || !b->limit || !b->duration)
        return fail(GUBER_E_INVALID_ARG, "batch is missing a mandatory array");
    if (!r->status || !r->limit || !r->remaining || !r->reset_time || !r->err)
        return fail(GUBER_E_INVALID_ARG, "result is missing an array");
    return 0;
}

extern "C" int guber_eval_batch_dev(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    int rc = check_batch_args(b, r);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    BatchView B{b->n, 0, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms};
    ResultView R{r->status, r->limit, r->remaining, r->reset_time, r->err};
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
    return launch_batch(e, B, R);
}

extern "C" int guber_eval_batches_dev(guber_engine_t* e, const guber_batch_t* batches, guber_result_t* results, uint32_t count,
                                      uint32_t* done) {
    if (done) *done = 0;
    if (!e || (count && (!batches || !results))) return fail(GUBER_E_INVALID_ARG, "null argument");
    for (uint32_t k = 0; k < count; ++k) {
        const int rc = check_batch_args(&batches[k], &results[k]);
        if (rc) return rc;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    for (uint32_t k = 0; k < count; ++k) {
        const guber_batch_t* b = &batches[k]; guber_result_t* r = &results[k];
        BatchView B{b->n, 0, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                    b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms};
        ResultView R{r->status, r->limit, r->remaining, r->reset_time, r->err};
        r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
        const int rc = launch_batch(e, B, R);
        if (rc) return rc;
        if (done) *done = k + 1;
    }
    return GUBER_OK;
}

// One dispatcher for several engines (the logical shards of a GPU, each a table of its own): batch k goes to
// engines[which[k]]; per engine the array order is kept, between engines there is nothing to order (disjoint keys, no
// shared state).  Each round takes the next batch of every engine that has one and, where the engines share device and
// stream and the batches take the two-launch pipeline, enqueues up to MULTI_MAX of them as ONE k_front_multi + ONE
// k_eval2_multi (guber_kernels.h): the batches' dependent memory trips then overlap inside a launch, without the
// per-stream kernel boundaries that throttle shards running on separate streams (profiles/archive/r02_m_shard_streams.txt).
static bool fits_fused(const guber_engine* e, uint32_t n) {
#ifdef GUBER_PHASE_TIMING
    return false;
#else
    return e->fuse && takes_fast_path(e, n);
#endif
}
// (a batch that may overflow the cache goes alone, through launch_batch and its eviction pre-pass)
static bool can_fuse(guber_engine* e, uint32_t n) { return fits_fused(e, n) && !lru_may_bind_unlocked(e, n); }   // (takes the engine mutex for the look)

// GUBER_FUSE_EP: a group's k_eval3_multi that has not been launched yet — held back until the same tables' next group comes (then
// it shares that group's first launch: k_evalpart_multi) or until anything else is about to be enqueued on its stream / the call ends
// (then it goes on its own).  Lives inside ONE guber_eval_batches_routed_dev call, one per stream the call uses.
// guber_front: "every evaluation of generation g on this stream has been launched" as an event the answers' way home waits for.
// A held-back evaluation carries the hook of its generation; whoever launches it — the dispatcher's next group (k_evalpart_multi), a
// flush, another thread's entry point — counts it off, and the last one records the event behind the launch.
struct EvalHook {
    hipEvent_t ev = nullptr; hipStream_t st = nullptr;
    std::atomic<int> outstanding{0}; std::atomic<bool> recorded{false};
    void launched() { if (outstanding.fetch_sub(1) == 1) { (void)hipEventRecord(ev, st); recorded.store(true, std::memory_order_release); } }
};
struct PendingEval {
    std::mutex pm;                                                 // two threads that each hold ONE of the group's engines may both come to launch it
    EvalHook* hook = nullptr;                                      // (written under pm)
    std::atomic<bool> valid{false};                                // (written under pm; the dispatcher also looks before it has taken the engines' locks, and again after)
    int n = 0; uint32_t tiles = 0; uint64_t units = 0;
    guber_engine* eng[MULTI_MAX]; MultiEval ME;
};
static thread_local int tl_ep_dispatcher = 0;                      // this thread is inside a routed call that holds evaluations back: it launches them itself
// launch it (if it has not been launched).  The caller holds the mutex of at least one of its engines: nothing can be enqueued on
// any of them by the dispatcher meanwhile (it takes them all), and the launch lands on their stream before whatever the caller enqueues next.
static int launch_held(PendingEval& p, bool spans) {
    std::lock_guard<std::mutex> lk(p.pm);
    if (!p.valid) return 0;
    p.valid = false;
    guber_engine* e0 = p.eng[0];
    if (hipSetDevice(e0->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    if (spans) e0->span_begin(KT_EVAL3_MULTI, p.units);            // (per-kernel timing belongs to the group's first engine: only under its mutex)
    hipLaunchKernelGGL(k_eval3_multi, dim3(p.tiles), dim3(256), 0, e0->stream, p.ME);
    if (spans) e0->span_end();
    if (p.hook) { p.hook->launched(); p.hook = nullptr; }
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    return 0;
}
// an entry point other than the dispatcher that holds it back, with e's mutex held
static void ep_flush_held(const guber_engine* e) {
    if (!e->held || tl_ep_dispatcher) return;
    (void)launch_held(*e->held, false);
    e->held = nullptr;
}
// the dispatcher's own: with all of its engines locked (engines_locked) or locking them here
static int flush_pending(PendingEval& p, bool engines_locked = false) {
    guber_engine* order[MULTI_MAX];
    for (int i = 0; i < p.n; ++i) order[i] = p.eng[i];
    std::sort(order, order + p.n);
    if (!engines_locked) for (int i = 0; i < p.n; ++i) order[i]->mu.lock();
    struct Unlock { guber_engine** o; int g; ~Unlock() { for (int i = g - 1; i >= 0; --i) o[i]->mu.unlock(); } } unlock{order, engines_locked ? 0 : p.n};
    const int rc = launch_held(p, true);
    for (int i = 0; i < p.n; ++i) if (p.eng[i]->held == &p) p.eng[i]->held = nullptr;
    return rc;
}
// every k_eval3 a call is holding back: at most one per set of engines (sets are disjoint: one that overlaps a new group without
// being it is launched before the group is)
struct PendSet {
    std::vector<std::unique_ptr<PendingEval>> items;
    // (a slot only ever serves ONE set of tables: an engine's `held` may outlive a foreign launch and must not come to mean another group)
    PendingEval* slot_for(guber_engine* const* grp, int g) {
        for (auto& q : items) {
            if (q->valid || q->n != g) continue;
            bool same = true;
            for (int i = 0; i < g && same; ++i) same = q->eng[i] == grp[i];
            if (same) return q.get();
        }
        items.emplace_back(new PendingEval());
        items.back()->n = g;
        for (int i = 0; i < g; ++i) items.back()->eng[i] = grp[i];
        return items.back().get();
    }
    int flush_touching(guber_engine* const* grp, int g, const PendingEval* keep = nullptr) {
        for (auto& q : items) {
            if (q.get() == keep) continue;                          // (also the ones a foreign thread launched: their engines' `held` is cleared here)
            bool overlap = false;
            for (int i = 0; i < q->n && !overlap; ++i) for (int j = 0; j < g && !overlap; ++j) overlap = q->eng[i] == grp[j];
            if (overlap) { const int rc = flush_pending(*q); if (rc) return rc; }
        }
        return 0;
    }
    int flush_all() { int r = 0; for (auto& q : items) { const int rc = flush_pending(*q); if (!r) r = rc; } return r; }
};
// GUBER_DISPATCH_PROFILE=1: where the dispatcher's time goes (printed at the end of every guber_eval_batches_routed_dev call):
// [0] waiting for the GPU's progress before a batch may be enqueued (can_fuse -> lru_may_bind), [1] locks + held-back launches,
// [2] preludes + plans, [3] argument blocks, [4] inside hipLaunchKernelGGL, [5] groups, [6] batches
static const bool g_dprof = getenv("GUBER_DISPATCH_PROFILE") != nullptr;
static thread_local uint64_t tl_dp[8];
static inline uint64_t dp_now() { return g_dprof ? (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0; }
struct DpSpan { int k; uint64_t t0; explicit DpSpan(int kk) : k(kk), t0(dp_now()) {} ~DpSpan() { if (g_dprof) tl_dp[k] += dp_now() - t0; } };
// (the batches as views: a caller's guber_batch_t, or an engine's share of a front's generation — guber_front.h; hook: the front's
// "this generation's evaluations on this stream have all been launched" — a held-back evaluation takes it along)
struct GroupItem { BatchView B; ResultView R; };
static int launch_group(guber_engine* const* grp, const GroupItem* it, int g, uint32_t* enqueued, PendSet* ps = nullptr, EvalHook* hook = nullptr) {
    if (g_dprof) { tl_dp[5]++; tl_dp[6] += (uint64_t)g; }
    auto views = [&](int i, BatchView& B, ResultView& R) { B = it[i].B; R = it[i].R; };
    // (one batch: launch_batch.  Measured in round 5 and not kept: a sequence of ONE table's batches through these fused launches —
    // 1.50 against 2.40 G decisions/s: 128 k_own workgroups for the whole chip take 36 us, profiles/r05_g_one_table_fused.txt)
    if (g == 1) {
        guber_engine* e = grp[0];
        if (ps) { const int rc = ps->flush_touching(grp, 1); if (rc) return rc; }
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->held) { (void)launch_held(*e->held, false); e->held = nullptr; }      // (another call's)
        if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
        BatchView B; ResultView R; views(0, B, R);
        const int rc = launch_batch(e, B, R);
        if (rc == 0) ++*enqueued;
        return rc;
    }
    // GUBER_FUSE_EP: the k_eval3 held back on this stream shares this group's first launch (k_evalpart_multi) if the group is the same
    // tables again, in the same order, all taking the owner-partitioned pipeline, and no prelude has anything to enqueue; otherwise it
    // goes first, on its own.  What needs no lock is decided here, before the group's locks are taken (flush_pending takes its own).
    bool same_set = false;
    PendingEval* pend = nullptr;                                   // the k_eval3 held back for exactly these tables, if there is one
    if (ps) {
        same_set = g <= EP_MAX;
        for (int i = 0; i < g && same_set; ++i) same_set = grp[i]->fuse_ep && takes_part_path(grp[i], it[i].B.n, false, true);
        for (auto& q : ps->items) {
            if (!q->valid || !same_set || q->n != g) continue;
            bool same = true;
            for (int i = 0; i < g && same; ++i) same = q->eng[i] == grp[i];
            if (same) { pend = q.get(); break; }
        }
        const int rc0 = ps->flush_touching(grp, g, pend);         // (one that holds some of these engines in another combination: first)
        if (rc0) return rc0;
    }
    // lock the group's engines in address order (any other caller holds at most one engine lock, or locks in this order)
    guber_engine* order[MULTI_MAX];
    uint64_t dp_t = dp_now();
    auto dp_lap = [&](int k) { if (g_dprof) { const uint64_t t = dp_now(); tl_dp[k] += t - dp_t; dp_t = t; } };
    for (int i = 0; i < g; ++i) order[i] = grp[i];
    std::sort(order, order + g);
    for (int i = 0; i < g; ++i) order[i]->mu.lock();
    struct Unlock { guber_engine** o; int g; ~Unlock() { for (int i = g - 1; i >= 0; --i) o[i]->mu.unlock(); } } unlock{order, g};
    for (int i = 0; i < g; ++i)                                    // a k_eval3 ANOTHER call holds back for one of these tables goes first
        if (grp[i]->held && grp[i]->held != pend) { (void)launch_held(*grp[i]->held, false); grp[i]->held = nullptr; }
    if (grp[0]->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    {   // can_fuse() looked at the cache bound BEFORE these locks were taken (it takes and drops each engine's mutex): another thread's
        // AddCacheItem / eval on one of the tables may have used the headroom since.  Looked at again here, under the locks, with the
        // cheap form of the bound; a table that is tight now leaves the group and goes through launch_batch and its eviction pre-pass,
        // one by one — the cache never grows past cache_size and the victims stay lrucache.go's
        bool tight = false;
        for (int i = 0; i < g; ++i) tight = tight || grp[i]->size_upper + it[i].B.n > grp[i]->cache_size;
        if (tight) {
            if (pend && pend->valid) { const int rcf = flush_pending(*pend, true); if (rcf) return rcf; }
            for (int i = g - 1; i >= 0; --i) order[i]->mu.unlock();
            unlock.g = 0;
            int rc1 = 0;
            for (int i = 0; i < g && !rc1; ++i) rc1 = launch_group(&grp[i], &it[i], 1, enqueued, ps, hook);
            return rc1;
        }
    }
    MultiFront MF{}; MultiEval ME{};
    uint32_t tiles = 0, ns[MULTI_MAX];
    int planned = 0, rc = 0;
    bool part = true;                                              // the group takes the owner-partitioned pipeline if all its batches do
    for (int i = 0; i < g; ++i) part = part && takes_part_path(grp[i], it[i].B.n, false, true);
    // GUBER_FUSE_EP: the k_eval3 held back on this stream shares this group's first launch if the group is the same tables again, in
    // the same order, and no prelude has anything to enqueue; otherwise it goes first, on its own
    const bool ep = ps && same_set && part;                       // (same_set, pend: decided before the locks were taken, below the g == 1 case)
    bool join = ep && pend && pend->valid;
    if (pend && pend->valid && !join) { rc = flush_pending(*pend, true); if (rc) return rc; }   // (pend => the same engines: locked)
    dp_lap(1);
    for (int i = 0; i < g; ++i) {
        guber_engine* e = grp[i];
        BatchView B; ResultView R; views(i, B, R);
        Work W; FastPlan P;
        bool defer = false;
        rc = batch_prelude(e, B, W, join ? &defer : nullptr);
        if (!rc && defer) {                                       // this prelude has something to enqueue or to read: the k_eval3 held back goes first
            join = false;
            rc = flush_pending(*pend, true);
            if (!rc) rc = batch_prelude(e, B, W);
        }
        if (!rc) rc = part ? plan_part(e, B, W, P) : plan_fast(e, B, false, W, P);
        if (rc) break;                                            // enqueue what is planned, then report
        tiles += P.ftiles;
        MF.end_tile[planned] = ME.end_tile[planned] = tiles;
        MF.sub[planned] = FrontArgs{e->T, P.B2, P.W};
        ME.sub[planned] = EvalArgs{e->T, P.B3, R, P.W};
        ns[planned++] = B.n;
    }
    dp_lap(2);
    if (planned) {
        static_assert(FT == 256, "k_eval2's workgroup is k_front's tile");
        MF.nb = ME.nb = (uint32_t)planned;
        uint64_t units = 0;
        for (int i = 0; i < planned; ++i) units += ns[i];
        if (part) {
            // (a prelude that was not quiet after all — a counter read-back now rides on this k_part — or a group cut short by an
            // error: the k_eval3 held back goes first)
            bool joined = join && pend->valid && planned == g;
            for (int i = 0; i < planned && joined; ++i) joined = MF.sub[i].T.buckets == pend->ME.sub[i].T.buckets;
            if (pend && pend->valid && !joined) { const int rcf = flush_pending(*pend, true); if (rcf) return rcf; }
            // a counter read-back riding on this k_part runs beside the held-back k_eval3: what it reads lies between the counters
            // before and after that batch, so the host counts that batch's requests among "enqueued since" as well (rb_fold_slot)
            for (int i = 0; i < planned && joined; ++i) {
                if (!MF.sub[i].W.snap_seq) continue;
                for (auto& slot : grp[i]->rb)
                    if (slot.armed && slot.seq == MF.sub[i].W.snap_seq) slot.mark -= std::min<uint64_t>(slot.mark, pend->ME.sub[i].B.n);
            }
            if (joined) {
                // ONE launch: workgroups [0, pending tiles) are the held-back k_eval3, the rest this group's k_part
                MultiEP EP{};
                EP.nb = (uint32_t)planned;
                for (int i = 0; i < planned; ++i) {
                    EP.end_e[i] = pend->ME.end_tile[i]; EP.end_p[i] = MF.end_tile[i];
                    EP.sub[i].E = pend->ME.sub[i]; EP.sub[i].Bp = MF.sub[i].B; EP.sub[i].did_p = MF.sub[i].W.did; EP.sub[i].pmslot_p = MF.sub[i].W.pmslot;
                    const Work& Wp = MF.sub[i].W;
                    EP.sub[i].snap_seq = Wp.snap_seq; EP.sub[i].snap_n = Wp.snap_n; EP.sub[i].snap_c = Wp.snap_c; EP.sub[i].snap_b = Wp.snap_b; EP.sub[i].snap_stamp = Wp.snap_stamp;
                }
                EvalHook* joined_hook;
                { std::lock_guard<std::mutex> pl(pend->pm); pend->valid = false; joined_hook = pend->hook; pend->hook = nullptr; }
                for (int i = 0; i < planned; ++i) grp[i]->held = nullptr;
                dp_lap(3);
                grp[0]->span_begin(KT_EVALPART_MULTI, pend->units);
                hipLaunchKernelGGL(k_evalpart_multi, dim3(pend->tiles + tiles), dim3(256), 0, grp[0]->stream, EP);
                grp[0]->span_end();
                if (joined_hook) joined_hook->launched();
                grp[0]->ep_launches++;
            } else {
                dp_lap(3);
                grp[0]->span_begin(KT_PART_MULTI, units);
                hipLaunchKernelGGL(k_part_multi, dim3(tiles), dim3(FT), 0, grp[0]->stream, MF);
                grp[0]->span_end();
            }
            grp[0]->span_begin(KT_OWN_MULTI, units);
            hipLaunchKernelGGL(k_own_multi, dim3((unsigned)planned * PT_PARTS), dim3(256), 0, grp[0]->stream, MF);
            grp[0]->span_end();
            dp_lap(4);
            if (ep && planned == g) {                             // held back: the same tables' next group, or flush_pending, launches it
                if (!pend) pend = ps->slot_for(grp, planned);
                {
                    std::lock_guard<std::mutex> pl(pend->pm);
                    pend->valid = true; pend->n = planned; pend->tiles = tiles; pend->units = units; pend->ME = ME;
                    pend->hook = hook;
                    if (hook) hook->outstanding.fetch_add(1);
                }
                for (int i = 0; i < planned; ++i) { pend->eng[i] = grp[i]; grp[i]->held = pend; grp[i]->batches++; grp[i]->part_batches++; grp[i]->fused_batches++; }
                *enqueued += (uint32_t)planned;
                if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
                dp_lap(3);
                return rc;
            }
            grp[0]->span_begin(KT_EVAL3_MULTI, units);
            hipLaunchKernelGGL(k_eval3_multi, dim3(tiles), dim3(256), 0, grp[0]->stream, ME);
            grp[0]->span_end();
            for (int i = 0; i < planned; ++i) { grp[i]->batches++; grp[i]->part_batches++; grp[i]->fused_batches++; }
            *enqueued += (uint32_t)planned;
            if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
            return rc;
        }
        grp[0]->span_begin(KT_FRONT_MULTI, units);                    // (per-kernel timing, when enabled, is kept by the group's first engine)
        hipLaunchKernelGGL(k_front_multi, dim3(tiles), dim3(FT), 0, grp[0]->stream, MF);
        grp[0]->span_end();
        grp[0]->span_begin(KT_EVAL2_MULTI, units);
        hipLaunchKernelGGL(k_eval2_multi, dim3(tiles), dim3(256), 0, grp[0]->stream, ME);
        grp[0]->span_end();
        for (int i = 0; i < planned; ++i) { finish_fast(grp[i], ns[i]); grp[i]->fused_batches++; }
        *enqueued += (uint32_t)planned;
        if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    }
    return rc;
}

// One round after the other: the next item of every engine that has one; engines that share device and stream share launches.
// fifo[j] = the items of engines[j] in their order.  *enqueued counts items.  The caller owns `ps` (and flushes it).
static int dispatch_rounds(guber_engine_t* const* engines, uint32_t n_engines, const std::vector<std::vector<GroupItem>>& fifo, PendSet* ps,
                           uint32_t* enqueued, EvalHook* const* hook_of_engine = nullptr) {
    std::vector<size_t> pos(n_engines, 0);
    for (;;) {
        guber_engine* grp[MULTI_MAX]; GroupItem git[MULTI_MAX]; int g = 0;
        EvalHook* hk = nullptr;
        bool any = false;
        int rc = 0;
        for (uint32_t j = 0; j < n_engines && !rc; ++j) {
            if (pos[j] >= fifo[j].size()) continue;
            any = true;
            guber_engine* e = engines[j];
            const GroupItem& item = fifo[j][pos[j]++];
            bool fits;
            { DpSpan sp(0); fits = can_fuse(e, item.B.n); }
            for (int i = 0; i < g && fits; ++i) fits = grp[i] != e;
            if (g && (!fits || g == MULTI_MAX || e->stream != grp[0]->stream || e->device != grp[0]->device)) {
                rc = launch_group(grp, git, g, enqueued, ps, hk);
                g = 0;
                if (rc) break;
            }
            grp[g] = e; git[g] = item; ++g;
            hk = hook_of_engine ? hook_of_engine[j] : nullptr;      // (engines of one stream share their generation's hook)
            if (!fits) { rc = launch_group(grp, git, g, enqueued, ps, hk); g = 0; }
        }
        if (!rc && g) rc = launch_group(grp, git, g, enqueued, ps, hk);
        if (rc) return rc;
        if (!any) break;
    }
    return 0;
}

extern "C" int guber_eval_batches_routed_dev(guber_engine_t* const* engines, uint32_t n_engines, const uint32_t* which,
                                             const guber_batch_t* batches, guber_result_t* results, uint32_t count, uint32_t* done) {
    if (done) *done = 0;
    if (!engines || !n_engines || (count && (!which || !batches || !results))) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::vector<std::vector<GroupItem>> fifo(n_engines);
    uint32_t empty = 0;
    for (uint32_t k = 0; k < count; ++k) {
        if (which[k] >= n_engines || !engines[which[k]]) return fail(GUBER_E_INVALID_ARG, "which[k] names no engine");
        const int rc = check_batch_args(&batches[k], &results[k]);
        if (rc) return rc;
        const guber_batch_t* b = &batches[k]; guber_result_t* r = &results[k];
        r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
        if (!b->n) { ++empty; continue; }
        fifo[which[k]].push_back(GroupItem{BatchView{b->n, 0, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                                                     b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms},
                                           ResultView{r->status, r->limit, r->remaining, r->reset_time, r->err}});
    }
    uint32_t enqueued = 0;
    // GUBER_FUSE_EP engines: the k_eval3 of a group of tables is held back for the same tables' next group (launch_group)
    PendSet pendset;
    bool any_ep = false;
    for (uint32_t j = 0; j < n_engines; ++j) any_ep = any_ep || (engines[j] && engines[j]->fuse_ep);
    PendSet* const ps = any_ep ? &pendset : nullptr;
    struct Dispatching { bool on; Dispatching(bool o) : on(o) { if (on) ++tl_ep_dispatcher; } ~Dispatching() { if (on) --tl_ep_dispatcher; } } dispatching(any_ep);
    const int rc = dispatch_rounds(engines, n_engines, fifo, ps, &enqueued);
    const int rcf = pendset.flush_all();                            // (what was enqueued is completed: its k_eval3 goes now)
    if (done) *done = enqueued;
    if (rc) return rc;
    if (rcf) return rcf;
    if (g_dprof && tl_dp[6]) {
        fprintf(stderr, "[dispatch] %llu batches in %llu groups; per batch: wait-for-progress %.2f us, locks %.2f, preludes+plans %.2f, argument blocks %.2f, launches %.2f\n",
                (unsigned long long)tl_dp[6], (unsigned long long)tl_dp[5], tl_dp[0] / 1e3 / tl_dp[6], tl_dp[1] / 1e3 / tl_dp[6], tl_dp[2] / 1e3 / tl_dp[6],
                tl_dp[3] / 1e3 / tl_dp[6], tl_dp[4] / 1e3 / tl_dp[6]);
        for (auto& v : tl_dp) v = 0;
    }
    if (done) *done = enqueued + empty;
    return GUBER_OK;
}

#include "guber_front.h"

// Host-pointer evaluation: stage -> H2D -> kernels -> D2H.  `idx` (optional) selects a subset of
// the caller's batch (used to re-submit GUBER_ITEM_E_RETRY items).
static void item_from_rec(const Rec& s, guber_item_t* out);
// the same prelude launch_batch has, for the one-launch path
static int small_prelude(guber_engine* e, const BatchView& B) {
    if (B.now_ms > e->clock_ms) e->clock_ms = B.now_ms;
    const int rc = maintain(e, B.n, B.now_ms);
    if (rc) return rc;
    note_enqueued(e, B.n);
    take_stamps(e, B.n);
    e->batches++; e->small_batches++;
    return 0;
}
static int launch_small(guber_engine* e, const BatchView& B, const ResultView& R, SmallOut* out, uint32_t seq) {
    const int rc = small_prelude(e, B);
    if (rc) return rc;
    hipLaunchKernelGGL(k_small, dim3(1), dim3(FT), 0, e->stream, e->T, B, R, out, seq, e->touch);
    HIPCHK(hipGetLastError());
    return 0;
}

static int eval_host_once(guber_engine* e, const guber_batch_t* b, guber_result_t* r, const uint32_t* idx, uint32_t n,
                          guber_store_events_t* sev = nullptr) {
    const bool has_burst = b->burst, has_created = b->created_at, has_greg = b->greg_expire && b->greg_duration;
    // key bytes of the (sub)batch
    size_t kbytes = 0;
    for (uint32_t j = 0; j < n; ++j) { uint32_t i = idx ? idx[j] : j; kbytes += b->key_off[i + 1] - b->key_off[i]; }
    if (kbytes > 0xfffffff0ull) return fail(GUBER_E_BATCH_TOO_LARGE, "key bytes exceed 4 GiB");
    const size_t n64 = (size_t)n * 7;   // hits limit duration burst created greg_expire greg_duration
    const size_t stage_bytes = (kbytes + 16) + (size_t)(n + 1) * 4 + n64 * 8 + (size_t)n * 4 + (size_t)n * 2 + 64 +
                               (size_t)n * (3 * 8 + 2) + 64 + sizeof(SmallOut);
    const bool zc = e->zero_copy;
    int rc = 0;
    if (zc) rc |= e->z_stage.ensure(stage_bytes + 256);
    else {
        rc |= e->h_stage.ensure(stage_bytes + 256);
        rc |= e->d_keys.ensure(kbytes + 16); rc |= e->d_off.ensure(n + 1); rc |= e->d_i64.ensure(n64);
        rc |= e->d_beh.ensure(n); rc |= e->d_u8.ensure((size_t)n * 2);
        rc |= e->d_out64.ensure((size_t)n * 3); rc |= e->d_out8.ensure((size_t)n * 2);
    }
    if (rc) return GUBER_E_NOMEM;
    // carve the arena (8-byte aligned pieces first)
    uint8_t* base = zc ? e->z_stage.p : e->h_stage.p;
    SmallOut* sout = (SmallOut*)base; base += 64;
    int64_t* s64 = (int64_t*)base; base += n64 * 8;
    int64_t* o64 = (int64_t*)base; base += (size_t)n * 3 * 8;
    uint32_t* soff = (uint32_t*)base; base += (size_t)(n + 1) * 4;
    uint32_t* sbeh = (uint32_t*)base; base += (size_t)n * 4;
    uint8_t* su8 = base; base += (size_t)n * 2;
    uint8_t* o8 = base; base += (size_t)n * 2;
    uint8_t* skeys = base;
    uint32_t off = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t i = idx ? idx[j] : j;
        const uint32_t len = b->key_off[i + 1] - b->key_off[i];
        memcpy(skeys + off, b->key_bytes + b->key_off[i], len);
        soff[j] = off; off += len;
        s64[j] = b->hits[i]; s64[n + j] = b->limit[i]; s64[2 * (size_t)n + j] = b->duration[i];
        s64[3 * (size_t)n + j] = has_burst ? b->burst[i] : 0;
        s64[4 * (size_t)n + j] = has_created ? b->created_at[i] : b->now_ms;
        s64[5 * (size_t)n + j] = has_greg ? b->greg_expire[i] : 0;
        s64[6 * (size_t)n + j] = has_greg ? b->greg_duration[i] : 0;
        sbeh[j] = b->behavior ? b->behavior[i] : 0;
        su8[j] = b->algorithm ? b->algorithm[i] : 0;
        su8[n + j] = b->is_owner ? b->is_owner[i] : 1;
    }
    soff[n] = off;
    memset(skeys + off, 0, 16);
    hipStream_t st = e->stream;
    std::vector<uint8_t> h_sflags; std::vector<Rec> h_safter;
    if (sev) {
        if (e->d_sflags.ensure(n) || e->d_safter.ensure(n)) return GUBER_E_NOMEM;
        HIPCHK(hipMemsetAsync(e->d_sflags.p, 0, n, st));
        e->W.store_flags = e->d_sflags.p; e->W.store_after = e->d_safter.p;
    }
    if (zc) {
        // the kernels read the request arrays and write the responses in place, over PCIe: no copy launches
        BatchView B{n, 0, skeys, soff, s64, s64 + n, s64 + 2 * (size_t)n, s64 + 3 * (size_t)n, s64 + 4 * (size_t)n,
                    su8, sbeh, su8 + n, has_greg ? s64 + 5 * (size_t)n : nullptr, has_greg ? s64 + 6 * (size_t)n : nullptr, b->now_ms};
        ResultView R{o8, o64, o64 + n, o64 + 2 * (size_t)n, o8 + n};
        bool done = false;
        if (n <= FT && !sev && !e->no_small && !e->careful && !lru_may_bind(e, n)) {
            // one launch, one workgroup; completion = a sequence number in host memory, polled
            const uint32_t seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
            sout->done = 0;
            rc = launch_small(e, B, R, sout, seq);
            if (rc) return rc;
            volatile unsigned int* flag = &sout->done;
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(st)); break; }
            }
            if (!sout->fallback) {
                e->last_ctr.over += sout->over; e->last_ctr.hits += sout->hits; e->last_ctr.misses += sout->misses; e->last_ctr.size += sout->size_delta;
                done = true;
            } else e->small_fallbacks++;
        }
        if (!done) {
            rc = launch_batch(e, B, R, true);
            e->W.store_flags = nullptr; e->W.store_after = nullptr;
            if (rc) return rc;
            if (sev) {
                try { h_sflags.resize(n); h_safter.resize(n); } catch (...) { return GUBER_E_NOMEM; }
                HIPCHK(hipMemcpyAsync(h_sflags.data(), e->d_sflags.p, n, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(h_safter.data(), e->d_safter.p, (size_t)n * sizeof(Rec), hipMemcpyDeviceToHost, st));
            }
            rc = enqueue_counter_readback(e);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(st));
            fold_counters(e);
        }
    } else {
        HIPCHK(hipMemcpyAsync(e->d_keys.p, skeys, kbytes + 16, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_off.p, soff, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_i64.p, s64, n64 * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_beh.p, sbeh, (size_t)n * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_u8.p, su8, (size_t)n * 2, hipMemcpyHostToDevice, st));
        int64_t* d64 = e->d_i64.p;
        BatchView B{n, 0, e->d_keys.p, e->d_off.p, d64, d64 + n, d64 + 2 * (size_t)n, d64 + 3 * (size_t)n, d64 + 4 * (size_t)n,
                    e->d_u8.p, e->d_beh.p, e->d_u8.p + n, has_greg ? d64 + 5 * (size_t)n : nullptr, has_greg ? d64 + 6 * (size_t)n : nullptr, b->now_ms};
        ResultView R{e->d_out8.p, e->d_out64.p, e->d_out64.p + n, e->d_out64.p + 2 * (size_t)n, e->d_out8.p + n};
        rc = launch_batch(e, B, R);
        e->W.store_flags = nullptr; e->W.store_after = nullptr;
        if (rc) return rc;
        if (sev) {
            try { h_sflags.resize(n); h_safter.resize(n); } catch (...) { return GUBER_E_NOMEM; }
            HIPCHK(hipMemcpyAsync(h_sflags.data(), e->d_sflags.p, n, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(h_safter.data(), e->d_safter.p, (size_t)n * sizeof(Rec), hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipMemcpyAsync(o64, e->d_out64.p, (size_t)n * 3 * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(o8, e->d_out8.p, (size_t)n * 2, hipMemcpyDeviceToHost, st));
        rc = enqueue_counter_readback(e);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(st));
        fold_counters(e);
    }
    e->W.store_flags = nullptr; e->W.store_after = nullptr;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t i = idx ? idx[j] : j;
        r->status[i] = o8[j]; r->err[i] = o8[n + j];
        r->limit[i] = o64[j]; r->remaining[i] = o64[n + j]; r->reset_time[i] = o64[2 * (size_t)n + j];
        if (sev && o8[n + j] != GUBER_ITEM_E_RETRY) {
            sev->flags[i] = h_sflags[j];
            if (h_sflags[j] & GUBER_STORE_ONCHANGE) {
                item_from_rec(h_safter[j], &sev->items[i]);
                sev->items[i].key = b->key_bytes + b->key_off[i];
                sev->items[i].key_len = b->key_off[i + 1] - b->key_off[i];
            }
        }
    }
    return 0;
}

// (engine mutex held by the caller: the GLOBAL exchange re-runs collided rows through here without letting go of its engines)
static int eval_batch_host_locked(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r, guber_store_events_t* sev) {
    int rc = 0;
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const DevCounters before = e->last_ctr;
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0;
    if (b->n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    if (b->n) {
        rc = eval_host_once(e, b, r, nullptr, b->n, sev);
        if (rc) return rc;
        // two new keys sharing one 64-bit hash inside one batch: re-submit the affected items; on the
        // second pass the first key is resident and the other one probes past it.
        for (int round = 0; round < 64; ++round) {
            std::vector<uint32_t> again;
            for (uint32_t i = 0; i < b->n; ++i) if (r->err[i] == GUBER_ITEM_E_RETRY) again.push_back(i);
            if (again.empty()) break;
            e->careful = true;
            rc = eval_host_once(e, b, r, again.data(), (uint32_t)again.size(), sev);
            e->careful = false;
            if (rc) return rc;
        }
        rc = maintain(e, 0, b->now_ms);
        if (rc) return rc;
    }
    r->over_limit_count = e->last_ctr.over - before.over;
    r->cache_hits = e->last_ctr.hits - before.hits;
    r->cache_misses = e->last_ctr.misses - before.misses;
    r->unexpired_evictions = e->last_ctr.evictions - before.evictions;
    r->cache_size = e->last_ctr.size;
    return GUBER_OK;
}
static int eval_batch_host(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r, guber_store_events_t* sev) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    const int rc = check_batch_args(b, r);
    if (rc) return rc;
    if (sev && b->n && (!sev->flags || !sev->items)) return fail(GUBER_E_INVALID_ARG, "null store event arrays");
    if (sev && b->n) memset(sev->flags, 0, b->n);
    std::lock_guard<std::mutex> lk(e->mu);
    return eval_batch_host_locked(e, b, r, sev);
}

// ---- stages: batch buffers in device-visible host memory that the CALLER fills in place and the kernels read / write in
// place.  A batcher that owns two of them fills one while the GPU evaluates the other: no staging copy, no copy launch,
// no allocation per batch (what SURVEY.md section 8d calls the overlapped end-to-end path).
struct guber_stage {
    guber_engine* e = nullptr;
    uint32_t max_n = 0, key_cap = 0;
    CohBuf<uint8_t> mem;
    guber_batch_t batch{}; guber_result_t result{};
    SmallOut* sout = nullptr;
    DevCounters* rb_ctr = nullptr; BlockCounters* rb_bctr = nullptr;     // this stage's own counter read-back after its batch ...
    DevCounters* rb0_ctr = nullptr; BlockCounters* rb0_bctr = nullptr;   // ... and before it: the difference is exactly this batch
    hipEvent_t ev = nullptr;
    guber_engine::GroupEv* gev = nullptr; uint32_t gev_seq = 0;   // submitted as one of a group: the group's completion event (the slot's use)
    MultiArgsMem* h_margs = nullptr;                               // argument blocks of a group this stage leads (device-visible host memory)
    uint32_t* h_dest = nullptr;                                    // guber_stage_submit_routed: per request, engine index << 24 | rank in that engine's share
    std::vector<guber_engine*> routed;                             // ... and the engines of the submission in flight (retries go back to them)
    // a routed stage of <= 256 requests: one workgroup per engine in ONE launch (k_small_routed); every share has its own outcome
    struct RoutedPart { guber_engine* e; uint32_t engine, n, seq; SmallOut* out; bool pending; int rc; };
    std::vector<RoutedPart> parts; uint8_t* h_parts_out = nullptr;  // (mode 4; a part is only touched under its engine's mutex)
    // guber_stage_route: the shares' sizes + completion flag (host, device-visible), per-request engine and per-tile tables (HBM)
    uint32_t* h_route = nullptr; DevBuf<uint8_t> d_route; uint32_t route_seq = 0; bool route_pending = false; uint32_t route_engines = 0;
    bool keys_resident = false;      // guber_stage_route left this batch's key bytes in the HBM mirror (dmem): guber_stage_submit_routed does not copy them again
    uint32_t seq = 0, n = 0; int64_t now_ms = 0;
    int mode = 0;                    // 0 idle, 1 small path complete, 2 pipeline in flight, 3 small path launched, outcome not looked at yet (guber_stages_submit),
                                     // 4 routed small path launched (guber_stage_submit_routed): outcomes per part
    bool no_agg = false;             // submitted without per-batch aggregates (guber_stages_submit)
    // Large batches: two DMA copies on a copy stream (the fixed-width columns present, the keys) bring the requests into the
    // stage's device mirror while the previous batches' kernels run; the pipeline then works on HBM and k_eval2 writes the
    // responses straight into the host arrays (posted writes).  The link carries the requests at the copy engine's rate
    // (46-48 GB/s) instead of at the rate of k_front's dependent reads (24 GB/s in total with everything in place).
    // Measured and dropped (profiles/archive/r02_v_end_to_end_variants.txt): responses to HBM and a DMA copy back (a hipMemcpyAsync
    // costs 40-60 us of host time), a copy kernel instead of the DMA (kernels of two streams overlap badly).
    DevBuf<uint8_t> dmem;            // device mirror of the in block
    uint8_t *h_in = nullptr, *h_out = nullptr; size_t in_fixed = 0, out_bytes = 0;   // host blocks; in_fixed = bytes before the keys
    hipEvent_t ev_in = nullptr;
};

static int resolve_small(guber_stage* s, bool block);
static int resolve_routed_small(guber_stage* s, bool block);
static int resolve_small_locked(guber_stage* s, bool block, guber_engine* holder = nullptr);
extern "C" int guber_stage_create(guber_engine_t* e, uint32_t max_n, uint32_t key_bytes_cap, guber_stage_t** out) {
    if (!e || !out || max_n == 0) return fail(GUBER_E_INVALID_ARG, "null argument");
    *out = nullptr;
    if (max_n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "stage larger than guber_config_t.max_batch");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    guber_stage* s = new guber_stage();
    s->e = e; s->max_n = max_n; s->key_cap = key_bytes_cap ? key_bytes_cap : max_n * 64u;
    const size_t n = max_n;
    // [counters | in block: key_off, hits, limit, duration, behavior, algorithm, is_owner, burst, created_at, keys |
    //  out block: limit, remaining, reset_time, status, err]; every column starts on a 64-byte boundary and is padded to one
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    const size_t in_fixed = col((n + 1) * 4) + 5 * col(n * 8) + col(n * 4) + 2 * col(n);
    const size_t in_bytes = col(in_fixed + (size_t)s->key_cap + 64);
    const size_t out_bytes = 3 * col(n * 8) + 2 * col(n);
    const size_t head = 256 + 2 * (col(sizeof(DevCounters)) + col((size_t)e->n_bctr * sizeof(BlockCounters))) + col(sizeof(MultiArgsMem)) + col(n * 4) + 16 * 64 + 128;
    const size_t bytes = head + in_bytes + out_bytes + 256;
    if (s->mem.ensure(bytes) || hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming) != hipSuccess) {
        s->mem.release(); delete s; return GUBER_E_NOMEM;
    }
    memset(s->mem.p, 0, bytes);
    uint8_t* p = s->mem.p;
    s->sout = (SmallOut*)p; p += 64;
    s->rb_ctr = (DevCounters*)p; p += col(sizeof(DevCounters));
    s->rb_bctr = (BlockCounters*)p; p += col((size_t)e->n_bctr * sizeof(BlockCounters));
    s->rb0_ctr = (DevCounters*)p; p += col(sizeof(DevCounters));
    s->rb0_bctr = (BlockCounters*)p; p += col((size_t)e->n_bctr * sizeof(BlockCounters));
    s->h_margs = (MultiArgsMem*)p; p += col(sizeof(MultiArgsMem));
    s->h_dest = (uint32_t*)p; p += col(n * 4);
    s->h_parts_out = p; p += 16 * 64;
    s->h_route = (uint32_t*)p;                                     // [0..15] counts, [16] done flag
    p = s->mem.p + head;
    s->h_in = p; s->h_out = p + in_bytes; s->in_fixed = in_fixed; s->out_bytes = out_bytes;
    guber_batch_t& b = s->batch; guber_result_t& r = s->result;
    b.key_off = (uint32_t*)p; p += col((n + 1) * 4);
    b.hits = (int64_t*)p; p += col(n * 8);
    b.limit = (int64_t*)p; p += col(n * 8);
    b.duration = (int64_t*)p; p += col(n * 8);
    b.behavior = (uint32_t*)p; p += col(n * 4);
    b.algorithm = p; p += col(n);
    b.is_owner = p; p += col(n);
    b.burst = (int64_t*)p; p += col(n * 8);
    b.created_at = (int64_t*)p; p += col(n * 8);
    b.key_bytes = p;
    p = s->h_out;
    r.limit = (int64_t*)p; p += col(n * 8);
    r.remaining = (int64_t*)p; p += col(n * 8);
    r.reset_time = (int64_t*)p; p += col(n * 8);
    r.status = p; p += col(n);
    r.err = p;
    *out = s;
    return GUBER_OK;
}
// no engine may keep pointing at a stage that is being abandoned or freed (its next submit would look at it: resolve_small_locked)
static void forget_small_pending(guber_stage* s) {
    std::vector<guber_engine*> engs;
    if (s->e) engs.push_back(s->e);
    for (auto& part : s->parts) if (part.e && std::find(engs.begin(), engs.end(), part.e) == engs.end()) engs.push_back(part.e);
    for (guber_engine* e : engs) {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->small_pending == s) e->small_pending = nullptr;
    }
}
extern "C" void guber_stage_destroy(guber_stage_t* s) {
    if (!s) return;
    if (s->mode) (void)guber_stage_wait(s);
    if (s->route_pending && s->e && !s->e->set_device()) (void)hipStreamSynchronize(s->e->stream);   // (the routing launches write into the stage)
    forget_small_pending(s);
    if (s->ev) (void)hipEventDestroy(s->ev);
    if (s->ev_in) (void)hipEventDestroy(s->ev_in);
    s->dmem.release();
    s->d_route.release();
    s->mem.release();
    delete s;
}
extern "C" guber_batch_t* guber_stage_batch(guber_stage_t* s) { return s ? &s->batch : nullptr; }
extern "C" guber_result_t* guber_stage_result(guber_stage_t* s) { return s ? &s->result : nullptr; }
extern "C" uint32_t guber_stage_capacity(guber_stage_t* s, uint32_t* key_bytes_cap) { if (s && key_bytes_cap) *key_bytes_cap = s->key_cap; return s ? s->max_n : 0; }

extern "C" int guber_stage_submit(guber_stage_t* s) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    if (s->mode) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
    guber_engine* e = s->e;
    const guber_batch_t& b = s->batch;
    s->n = b.n; s->now_ms = b.now_ms; s->no_agg = false;
    if (b.n == 0) { s->mode = 0; return GUBER_OK; }
    if (b.n > s->max_n || b.key_off[b.n] > s->key_cap) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled");
    memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);            // the kernels read keys as 8-byte words
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    if (e->small_pending) { const int rcp = resolve_small_locked(e->small_pending, true, e); if (rcp < 0) return rcp; }
    BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                b.greg_expire, b.greg_duration, b.now_ms};
    ResultView R{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
    if (b.n <= FT && !e->no_small && !lru_may_bind(e, b.n)) {
        s->seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
        s->sout->done = 0;
        int rc = launch_small(e, B, R, s->sout, s->seq);
        if (rc) return rc;
        // One workgroup, a few microseconds: wait for it here.  If the small path declined the batch (requests of one key
        // that differ, a hash collision) the general pipeline has to run it BEFORE anything submitted later, so the
        // decision cannot be left to guber_stage_wait.
        volatile unsigned int* flag = &s->sout->done;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != s->seq) {
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(e->stream)); break; }
        }
        if (!s->sout->fallback) {
            e->last_ctr.over += s->sout->over; e->last_ctr.hits += s->sout->hits; e->last_ctr.misses += s->sout->misses; e->last_ctr.size += s->sout->size_delta;
            s->mode = 1;
            return GUBER_OK;
        }
        e->small_fallbacks++;
    }
    // (maintenance first: it may synchronise and rebuild; the read-back pair must bracket the kernels only)
    int rc = maintain(e, b.n, b.now_ms);
    if (rc) return rc;
    // a batch that fills at least half of the stage reaches HBM by DMA; smaller ones are read in place
    const bool dma = e->stage_dma && s->max_n >= 4096 && (size_t)b.n * 2 >= s->max_n && !b.greg_expire && !b.greg_duration;
    if (dma) {
        if (!e->copy_in && hipStreamCreateWithFlags(&e->copy_in, hipStreamNonBlocking) != hipSuccess) return fail(GUBER_E_HIP, "hipStreamCreate");
        const size_t in_bytes = (size_t)(s->h_out - s->h_in);
        if (s->dmem.ensure(in_bytes)) return GUBER_E_NOMEM;
        uint8_t* d_in = s->dmem.p;
        // two copies: the fixed-width columns up to the last one present, then the keys
        const void* last = b.created_at ? (const void*)(b.created_at + b.n) : b.burst ? (const void*)(b.burst + b.n) : b.is_owner ? (const void*)(b.is_owner + b.n) : (const void*)(b.algorithm + b.n);
        const size_t fixed = (size_t)((const uint8_t*)last - s->h_in);
        HIPCHK(hipMemcpyAsync(d_in, s->h_in, fixed, hipMemcpyHostToDevice, e->copy_in));
        HIPCHK(hipMemcpyAsync(d_in + s->in_fixed, s->h_in + s->in_fixed, (size_t)b.key_off[b.n] + 16, hipMemcpyHostToDevice, e->copy_in));
        HIPCHK(hipEventRecord(s->ev_in, e->copy_in));
        HIPCHK(hipStreamWaitEvent(e->stream, s->ev_in, 0));
        auto dev = [&](const void* hp) { return hp ? d_in + ((const uint8_t*)hp - s->h_in) : nullptr; };
        B = BatchView{b.n, 0, dev(b.key_bytes), (const uint32_t*)dev(b.key_off), (const int64_t*)dev(b.hits), (const int64_t*)dev(b.limit),
                      (const int64_t*)dev(b.duration), (const int64_t*)dev(b.burst), (const int64_t*)dev(b.created_at), dev(b.algorithm),
                      (const uint32_t*)dev(b.behavior), dev(b.is_owner), nullptr, nullptr, b.now_ms};
    }
    hipLaunchKernelGGL(k_ctr_snapshot, dim3(1), dim3(256), 0, e->stream, e->ctr.p, e->bctr.p, e->n_bctr, s->rb0_ctr, s->rb0_bctr, (uint32_t*)nullptr, 0u);
    rc = launch_batch(e, B, R, !dma);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ctr_snapshot, dim3(1), dim3(256), 0, e->stream, e->ctr.p, e->bctr.p, e->n_bctr, s->rb_ctr, s->rb_bctr, (uint32_t*)nullptr, 0u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev, e->stream));
    s->gev = nullptr; s->mode = 2;
    return GUBER_OK;
}

extern "C" int guber_stage_wait(guber_stage_t* s) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    guber_engine* e = s->e;
    guber_result_t& r = s->result;
    r.over_limit_count = r.cache_hits = r.cache_misses = r.unexpired_evictions = 0;
    if (s->mode == 0) return GUBER_OK;
    if (s->mode == 3) {                                      // launched by guber_stages_submit: look at the outcome now
        const int rc3 = resolve_small(s, true);
        if (rc3 < 0) return rc3;
    }
    if (s->mode == 4) {                                      // a routed stage on the one-launch path: every share's outcome
        const int rc4 = resolve_routed_small(s, true);
        if (rc4 < 0) { forget_small_pending(s); s->mode = 0; s->routed.clear(); s->parts.clear(); return rc4; }
    }
    bool general = s->mode == 2;
    if (s->mode == 1) {                                      // answered by the one-launch path, already complete (guber_stage_submit)
        s->mode = 0;
        if (s->parts.empty()) {
            std::lock_guard<std::mutex> lk(e->mu);
            r.over_limit_count = s->sout->over; r.cache_hits = s->sout->hits; r.cache_misses = s->sout->misses; r.cache_size = e->last_ctr.size;
            return GUBER_OK;
        }
        s->parts.clear();                                    // (a routed stage: no aggregates; a re-run share may have left internal retries)
    }
    if (general) {
        if (s->gev) {
            if (s->gev->seq.load(std::memory_order_acquire) == s->gev_seq && hipEventSynchronize(s->gev->ev) != hipSuccess) return fail(GUBER_E_HIP, "hipEventSynchronize");
            s->gev = nullptr;
        } else if (hipEventSynchronize(s->ev) != hipSuccess) return fail(GUBER_E_HIP, "hipEventSynchronize");
        s->mode = 0;
        std::lock_guard<std::mutex> lk(e->mu);
        if (s->no_agg) general = false;                       // no read-backs were taken: the aggregates stay 0 (guber_stats has the totals)
        // the read-backs taken right before and right after this batch's kernels: their difference is this batch alone
        auto fold = [&](const DevCounters* c0, const BlockCounters* b0) {
            DevCounters c = *c0;
            for (uint32_t k = 0; k < e->n_bctr; ++k) { c.over += b0[k].over; c.hits += b0[k].hits; c.misses += b0[k].misses; c.size += b0[k].size_delta; }
            return c;
        };
        if (general) {
            const DevCounters c1 = fold(s->rb_ctr, s->rb_bctr), c0 = fold(s->rb0_ctr, s->rb0_bctr);
            r.over_limit_count = c1.over - c0.over; r.cache_hits = c1.hits - c0.hits; r.cache_misses = c1.misses - c0.misses;
            r.cache_size = c1.size;
        }
    }
    // two new keys sharing one 64-bit hash (or one claim fingerprint) inside the batch: re-submit those items on the host
    // path, which runs the careful rounds
    if (memchr(r.err, GUBER_ITEM_E_RETRY, s->n)) {
        std::vector<guber_engine*> engs = s->routed;
        if (engs.empty()) engs.push_back(e);
        for (size_t j = 0; j < engs.size(); ++j) {
            guber_engine* ej = engs[j];
            std::vector<uint32_t> again;
            for (uint32_t i = 0; i < s->n; ++i)
                if (r.err[i] == GUBER_ITEM_E_RETRY && (s->routed.empty() || (s->h_dest[i] >> 24) == j)) again.push_back(i);
            if (again.empty()) continue;
            std::lock_guard<std::mutex> lk(ej->mu);
            if (ej->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
            guber_batch_t hb = s->batch;
            int rc0 = engine_refresh_counters(ej);
            if (rc0) return rc0;
            const DevCounters t0 = ej->last_ctr;
            for (int round = 0; round < 64 && !again.empty(); ++round) {
                ej->careful = true;
                const int rc = eval_host_once(ej, &hb, &r, again.data(), (uint32_t)again.size(), nullptr);
                ej->careful = false;
                if (rc) return rc;
                std::vector<uint32_t> next;
                for (uint32_t i : again) if (r.err[i] == GUBER_ITEM_E_RETRY) next.push_back(i);
                again.swap(next);
            }
            r.over_limit_count += ej->last_ctr.over - t0.over; r.cache_hits += ej->last_ctr.hits - t0.hits; r.cache_misses += ej->last_ctr.misses - t0.misses;
            r.cache_size = ej->last_ctr.size;
        }
    }
    s->routed.clear();
    return GUBER_OK;
}

// ---- several stages in one submission: what the dispatcher of a GPUWorkerPool calls (worker_pool.cpp).  Never waits for the GPU.
// A <= 256-request stage launched here is in mode 3 until somebody looks at its outcome (guber_stage_poll / guber_stage_wait,
// or the next submission on its engine): the one-launch path may decline a batch (requests of one key that differ, a hash
// collision), and then the general pipeline has to run it before anything later of the same engine.
// the shares of a routed small stage that belong to `holder` (its mutex held): outcome looked at, a declined share re-run through
// the general pipeline — synchronously, on the host-pointer path, picking the share out of the stage by its ranks
static int resolve_routed_parts_locked(guber_stage* s, guber_engine* holder, bool block) {
    for (auto& part : s->parts) {
        if (part.e != holder || !part.pending) continue;
        volatile unsigned int* flag = &part.out->done;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != part.seq) {
            if (!block) return 0;
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(holder->stream)); break; }
        }
        part.pending = false;
        if (holder->small_pending == s) holder->small_pending = nullptr;
        if (!part.out->fallback) {
            holder->last_ctr.over += part.out->over; holder->last_ctr.hits += part.out->hits; holder->last_ctr.misses += part.out->misses; holder->last_ctr.size += part.out->size_delta;
            continue;
        }
        holder->small_fallbacks++;
        if (holder->set_device()) return part.rc = fail(GUBER_E_HIP, "hipSetDevice");
        // the kernel also declines a share whose ranks in dest are not a permutation of 0..n-1 (the caller wrote dest): that is a
        // caller error, not a batch for the general pipeline
        std::vector<uint32_t> idx(part.n, 0xffffffffu);
        uint32_t placed = 0;
        for (uint32_t i = 0; i < s->n; ++i) {
            if ((s->h_dest[i] >> 24) != part.engine) continue;
            const uint32_t rk = s->h_dest[i] & 0xffffffu;
            if (rk >= part.n || idx[rk] != 0xffffffffu) { placed = 0xffffffffu; break; }
            idx[rk] = i; ++placed;
        }
        if (placed != part.n) return part.rc = fail(GUBER_E_INVALID_ARG, "dest: the ranks of an engine's share are not a permutation of 0 .. count-1");
        guber_batch_t hb = s->batch;
        part.rc = eval_host_once(holder, &hb, &s->result, idx.data(), part.n, nullptr);
        if (part.rc) return part.rc;
    }
    return 1;
}
static int resolve_small_locked(guber_stage* s, bool block, guber_engine* holder) {   // engine mutex held; 1 = resolved, 0 = still running
    guber_engine* e = s->e;
    if (s->mode == 4) return resolve_routed_parts_locked(s, holder ? holder : e, block);
    if (s->mode != 3) return 1;
    volatile unsigned int* flag = &s->sout->done;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != s->seq) {
        if (!block) return 0;
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(e->stream)); break; }
    }
    if (e->small_pending == s) e->small_pending = nullptr;
    if (!s->sout->fallback) {
        e->last_ctr.over += s->sout->over; e->last_ctr.hits += s->sout->hits; e->last_ctr.misses += s->sout->misses; e->last_ctr.size += s->sout->size_delta;
        s->mode = 1;
        return 1;
    }
    e->small_fallbacks++;
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const guber_batch_t& b = s->batch;
    BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                b.greg_expire, b.greg_duration, b.now_ms};
    ResultView R{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
    int rc = launch_batch(e, B, R, true);
    if (rc) { s->mode = 0; return rc; }
    HIPCHK(hipEventRecord(s->ev, e->stream));
    s->gev = nullptr; s->mode = 2;
    return 1;
}
static int resolve_small(guber_stage* s, bool block) {
    std::lock_guard<std::mutex> lk(s->e->mu);
    return resolve_small_locked(s, block);
}
// every share of a routed small stage (mode 4), each under its engine's mutex; all resolved: the stage is complete (mode 1)
static int resolve_routed_small(guber_stage* s, bool block) {
    for (size_t k = 0; k < s->parts.size(); ++k) {
        guber_engine* e = s->parts[k].e;
        std::lock_guard<std::mutex> lk(e->mu);
        if (s->parts[k].rc) return s->parts[k].rc;
        if (!s->parts[k].pending) continue;
        const int rc = resolve_routed_parts_locked(s, e, block);
        if (rc <= 0) return rc;
    }
    s->mode = 1;
    return 1;
}

extern "C" int guber_stage_poll(guber_stage_t* s) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    if (s->mode == 3) {
        const int rc = resolve_small(s, false);
        if (rc <= 0) return rc;
    }
    if (s->mode == 4) {
        const int rc = resolve_routed_small(s, false);
        if (rc <= 0) return rc;
    }
    if (s->mode == 2) {
        if (s->gev && s->gev->seq.load(std::memory_order_acquire) != s->gev_seq) return 1;   // the group's event slot has moved on: complete
        const hipError_t q = hipEventQuery(s->gev ? s->gev->ev : s->ev);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return fail(GUBER_E_HIP, "hipEventQuery", q);
    }
    return 1;
}

namespace {
struct StagePlan { guber_stage* s; BatchView B; ResultView R; bool copy; StageIn in; };
}
// the views of a stage batch: it reaches HBM through the copy kernel (the stage's device mirror; one PCIe round trip for all of
// it, where k_front reading host memory in place pays one per dependent load: key offset, key bytes, fields) unless the engine
// was told otherwise (GUBER_STAGE_COPY_MIN / GUBER_NO_STAGE_DMA: k_front then keeps a copy of the request columns for k_eval2)
static int stage_views(guber_stage* s, StagePlan& P) {
    guber_engine* e = s->e;
    const guber_batch_t& b = s->batch;
    P.s = s;
    P.B = BatchView{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                    b.greg_expire, b.greg_duration, b.now_ms};
    P.R = ResultView{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
    P.copy = e->stage_dma && b.n >= e->stage_copy_min && !b.greg_expire && !b.greg_duration;
    P.in = StageIn{};
    if (!P.copy) return 0;
    const size_t in_bytes = (size_t)(s->h_out - s->h_in);
    if (s->dmem.ensure(in_bytes)) return GUBER_E_NOMEM;
    uint8_t* d_in = s->dmem.p;
    P.in.src = (const uint4*)s->h_in; P.in.dst = (uint4*)d_in;
    auto seg = [&](const void* col, size_t bytes) {                 // the first `bytes` of a column that is present
        if (!col || !bytes) return;
        P.in.off16[P.in.nseg] = (uint32_t)(((const uint8_t*)col - s->h_in) / 16);
        P.in.n16[P.in.nseg] = (uint32_t)((bytes + 15) / 16);
        P.in.nseg++;
    };
    const size_t n = b.n;
    seg(b.key_off, (n + 1) * 4); seg(b.hits, n * 8); seg(b.limit, n * 8); seg(b.duration, n * 8); seg(b.behavior, n * 4);
    seg(b.algorithm, n); seg(b.is_owner, n); seg(b.burst, n * 8); seg(b.created_at, n * 8); seg(b.key_bytes, (size_t)b.key_off[b.n] + 16);
    auto dev = [&](const void* hp) { return hp ? d_in + ((const uint8_t*)hp - s->h_in) : nullptr; };
    P.B = BatchView{b.n, 0, dev(b.key_bytes), (const uint32_t*)dev(b.key_off), (const int64_t*)dev(b.hits), (const int64_t*)dev(b.limit),
                    (const int64_t*)dev(b.duration), (const int64_t*)dev(b.burst), (const int64_t*)dev(b.created_at), dev(b.algorithm),
                    (const uint32_t*)dev(b.behavior), dev(b.is_owner), nullptr, nullptr, b.now_ms};
    return 0;
}

// one group: <= MULTI_MEM_MAX large stages of engines that share device and stream (engine mutexes held by the caller).  Up to
// MULTI_MAX of them take the launches whose arguments travel by value; more (a pool dispatcher's generation over 8, 12 shards)
// take the same launches with their argument blocks in device memory: written into the leading stage's host block here,
// brought over by the copy kernel that also moves the request columns.  Three launches and one event record per group.
static int launch_stage_group(StagePlan* P, int g) {
    guber_engine* e0 = P[0].s->e;
    const bool mem_args = g > MULTI_MAX;
    MultiStageIn MI{}; MultiFront MF{}; MultiEval ME{};
    MultiArgsMem* HA = P[0].s->h_margs;
    uint32_t tiles = 0; int planned = 0, rc = 0; bool any_copy = false;
    FastPlan FP[MULTI_MEM_MAX];
    for (int i = 0; i < g; ++i) {
        guber_engine* e = P[i].s->e;
        Work W;
        rc = batch_prelude(e, P[i].B, W);
        if (!rc) rc = plan_fast(e, P[i].B, !P[i].copy, W, FP[i]);
        if (rc) break;
        tiles += FP[i].ftiles;
        if (mem_args) {
            HA->F.end_tile[planned] = HA->E.end_tile[planned] = tiles;
            HA->F.sub[planned] = FrontArgs{e->T, FP[i].B2, FP[i].W};
            HA->E.sub[planned] = EvalArgs{e->T, FP[i].B3, P[i].R, FP[i].W};
        } else {
            MF.end_tile[planned] = ME.end_tile[planned] = tiles;
            MF.sub[planned] = FrontArgs{e->T, FP[i].B2, FP[i].W};
            ME.sub[planned] = EvalArgs{e->T, FP[i].B3, P[i].R, FP[i].W};
        }
        MI.sub[planned] = P[i].in;
        any_copy = any_copy || P[i].copy;
        ++planned;
    }
    if (!planned) return rc;
    hipStream_t st = e0->stream;
    MultiArgsMem* DA = nullptr;
    MI.nb = (uint32_t)planned;
    if (mem_args) {
        if (e0->d_margs.ensure(sizeof(MultiArgsMem))) return GUBER_E_NOMEM;
        DA = (MultiArgsMem*)e0->d_margs.p;
        HA->F.nb = HA->E.nb = (uint32_t)planned;
        StageIn& a = MI.sub[MI.nb++];                                // the argument blocks: one more segment list of the copy kernel
        a = StageIn{};
        a.src = (const uint4*)HA; a.dst = (uint4*)DA; a.nseg = 2;
        a.off16[0] = 0; a.n16[0] = (uint32_t)((offsetof(MultiFrontMem, sub) + (size_t)planned * sizeof(FrontArgs) + 15) / 16);
        a.off16[1] = (uint32_t)(offsetof(MultiArgsMem, E) / 16); a.n16[1] = (uint32_t)((offsetof(MultiEvalMem, sub) + (size_t)planned * sizeof(EvalArgs) + 15) / 16);
    }
    if (any_copy || mem_args) {
        MI.wg_per = 64;
        hipLaunchKernelGGL(k_stage_in_multi, dim3(MI.nb * MI.wg_per), dim3(256), 0, st, MI);
    }
    uint64_t units = 0;
    for (int i = 0; i < planned; ++i) units += P[i].B.n;
    if (planned == 1) {
        e0->span_begin(KT_FRONT, units);
        hipLaunchKernelGGL(k_front, dim3(FP[0].ftiles), dim3(FT), 0, st, e0->T, FP[0].B2, FP[0].W);
        e0->span_end();
        e0->span_begin(KT_EVAL2, units);
        hipLaunchKernelGGL(k_eval2, dim3(FP[0].ftiles), dim3(256), 0, st, EvalArgs{e0->T, FP[0].B3, P[0].R, FP[0].W});
        e0->span_end();
    } else if (mem_args) {
        e0->span_begin(KT_FRONT_MULTI, units);
        hipLaunchKernelGGL(k_front_multi_mem, dim3(tiles), dim3(FT), 0, st, (const MultiFrontMem*)&DA->F);
        e0->span_end();
        e0->span_begin(KT_EVAL2_MULTI, units);
        hipLaunchKernelGGL(k_eval2_multi_mem, dim3(tiles), dim3(256), 0, st, (const MultiEvalMem*)&DA->E);
        e0->span_end();
    } else {
        MF.nb = ME.nb = (uint32_t)planned;
        e0->span_begin(KT_FRONT_MULTI, units);
        hipLaunchKernelGGL(k_front_multi, dim3(tiles), dim3(FT), 0, st, MF);
        e0->span_end();
        e0->span_begin(KT_EVAL2_MULTI, units);
        hipLaunchKernelGGL(k_eval2_multi, dim3(tiles), dim3(256), 0, st, ME);
        e0->span_end();
    }
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    guber_engine::GroupEv* G = nullptr; uint32_t gseq = 0;
    if (planned > 1) {
        G = &e0->gev[e0->gev_next++ % guber_engine::kGroupEvs];
        if (!G->ev) { if (hipEventCreateWithFlags(&G->ev, hipEventDisableTiming) != hipSuccess) return fail(GUBER_E_HIP, "hipEventCreate"); }
        else if (hipEventSynchronize(G->ev) != hipSuccess) return fail(GUBER_E_HIP, "hipEventSynchronize");   // (kGroupEvs groups back: long complete)
        gseq = G->seq.load(std::memory_order_relaxed) + 1;
        if (hipEventRecord(G->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
        G->seq.store(gseq, std::memory_order_release);
    }
    for (int i = 0; i < planned; ++i) {
        guber_stage* s = P[i].s;
        finish_fast(s->e, P[i].B.n);
        if (planned > 1) s->e->fused_batches++;
        s->gev = G; s->gev_seq = gseq;
        if (!G && hipEventRecord(s->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
        s->mode = 2;
    }
    return rc;
}

extern "C" int guber_stages_submit(guber_stage_t* const* stages, uint32_t n, uint32_t flags, uint32_t* done) {
    if (done) *done = 0;
    if (!stages && n) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (!(flags & GUBER_STAGES_NO_AGGREGATES)) {                  // per-batch aggregates wanted: the stages go one by one
        for (uint32_t k = 0; k < n; ++k) {
            const int rc = guber_stage_submit(stages[k]);
            if (rc) return rc;
            if (done) *done = k + 1;
        }
        return GUBER_OK;
    }
    for (uint32_t k = 0; k < n; ++k) {
        guber_stage* s = stages[k];
        if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
        if (s->mode) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
        const guber_batch_t& b = s->batch;
        if (b.n > s->max_n || (b.n && b.key_off[b.n] > s->key_cap)) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled");
        for (uint32_t q = 0; q < k; ++q) if (stages[q]->e == s->e) return fail(GUBER_E_INVALID_ARG, "two stages of one engine in one submission");
    }
    uint32_t enq = 0;
    StagePlan grp[MULTI_MEM_MAX]; int g = 0;
    const int group_max = MULTI_MEM_MAX;
    auto flush = [&]() -> int {
        if (!g) return 0;
        guber_engine* order[MULTI_MEM_MAX];
        for (int i = 0; i < g; ++i) order[i] = grp[i].s->e;
        std::sort(order, order + g);                               // engine locks in address order (launch_group's rule)
        for (int i = 0; i < g; ++i) { order[i]->mu.lock(); ep_flush_held(order[i]); }
        int rc = 0;
        if (grp[0].s->e->set_device()) rc = fail(GUBER_E_HIP, "hipSetDevice");
        for (int i = 0; i < g && !rc; ++i) {
            guber_engine* e = grp[i].s->e;
            if (e->small_pending) { const int r2 = resolve_small_locked(e->small_pending, true, e); if (r2 < 0) rc = r2; }
            if (!rc) rc = stage_views(grp[i].s, grp[i]);
        }
        if (!rc) rc = launch_stage_group(grp, g);
        for (int i = g - 1; i >= 0; --i) order[i]->mu.unlock();
        if (!rc) enq += (uint32_t)g;
        g = 0;
        return rc;
    };
    // batches of <= 256 requests of engines that share device and stream: ONE k_small_multi, one workgroup per batch
    guber_stage* sgrp[SMALL_MULTI_MAX]; int sg = 0;
    auto flush_small = [&]() -> int {
        if (!sg) return 0;
        guber_engine* order[SMALL_MULTI_MAX];
        for (int i = 0; i < sg; ++i) order[i] = sgrp[i]->e;
        std::sort(order, order + sg);
        for (int i = 0; i < sg; ++i) order[i]->mu.lock();
        int rc = 0, planned = 0;
        MultiSmall MS{};
        guber_engine* e0 = sgrp[0]->e;
        if (e0->set_device()) rc = fail(GUBER_E_HIP, "hipSetDevice");
        for (int i = 0; i < sg && !rc; ++i) {
            guber_stage* s = sgrp[i]; guber_engine* e = s->e;
            if (e->small_pending) { const int r2 = resolve_small_locked(e->small_pending, true, e); if (r2 < 0) { rc = r2; break; } }
            const guber_batch_t& b = s->batch;
            BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                        b.greg_expire, b.greg_duration, b.now_ms};
            rc = small_prelude(e, B);
            if (rc) break;
            s->seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
            s->sout->done = 0;
            MS.sub[planned] = SmallArgs{e->T, B, ResultView{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err},
                                        s->sout, e->touch, s->seq};
            ++planned;
        }
        if (planned) {
            MS.nb = (uint32_t)planned;
            if (planned == 1) hipLaunchKernelGGL(k_small, dim3(1), dim3(FT), 0, e0->stream, MS.sub[0].T, MS.sub[0].B, MS.sub[0].R, MS.sub[0].out, MS.sub[0].seq, MS.sub[0].touch);
            else hipLaunchKernelGGL(k_small_multi, dim3(planned), dim3(FT), 0, e0->stream, MS);
            if (hipGetLastError() != hipSuccess && !rc) rc = fail(GUBER_E_HIP, "kernel launch");
            for (int i = 0; i < planned; ++i) { sgrp[i]->mode = 3; sgrp[i]->e->small_pending = sgrp[i]; }
            enq += (uint32_t)planned;
        }
        for (int i = sg - 1; i >= 0; --i) order[i]->mu.unlock();
        sg = 0;
        return rc;
    };
    int rc = 0;
    for (uint32_t k = 0; k < n && !rc; ++k) {
        guber_stage* s = stages[k];
        guber_engine* e = s->e;
        const guber_batch_t& b = s->batch;
        s->n = b.n; s->now_ms = b.now_ms; s->no_agg = true;
        if (b.n == 0) { s->mode = 0; ++enq; continue; }
        memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);        // the kernels read keys as 8-byte words
        const bool small = b.n <= FT && !e->no_small && !lru_may_bind_unlocked(e, b.n);
        const bool fusable = !small && can_fuse(e, b.n);
        if (small) {
            if (sg && (sg == SMALL_MULTI_MAX || e->stream != sgrp[0]->e->stream || e->device != sgrp[0]->e->device)) rc = flush_small();
            if (rc) break;
            sgrp[sg++] = s;
            continue;
        }
        if (g && (!fusable || g == group_max || e->stream != grp[0].s->e->stream || e->device != grp[0].s->e->device)) rc = flush();
        if (rc) break;
        if (fusable) { grp[g++].s = s; continue; }
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->set_device()) { rc = fail(GUBER_E_HIP, "hipSetDevice"); break; }
        if (e->small_pending) { const int r2 = resolve_small_locked(e->small_pending, true, e); if (r2 < 0) { rc = r2; break; } }
        BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                    b.greg_expire, b.greg_duration, b.now_ms};
        ResultView R{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
        // the radix pipeline (n > 65 536) or a test configuration
        rc = launch_batch(e, B, R, true);
        if (rc) break;
        if (hipEventRecord(s->ev, e->stream) != hipSuccess) { rc = fail(GUBER_E_HIP, "hipEventRecord"); break; }
        s->gev = nullptr; s->mode = 2;
        ++enq;
    }
    if (!rc) rc = flush();
    if (!rc) rc = flush_small();
    if (done) {                                                      // the leading stages (array order) that were enqueued; after an
        uint32_t lead = 0;                                           // error the caller settles the others with guber_stage_wait
        while (lead < n && (stages[lead]->mode != 0 || stages[lead]->batch.n == 0)) ++lead;
        *done = rc ? lead : n;
    }
    (void)enq;
    return rc;
}

// ---- ONE stage for several engines: the device-level stage of a pool.  Callers fill it in arrival order and say, per request,
// which engine it belongs to and which place it has in that engine's share (guber_stage_dest); the copy kernel scatters the
// request columns into HBM so that every share is contiguous (k_stage_in_routed), the shares then run as the batches of ONE
// k_front_multi_mem + ONE k_eval2_multi_mem, and a last launch takes the answers back to the callers' slots.  Four launches
// and one event for a whole generation, whatever the number of shards; nothing on the host is proportional to the requests.
extern "C" uint32_t* guber_stage_dest(guber_stage_t* s) { return s ? s->h_dest : nullptr; }
// bytes of a routed stage's HBM mirror before the key bytes (guber_stage_submit_routed lays the request and answer columns out there)
static size_t routed_mirror_fixed(size_t cap) {
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    return 3 * col(cap * 4 + 4) + 5 * col(cap * 8) + col(cap * 4) + 2 * col(cap) + 3 * col(cap * 8) + 2 * col(cap);
}
// The routing of a front stage done by the device (k_route_count + k_route_dest): the callers wrote their requests in arrival
// order and nothing else; afterwards guber_stage_dest(s) holds what they would have written and *counts the shares' sizes —
// exactly the inputs of guber_stage_submit_routed.  The rule is the placement's (guber_placement_export); it is copied to the
// device when given (NULL = the one given last).  Never waits for the GPU except when a rule is uploaded (a placement change).
extern "C" int guber_stage_route(guber_stage_t* s, const guber_route_rule_t* rule, uint32_t n_engines) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    if (n_engines == 0 || n_engines > (uint32_t)MULTI_MEM_MAX) return fail(GUBER_E_INVALID_ARG, "1 .. 16 engines per routed stage");
    if (s->mode || s->route_pending) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
    const guber_batch_t& b = s->batch;
    if (b.n > s->max_n || b.n > 65536u || (b.n && b.key_off[b.n] > s->key_cap)) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled (a routed stage holds at most 65 536 requests)");
    if (!b.behavior) return fail(GUBER_E_INVALID_ARG, "a routed stage carries every request column");
    guber_engine* e = s->e;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    if (rule) {
        if (rule->n_shards == 0 || rule->per == 0 || !rule->table || rule->n_shards > 4096 || (rule->ex_cells & (rule->ex_cells - 1)) ||
            (rule->ex_n && (!rule->ex_hash || !rule->ex_shard || rule->ex_n >= rule->ex_cells)))
            return fail(GUBER_E_INVALID_ARG, "malformed route rule");
        const size_t slots = (size_t)rule->n_shards * rule->per, cells = rule->ex_cells ? rule->ex_cells : 1;
        if (e->d_rt_table.ensure(slots) || e->d_rt_exh.ensure(cells) || e->d_rt_exs.ensure(cells)) return GUBER_E_NOMEM;
        hipError_t he = hipStreamSynchronize(e->stream);                // (launches still reading the previous rule)
        if (he == hipSuccess) he = hipMemcpy(e->d_rt_table.p, rule->table, slots * 2, hipMemcpyHostToDevice);
        if (he == hipSuccess && rule->ex_n) he = hipMemcpy(e->d_rt_exh.p, rule->ex_hash, cells * 8, hipMemcpyHostToDevice);
        if (he == hipSuccess && rule->ex_n) he = hipMemcpy(e->d_rt_exs.p, rule->ex_shard, cells * 2, hipMemcpyHostToDevice);
        if (he != hipSuccess) { e->have_rule = false; return fail(GUBER_E_HIP, "guber_stage_route: rule upload", he); }
        e->rule = RouteRule{rule->n_shards, rule->per, rule->ex_cells, rule->ex_n, rule->global_engine, rule->step, rule->inv_step, rule->inv_sub,
                            e->d_rt_table.p, (const unsigned long long*)e->d_rt_exh.p, e->d_rt_exs.p};
        e->have_rule = true;
    }
    if (!e->have_rule) return fail(GUBER_E_INVALID_ARG, "guber_stage_route: no rule given yet");
    s->route_engines = n_engines;
    for (uint32_t j = 0; j < (uint32_t)MULTI_MEM_MAX; ++j) s->h_route[j] = 0;
    if (b.n == 0) { s->route_pending = false; return GUBER_OK; }
    const uint32_t tiles = (b.n + 255u) / 256u;
    const size_t tab = (size_t)256 * MULTI_MEM_MAX * 4;
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    const size_t cap = s->max_n, off_bytes = col(cap * 4 + 4), beh_bytes = col(cap * 4);
    const bool fresh = s->d_route.p == nullptr;
    if (s->d_route.ensure(2 * tab + 64 + col(cap) + off_bytes + beh_bytes) || s->dmem.ensure(routed_mirror_fixed(cap) + col((size_t)s->key_cap + 64))) return GUBER_E_NOMEM;
    if (fresh && hipMemsetAsync(s->d_route.p + 2 * tab, 0, 64, e->stream) != hipSuccess) return fail(GUBER_E_HIP, "hipMemsetAsync");
    memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);            // the kernels read keys as 8-byte words
    uint8_t* d_keys = s->dmem.p + routed_mirror_fixed(cap);           // (where guber_stage_submit_routed expects them)
    uint32_t* d_off = (uint32_t*)(s->d_route.p + 2 * tab + 64 + col(cap)); uint32_t* d_beh = (uint32_t*)((uint8_t*)d_off + off_bytes);
    RouteIn I{};
    I.src[0] = (const uint4*)b.key_bytes; I.dst[0] = (uint4*)d_keys; I.n16[0] = (uint32_t)(((size_t)b.key_off[b.n] + 16 + 15) / 16);
    I.src[1] = (const uint4*)b.key_off; I.dst[1] = (uint4*)d_off; I.n16[1] = (uint32_t)(((size_t)b.n * 4 + 4 + 15) / 16);
    I.src[2] = (const uint4*)b.behavior; I.dst[2] = (uint4*)d_beh; I.n16[2] = (uint32_t)(((size_t)b.n * 4 + 15) / 16);
    for (int k = 0; k < 3; ++k) I.nb[k] = std::max<uint32_t>(1u, std::min<uint32_t>(256u, (I.n16[k] + 1023) / 1024));
    hipLaunchKernelGGL(k_route_in, dim3(I.nb[0] + I.nb[1] + I.nb[2]), dim3(256), 0, e->stream, I);
    RouteArgs A{};
    A.n = b.n; A.n_engines = n_engines; A.max_key = e->max_key; A.seq = ++s->route_seq ? s->route_seq : ++s->route_seq;
    A.key_bytes = d_keys; A.key_off = d_off; A.behavior = d_beh;
    A.tile_cnt = (uint32_t*)s->d_route.p; A.tile_base = (uint32_t*)(s->d_route.p + tab); A.ticket = (uint32_t*)(s->d_route.p + 2 * tab);
    A.eng = s->d_route.p + 2 * tab + 64;
    A.dest = s->h_dest; A.counts = s->h_route; A.done = (unsigned int*)(s->h_route + MULTI_MEM_MAX);
    A.R = e->rule;
    hipLaunchKernelGGL(k_route_count, dim3(tiles), dim3(256), 0, e->stream, A);
    hipLaunchKernelGGL(k_route_dest, dim3(tiles), dim3(256), 0, e->stream, A);
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    s->route_pending = true; s->keys_resident = true;
    return GUBER_OK;
}
// 1 = the shares' sizes are in counts[0 .. n_engines) (guber_stage_dest is complete by the time anything enqueued later on the
// engines' stream runs: guber_stage_submit_routed may follow at once), 0 = still running
extern "C" int guber_stage_route_poll(guber_stage_t* s, uint32_t* counts) {
    if (!s || !counts) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (s->route_pending) {
        if (__atomic_load_n((volatile unsigned int*)(s->h_route + MULTI_MEM_MAX), __ATOMIC_ACQUIRE) != s->route_seq) return 0;
        s->route_pending = false;
    }
    for (uint32_t j = 0; j < s->route_engines; ++j) counts[j] = s->h_route[j];
    return 1;
}
extern "C" int guber_stage_submit_routed(guber_stage_t* s, guber_engine_t* const* engines, uint32_t n_engines, const uint32_t* counts) {
    if (!s || !engines || !counts) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n_engines == 0 || n_engines > (uint32_t)MULTI_MEM_MAX) return fail(GUBER_E_INVALID_ARG, "1 .. 16 engines per routed stage");
    if (s->mode) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
    const guber_batch_t& b = s->batch;
    const bool keys_there = s->keys_resident;                        // (guber_stage_route brought this batch's key bytes to the HBM mirror: same bytes, same place)
    s->keys_resident = false;
    s->n = b.n; s->now_ms = b.now_ms; s->no_agg = true; s->routed.clear();
    if (b.n > s->max_n || (b.n && b.key_off[b.n] > s->key_cap)) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled");
    if (b.greg_expire || b.greg_duration) return fail(GUBER_E_INVALID_ARG, "a routed stage takes its calendar intervals from the device");
    if (!b.burst || !b.created_at || !b.behavior || !b.algorithm || !b.is_owner) return fail(GUBER_E_INVALID_ARG, "a routed stage carries every request column");
    uint64_t total = 0; bool own = false;
    for (uint32_t j = 0; j < n_engines; ++j) {
        guber_engine* e = engines[j];
        if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
        if (e->device != s->e->device || e->stream != s->e->stream) return fail(GUBER_E_INVALID_ARG, "the engines of a routed stage share device and stream");
        for (uint32_t q = 0; q < j; ++q) if (engines[q] == e) return fail(GUBER_E_INVALID_ARG, "an engine twice in one routed stage");
        if (counts[j] && !fits_fused(e, counts[j])) return fail(GUBER_E_BATCH_TOO_LARGE, "an engine's share is larger than its two-launch pipeline takes");
        own = own || e == s->e;
        total += counts[j];
    }
    if (!own) return fail(GUBER_E_INVALID_ARG, "the stage's own engine is one of the engines");
    if (total != b.n) return fail(GUBER_E_INVALID_ARG, "the shares do not add up to the batch");
    if (b.n == 0) { s->mode = 0; return GUBER_OK; }
    memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);            // the kernels read keys as 8-byte words
    guber_engine* order[MULTI_MEM_MAX];
    for (uint32_t j = 0; j < n_engines; ++j) order[j] = engines[j];
    std::sort(order, order + n_engines);                             // engine locks in address order (launch_group's rule)
    for (uint32_t j = 0; j < n_engines; ++j) { order[j]->mu.lock(); ep_flush_held(order[j]); }
    struct Unlock { guber_engine** o; uint32_t n; ~Unlock() { for (uint32_t j = n; j-- > 0;) o[j]->mu.unlock(); } } unlock{order, n_engines};
    guber_engine* e0 = s->e;
    if (e0->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    for (uint32_t j = 0; j < n_engines; ++j)
        if (engines[j]->small_pending) { const int r2 = resolve_small_locked(engines[j]->small_pending, true, engines[j]); if (r2 < 0) return r2; }
    if (b.n <= FT) {                                                 // a handful of requests: ONE launch, a workgroup per share, in place
        bool small_ok = true;
        for (uint32_t j = 0; j < n_engines; ++j) small_ok = small_ok && !engines[j]->no_small && !(counts[j] && lru_may_bind(engines[j], counts[j]));
        if (small_ok) {
            MultiSmallRouted MS{};
            s->parts.clear();
            BatchView BH{0, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner, nullptr, nullptr, b.now_ms};
            for (uint32_t j = 0; j < n_engines; ++j) {
                if (!counts[j]) continue;
                guber_engine* e = engines[j];
                BatchView Bj = BH; Bj.n = counts[j];
                const int rc = small_prelude(e, Bj);
                if (rc) { s->parts.clear(); return rc; }             // (nothing has been launched; the stage stays idle)
                const uint32_t seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
                SmallOut* out = (SmallOut*)(s->h_parts_out + 64 * s->parts.size());
                out->done = 0;
                MS.sub[s->parts.size()] = SmallRoutedSub{e->T, out, e->touch, seq, counts[j], j};
                s->parts.push_back(guber_stage::RoutedPart{e, j, counts[j], seq, out, true, 0});
            }
            MS.nb = (uint32_t)s->parts.size(); MS.n_total = b.n; MS.dest = s->h_dest; MS.B = BH;
            MS.R = ResultView{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
            hipLaunchKernelGGL(k_small_routed, dim3(MS.nb), dim3(FT), 0, e0->stream, MS);
            if (hipGetLastError() != hipSuccess) { s->parts.clear(); return fail(GUBER_E_HIP, "kernel launch"); }
            for (auto& part : s->parts) part.e->small_pending = s;
            s->routed.assign(engines, engines + n_engines);
            s->gev = nullptr; s->mode = 4;
            return GUBER_OK;
        }
    }
    // the HBM mirror: every fixed-width column for max_n requests (each 64-byte aligned), where the request came from, the keys
    const size_t n = b.n, cap = s->max_n;
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    const size_t fixed = routed_mirror_fixed(cap);
    if (s->dmem.ensure(fixed + col((size_t)s->key_cap + 64)) || e0->d_margs.ensure(sizeof(MultiArgsMem))) return GUBER_E_NOMEM;

    uint8_t* p = s->dmem.p;
    RoutedIn A{};
    A.d_key_off = (uint32_t*)p; p += col(cap * 4 + 4); A.d_key_len = (uint32_t*)p; p += col(cap * 4 + 4); A.d_fwd = (uint32_t*)p; p += col(cap * 4 + 4);
    A.d_hits = (int64_t*)p; p += col(cap * 8); A.d_limit = (int64_t*)p; p += col(cap * 8); A.d_duration = (int64_t*)p; p += col(cap * 8);
    A.d_burst = (int64_t*)p; p += col(cap * 8); A.d_created_at = (int64_t*)p; p += col(cap * 8);
    A.d_behavior = (uint32_t*)p; p += col(cap * 4); A.d_algorithm = p; p += col(cap); A.d_is_owner = p; p += col(cap);
    RoutedOut O{};                                                   // the answers: HBM in the shares' order, then home in arrival order
    O.n = (uint32_t)n; O.fwd = A.d_fwd;
    int64_t* o_limit = (int64_t*)p; p += col(cap * 8); int64_t* o_remaining = (int64_t*)p; p += col(cap * 8); int64_t* o_reset = (int64_t*)p; p += col(cap * 8);
    uint8_t* o_status = p; p += col(cap); uint8_t* o_err = p; p += col(cap);
    O.d_status = o_status; O.d_err = o_err; O.d_limit = o_limit; O.d_remaining = o_remaining; O.d_reset_time = o_reset;
    O.status = s->result.status; O.err = s->result.err; O.limit = s->result.limit; O.remaining = s->result.remaining; O.reset_time = s->result.reset_time;
    uint8_t* d_keys = p;
    A.n = (uint32_t)n; A.dest = s->h_dest;
    A.key_off = b.key_off; A.hits = b.hits; A.limit = b.limit; A.duration = b.duration; A.burst = b.burst; A.created_at = b.created_at;
    A.behavior = b.behavior; A.algorithm = b.algorithm; A.is_owner = b.is_owner;
    A.key_src = (const uint4*)b.key_bytes; A.key_dst = (uint4*)d_keys; A.key_n16 = (uint32_t)(((size_t)b.key_off[b.n] + 16 + 15) / 16);
    MultiArgsMem* HA = s->h_margs; MultiArgsMem* DA = (MultiArgsMem*)e0->d_margs.p;
    uint32_t tiles = 0, base = 0; int planned = 0;
    guber_engine* took[MULTI_MEM_MAX]; uint32_t took_n[MULTI_MEM_MAX];
    // a share that may overflow its engine's cache needs the eviction pre-pass (launch_batch), which reads the share's keys: then the
    // shares are brought to HBM first and evaluated engine by engine
    bool exact = false;
    for (uint32_t j = 0; j < n_engines; ++j) exact = exact || (counts[j] && lru_may_bind(engines[j], counts[j]));
    BatchView XB[MULTI_MEM_MAX]; ResultView XR[MULTI_MEM_MAX];
    for (uint32_t j = 0; j < n_engines; ++j) {
        A.base[j] = base;
        const uint32_t nj = counts[j];
        if (!nj) continue;
        guber_engine* e = engines[j];
        BatchView B{nj, 0, d_keys, A.d_key_off + base, A.d_hits + base, A.d_limit + base, A.d_duration + base, A.d_burst + base, A.d_created_at + base,
                    A.d_algorithm + base, A.d_behavior + base, A.d_is_owner + base, nullptr, nullptr, b.now_ms, 0, A.d_key_len + base};
        ResultView R{o_status + base, o_limit + base, o_remaining + base, o_reset + base, o_err + base};
        if (exact) { XB[planned] = B; XR[planned] = R; took[planned] = e; took_n[planned] = nj; ++planned; base += nj; continue; }
        Work W; FastPlan FP;
        int rc = batch_prelude(e, B, W);
        if (!rc) rc = plan_fast(e, B, false, W, FP);
        if (rc) return rc;                                           // (nothing has been launched; the stage stays idle)
        tiles += FP.ftiles;
        HA->F.end_tile[planned] = HA->E.end_tile[planned] = tiles;
        HA->F.sub[planned] = FrontArgs{e->T, FP.B2, FP.W};
        HA->E.sub[planned] = EvalArgs{e->T, FP.B3, R, FP.W};
        took[planned] = e; took_n[planned] = nj;
        ++planned;
        base += nj;
    }
    HA->F.nb = HA->E.nb = (uint32_t)planned;
    A.arg_src = (const uint4*)HA; A.arg_dst = (uint4*)DA;
    A.arg_off16[0] = 0; A.arg_n16[0] = (uint32_t)((offsetof(MultiFrontMem, sub) + (size_t)planned * sizeof(FrontArgs) + 15) / 16);
    A.arg_off16[1] = (uint32_t)(offsetof(MultiArgsMem, E) / 16); A.arg_n16[1] = (uint32_t)((offsetof(MultiEvalMem, sub) + (size_t)planned * sizeof(EvalArgs) + 15) / 16);
    A.nb_req = (uint32_t)((n + 255) / 256);
    A.nb_key = keys_there ? 0u : std::max<uint32_t>(1u, std::min<uint32_t>(256u, (A.key_n16 + 1023) / 1024));
    A.nb_arg = 4;
    hipStream_t st = e0->stream;
    hipLaunchKernelGGL(k_stage_in_routed, dim3(A.nb_req + A.nb_key + A.nb_arg), dim3(256), 0, st, A);
    if (exact) {
        for (int i = 0; i < planned; ++i) { const int rc = launch_batch(took[i], XB[i], XR[i]); if (rc) return rc; }
        hipLaunchKernelGGL(k_stage_out_routed, dim3(A.nb_req), dim3(256), 0, st, O);
        if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
        if (hipEventRecord(s->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
        s->routed.assign(engines, engines + n_engines);
        s->gev = nullptr; s->mode = 2;
        return GUBER_OK;
    }
    e0->span_begin(KT_FRONT_MULTI, n);
    hipLaunchKernelGGL(k_front_multi_mem, dim3(tiles), dim3(FT), 0, st, (const MultiFrontMem*)&DA->F);
    e0->span_end();
    e0->span_begin(KT_EVAL2_MULTI, n);
    hipLaunchKernelGGL(k_eval2_multi_mem, dim3(tiles), dim3(256), 0, st, (const MultiEvalMem*)&DA->E);
    e0->span_end();
    hipLaunchKernelGGL(k_stage_out_routed, dim3(A.nb_req), dim3(256), 0, st, O);
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    for (int i = 0; i < planned; ++i) { finish_fast(took[i], took_n[i]); if (planned > 1) took[i]->fused_batches++; }
    if (hipEventRecord(s->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
    s->routed.assign(engines, engines + n_engines);
    s->gev = nullptr; s->mode = 2;
    return GUBER_OK;
}

extern "C" int guber_eval_batch(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r) { return eval_batch_host(e, b, r, nullptr); }
extern "C" int guber_eval_batch_store(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r, guber_store_events_t* ev) {
    if (!ev) return fail(GUBER_E_INVALID_ARG, "null store events");
    return eval_batch_host(e, b, r, ev);
}

// Store.Get is due for a request whose key is not resident when the request is applied (algorithms.go:45-51,
// :274-280): report the keys that are absent or expired at now_ms BEFORE the batch, so that the host can ask
// the Store and hand what it finds to guber_add_items first.
extern "C" int guber_probe_missing(guber_engine_t* e, const guber_batch_t* b, uint8_t* missing) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    if (!b || (b->n && (!b->key_bytes || !b->key_off || !missing))) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (b->n == 0) return GUBER_OK;
    if (b->n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const uint32_t n = b->n;
    const size_t kbytes = b->key_off[n] - b->key_off[0];
    if (e->d_keys.ensure(kbytes + 16) || e->d_off.ensure(n + 1) || e->d_out8.ensure((size_t)n * 2) || e->h_stage.ensure(kbytes + 16 + (size_t)(n + 1) * 4 + n + 64))
        return GUBER_E_NOMEM;
    uint32_t* soff = (uint32_t*)e->h_stage.p;
    uint8_t* skeys = e->h_stage.p + (size_t)(n + 1) * 4;
    uint8_t* sout = skeys + ((kbytes + 16 + 7) & ~(size_t)7);
    for (uint32_t i = 0; i <= n; ++i) soff[i] = b->key_off[i] - b->key_off[0];
    memcpy(skeys, b->key_bytes + b->key_off[0], kbytes);
    memset(skeys + kbytes, 0, 16);
    hipStream_t st = e->stream;
    HIPCHK(hipMemcpyAsync(e->d_keys.p, skeys, kbytes + 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->d_off.p, soff, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_probe_missing, dim3((n + 255) / 256), dim3(256), 0, st, e->T, e->d_keys.p, e->d_off.p, n, b->now_ms, e->d_out8.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sout, e->d_out8.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(missing, sout, n);
    return GUBER_OK;
}

// ---------------------------------------------------------------------------------------------
static Rec rec_from_item(const guber_item_t& in) {
    Rec s; rec_clear(s);
    s.limit = in.limit; s.duration = in.duration; s.stamp = in.stamp; s.burst = in.burst;
    s.expire_at = in.expire_at; s.invalid_at = in.invalid_at;
    if (in.algorithm == GUBER_ALGO_TOKEN_BUCKET) { s.remaining = in.remaining; s.burst = 0; s.meta = make_meta(K_TOKEN, in.status, ALGO_TOKEN); }
    else if (in.algorithm == GUBER_ALGO_LEAKY_BUCKET) { s.remaining = f2bits(in.remaining_f); s.meta = make_meta(K_LEAKY, 0, ALGO_LEAKY); }
    else s.meta = make_meta(K_NIL, 0, in.algorithm);   // gubernator.go:435-455: no Value for other algorithms
    return s;
}
static void item_from_rec(const Rec& s, guber_item_t* out) {
    memset(out, 0, sizeof(*out));
    out->limit = s.limit; out->duration = s.duration; out->stamp = s.stamp; out->burst = s.burst;
    out->expire_at = s.expire_at; out->invalid_at = s.invalid_at;
    if (rec_kind(s) == K_TOKEN) { out->algorithm = GUBER_ALGO_TOKEN_BUCKET; out->status = (uint8_t)rec_status(s); out->remaining = s.remaining; out->burst = 0; }
    else if (rec_kind(s) == K_LEAKY) { out->algorithm = GUBER_ALGO_LEAKY_BUCKET; out->remaining_f = bits2f(s.remaining); }
    else {   // CacheItem without a Value: only the CacheItem fields exist
        out->algorithm = (uint8_t)rec_algo(s);
        out->limit = out->duration = out->stamp = out->burst = 0;
    }
}

static int add_items_once(guber_engine* e, const guber_item_t* items, const std::vector<uint32_t>& sel, uint8_t* res_out, uint64_t stamp0) {
    const uint32_t n = (uint32_t)sel.size();
    size_t kbytes = 0;
    for (uint32_t j : sel) kbytes += items[j].key_len;
    std::vector<ItemIn> host(n);
    std::vector<uint8_t> keys(kbytes + 16, 0);
    size_t off = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const guber_item_t& it = items[sel[j]];
        host[j].rec = rec_from_item(it);
        rec_set_stamp(host[j].rec, stamp0 + sel[j]);             // the item's place in the CALL (lrucache.go:91,96: Add moves to the front, item by item)
        host[j].key_off = (uint32_t)off; host[j].key_len = it.key_len;
        if (it.key_len) memcpy(keys.data() + off, it.key, it.key_len);
        off += it.key_len;
    }
    // engine-owned scratch (grown on demand, kept): no allocation on the AddCacheItem / UpdatePeerGlobals path
    DevBuf<ItemIn>& d_items = e->d_items; DevBuf<uint8_t>&d_keys = e->d_ikeys, &d_flags = e->d_iflags, &d_res = e->d_ires;
    DevBuf<uint32_t>& d_slots = e->d_islots;
    int rc = 0;
    rc |= d_items.ensure(n); rc |= d_keys.ensure(keys.size()); rc |= d_flags.ensure(n); rc |= d_res.ensure(n); rc |= d_slots.ensure(n);
    auto cleanup = [&]() {};
    if (rc) return GUBER_E_NOMEM;
    hipStream_t st = e->stream;
    hipError_t he;
    if ((he = hipMemcpyAsync(d_items.p, host.data(), n * sizeof(ItemIn), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (he = hipMemcpyAsync(d_keys.p, keys.data(), keys.size(), hipMemcpyHostToDevice, st)) != hipSuccess) {
        cleanup(); return fail(GUBER_E_HIP, "add_items H2D", he);
    }
    hipLaunchKernelGGL(k_items_probe, dim3((n + 255) / 256), dim3(256), 0, st, e->T, d_items.p, d_keys.p, n, d_slots.p, d_flags.p);
    hipLaunchKernelGGL(k_items_commit, dim3((n + 255) / 256), dim3(256), 0, st, e->T, d_items.p, d_keys.p, n, d_slots.p, d_flags.p, d_res.p, ITEMS_KEEP_STAMP);
    std::vector<uint8_t> res(n);
    if ((he = hipMemcpyAsync(res.data(), d_res.p, n, hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipStreamSynchronize(st)) != hipSuccess) {
        cleanup(); return fail(GUBER_E_HIP, "add_items D2H", he);
    }
    cleanup();
    for (uint32_t j = 0; j < n; ++j) res_out[sel[j]] = res[j];
    return 0;
}

static int add_items_locked(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed);
extern "C" int guber_add_items(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed) {
    if (!e || (!items && n)) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n == 0) return GUBER_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    return add_items_locked(e, items, n, existed);
}
static int add_items_locked(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed) {
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    for (uint32_t i = 0; i < n; ++i) {
        if (!items[i].key || items[i].key_len == 0) return fail(GUBER_E_INVALID_ARG, "item without a key");
        if (items[i].key_len > e->max_key) return fail(GUBER_E_KEY_TOO_LONG, "item key longer than max_key_bytes");
    }
    {
        const int rc = maintain(e, n, e->clock_ms);
        if (rc) return rc;
    }
    note_enqueued(e, n);
    // LRUCache.Add is applied item by item (workers.go:566-581): with duplicates of a key in one call
    // the LAST one must win and the later ones report existed = 1.  Waves of distinct keys keep that; every item carries the
    // recency number of its place in the call, so the order among the call's keys is the reference's too (round 4 numbered the
    // items wave by wave: [A, A, D] left A in front of D).
    const uint64_t stamp0 = take_stamps(e, n);
    std::vector<uint8_t> res(n, 0);
    std::vector<uint32_t> pending(n);
    for (uint32_t i = 0; i < n; ++i) pending[i] = i;
    int guard = 0;
    while (!pending.empty()) {
        if (++guard > 64) return fail(GUBER_E_HIP, "add_items did not converge");
        std::unordered_map<std::string, int> seen;
        std::vector<uint32_t> wave, later;
        for (uint32_t i : pending) {
            std::string k((const char*)items[i].key, items[i].key_len);
            if (seen.emplace(std::move(k), 1).second) wave.push_back(i); else later.push_back(i);
        }
        int rc = add_items_once(e, items, wave, res.data(), stamp0);
        if (rc) return rc;
        std::vector<uint32_t> next;
        for (uint32_t i : wave) {
            if (res[i] == 0xFF) next.push_back(i);            // in-call hash collision: resubmit
            else if (res[i] == 0xFE) return fail(GUBER_E_TABLE_FULL, "no directory entry for item");
        }
        // keep original relative order for the next wave
        next.insert(next.end(), later.begin(), later.end());
        std::sort(next.begin(), next.end());
        pending.swap(next);
    }
    if (existed) for (uint32_t i = 0; i < n; ++i) existed[i] = res[i];
    return maintain(e, 0, e->clock_ms);       // Add evicts as soon as the cache is over its size (lrucache.go:98-100)
}

static int item_lookup(guber_engine* e, const uint8_t* key, uint32_t key_len, int64_t now_ms, int mode, guber_item_t* out, int* found) {
    if (!e || !key || !found) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    *found = 0;
    if (key_len == 0 || key_len > e->max_key) return GUBER_OK;
    DevBuf<uint8_t>& d_key = e->d_lkey; DevBuf<Rec>& d_rec = e->d_lrec; DevBuf<int>& d_found = e->d_lfound;   // engine-owned scratch
    int rc = d_key.ensure(key_len + 16) | d_rec.ensure(1) | d_found.ensure(1);
    auto cleanup = [&]() {};
    if (rc) return GUBER_E_NOMEM;
    std::vector<uint8_t> kb(key_len + 16, 0);
    memcpy(kb.data(), key, key_len);
    Rec hrec; int hfound = 0;
    hipStream_t st = e->stream;
    hipError_t he;
    if ((he = hipMemcpyAsync(d_key.p, kb.data(), kb.size(), hipMemcpyHostToDevice, st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "lookup H2D", he); }
    if (mode == 0 && now_ms > e->clock_ms) e->clock_ms = now_ms;
    hipLaunchKernelGGL(k_item_lookup, dim3(1), dim3(64), 0, st, e->T, d_key.p, key_len, now_ms, mode, d_rec.p, d_found.p, take_stamps(e, 1));
    if ((he = hipMemcpyAsync(&hfound, d_found.p, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipMemcpyAsync(&hrec, d_rec.p, sizeof(Rec), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipStreamSynchronize(st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "lookup D2H", he); }
    cleanup();
    *found = hfound;
    if (hfound && out) { item_from_rec(hrec, out); out->key = nullptr; out->key_len = key_len; }
    return GUBER_OK;
}

// A hot key changes its logical shard (GPUWorkerPool's placement): its bucket leaves `from`'s table and enters `to`'s, on the
// device (both engines live on one GPU).  The caller guarantees that no batch of either engine is being formed or is in
// flight for those keys (the pool quiesces its stages first).
extern "C" int guber_move_items_by_hash(guber_engine_t* from, guber_engine_t* to, const uint64_t* hashes, uint32_t n, uint32_t* moved) {
    if (moved) *moved = 0;
    if (!from || !to || from == to || (n && !hashes)) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (from->device != to->device) return fail(GUBER_E_INVALID_ARG, "engines on different devices");
    if (n == 0) return GUBER_OK;
    guber_engine* a = from < to ? from : to; guber_engine* b = from < to ? to : from;     // address order, as every multi-locker
    std::lock_guard<std::mutex> la(a->mu); std::lock_guard<std::mutex> lb(b->mu);
    ep_flush_held(a); ep_flush_held(b);
    if (from->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const uint32_t stride = (std::min(from->max_key, to->max_key) + 23u) & ~7u;
    DevBuf<uint64_t>& d_h = from->d_mvh;
    if (d_h.ensure(n) || from->d_items.ensure(n) || from->d_ikeys.ensure((size_t)n * stride + 16) || to->d_islots.ensure(n) || to->d_iflags.ensure(n) ||
        to->d_ires.ensure(n)) return GUBER_E_NOMEM;
    std::vector<uint8_t> res(n, 0xFF);
    hipError_t he = hipSuccess;
    int rc = 0;
    bool taken = false;
    do {
        if ((he = hipMemcpyAsync(d_h.p, hashes, (size_t)n * 8, hipMemcpyHostToDevice, from->stream)) != hipSuccess) break;
        if ((he = hipMemsetAsync(to->d_ires.p, 0xFF, n, from->stream)) != hipSuccess) break;      // "not taken over" until the commit says otherwise
        hipLaunchKernelGGL(k_items_take_by_hash, dim3((n + 63) / 64), dim3(64), 0, from->stream, from->T, d_h.p, n, stride, from->d_items.p, from->d_ikeys.p);
        if ((he = hipStreamSynchronize(from->stream)) != hipSuccess) break;
        taken = true;
        rc = maintain(to, n, to->clock_ms);
        if (rc) break;
        note_enqueued(to, n);
        hipStream_t st = to->stream;
        hipLaunchKernelGGL(k_items_probe, dim3((n + 255) / 256), dim3(256), 0, st, to->T, from->d_items.p, from->d_ikeys.p, n, to->d_islots.p, to->d_iflags.p);
        hipLaunchKernelGGL(k_items_commit, dim3((n + 255) / 256), dim3(256), 0, st, to->T, from->d_items.p, from->d_ikeys.p, n, to->d_islots.p, to->d_iflags.p,
                           to->d_ires.p, take_stamps(to, n));
        if ((he = hipMemcpyAsync(res.data(), to->d_ires.p, n, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
        he = hipStreamSynchronize(st);
    } while (0);
    if (taken) {
        // whatever the destination did not take over goes back into the source (the commit's verdicts, or 0xFF for everything
        // when it never ran): a migration that fails loses no bucket
        (void)hipStreamSynchronize(to->stream);
        hipLaunchKernelGGL(k_items_restore, dim3((n + 63) / 64), dim3(64), 0, from->stream, from->T, from->d_items.p, from->d_ikeys.p, n, to->d_ires.p);
        (void)hipMemcpyAsync(res.data(), to->d_ires.p, n, hipMemcpyDeviceToHost, from->stream);
        (void)hipStreamSynchronize(from->stream);
    }
    if (he != hipSuccess) return fail(GUBER_E_HIP, "guber_move_items_by_hash", he);
    if (rc) return rc;
    uint32_t m = 0, back = 0;
    for (uint32_t i = 0; i < n; ++i) { m += res[i] <= 1; back += res[i] == 0xFD; }   // (0xFE / 0xFF left: the hash named no live bucket)
    if (moved) *moved = m;
    if (back) return fail(GUBER_E_TABLE_FULL, "guber_move_items_by_hash: the destination did not take every bucket; those went back to their table");
    if (moved) *moved = m;
    return GUBER_OK;
}
extern "C" void* guber_engine_stream(guber_engine_t* e) { return e ? (void*)e->stream : nullptr; }

extern "C" int guber_get_item(guber_engine_t* e, const uint8_t* key, uint32_t key_len, int64_t now_ms, guber_item_t* out, int* found) {
    return item_lookup(e, key, key_len, now_ms, 0, out, found);
}
extern "C" int guber_remove_item(guber_engine_t* e, const uint8_t* key, uint32_t key_len) {
    int found = 0;
    return item_lookup(e, key, key_len, 0, 1, nullptr, &found);
}

extern "C" int guber_stats(guber_engine_t* e, guber_stats_t* out) {
    if (!e || !out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    rc = maintain(e, 0, e->clock_ms);
    if (rc) return rc;
    const DevCounters& c = e->last_ctr;
    out->over_limit_count = c.over; out->cache_hits = c.hits; out->cache_misses = c.misses;
    out->unexpired_evictions = c.evictions; out->cache_size = c.size; out->table_slots = e->slots;
    out->tags_used = c.tags_used; out->batches = e->batches; out->retries = c.retries; out->compactions = e->compactions;
    out->small_batches = e->small_batches; out->fused_batches = e->fused_batches;
    out->eviction_passes = e->lru_applied; out->tail_rebuilds = e->lru_rebuilds; out->batch_cuts = e->lru_cuts;
    return GUBER_OK;
}
extern "C" int64_t guber_size(guber_engine_t* e) {
    guber_stats_t s;
    if (guber_stats(e, &s) != GUBER_OK) return -1;
    return s.cache_size;
}
extern "C" int guber_synchronize(guber_engine_t* e) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamSynchronize(e->stream));
    return GUBER_OK;
}

extern "C" int guber_dump(guber_engine_t* e, guber_item_t* items, uint64_t cap, uint8_t* key_arena, uint64_t arena_cap,
                          uint64_t* n_out, uint64_t* arena_out) {
    if (!e || !n_out || !arena_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    const uint64_t resident = (uint64_t)std::max<long long>(e->last_ctr.size, 0);
    DevBuf<Rec> d_recs; DevBuf<KeyCell> d_cells; DevBuf<unsigned long long> d_count;
    rc = d_recs.ensure(resident + 1) | d_cells.ensure(resident + 1) | d_count.ensure(1);
    auto cleanup = [&]() { d_recs.release(); d_cells.release(); d_count.release(); };
    if (rc) { cleanup(); return GUBER_E_NOMEM; }
    hipStream_t st = e->stream;
    hipError_t he;
    unsigned long long count = 0;
    std::vector<Rec> recs(resident + 1);
    std::vector<KeyCell> cells(resident + 1);
    if ((he = hipMemsetAsync(d_count.p, 0, sizeof(unsigned long long), st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "dump", he); }
    hipLaunchKernelGGL(k_dump, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, st, e->T, e->slots, d_recs.p, d_cells.p, resident + 1, d_count.p);
    if ((he = hipMemcpyAsync(&count, d_count.p, sizeof(count), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipStreamSynchronize(st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "dump", he); }
    if (count > resident + 1) count = resident + 1;
    if ((he = hipMemcpy(recs.data(), d_recs.p, count * sizeof(Rec), hipMemcpyDeviceToHost)) != hipSuccess ||
        (he = hipMemcpy(cells.data(), d_cells.p, count * sizeof(KeyCell), hipMemcpyDeviceToHost)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "dump D2H", he); }
    cleanup();
    uint64_t need_arena = 0;
    for (uint64_t i = 0; i < count; ++i) need_arena += (uint32_t)(cells[i].w[7] >> 48);
    *n_out = count; *arena_out = need_arena;
    if (count > cap || need_arena > arena_cap || (!items && count) || (!key_arena && need_arena)) return fail(GUBER_E_NOMEM, "dump buffers too small");
    uint64_t aoff = 0;
    for (uint64_t i = 0; i < count; ++i) {
        item_from_rec(recs[i], &items[i]);
        const uint32_t len = (uint32_t)(cells[i].w[7] >> 48);
        uint8_t* dst = key_arena + aoff;
        if (len <= INLINE_KEY) memcpy(dst, cells[i].w, len);
        else if ((he = hipMemcpy(dst, e->arena.p + cells[i].w[0], len, hipMemcpyDeviceToHost)) != hipSuccess) return fail(GUBER_E_HIP, "dump long key", he);
        items[i].key = dst; items[i].key_len = len;
        aoff += len;
    }
    return GUBER_OK;
}

extern "C" void* guber_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void guber_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

// ---- device routing on the consistent-hash ring ------------------------------------------------
// ring image on the device, uploaded once per (engine, ring)
static int ensure_ring_on_device(guber_engine* e, const guber_ring_t* r) {
    if (e->ring_cached_id == guber_ring_id(r) && e->ring_npts) return 0;
    const uint32_t npts = guber_ring_points(r, nullptr, nullptr, 0);
    if (npts == 0) return fail(GUBER_E_INVALID_ARG, "empty ring");
    if ((size_t)npts * 8 > 150 * 1024) return fail(GUBER_E_INVALID_ARG, "ring does not fit in LDS");
    std::vector<uint64_t> hh(npts); std::vector<uint32_t> oo(npts);
    guber_ring_points(r, hh.data(), oo.data(), npts);
    if (e->d_ring_h.ensure(npts) || e->d_ring_o.ensure(npts)) return GUBER_E_NOMEM;
    HIPCHK(hipMemcpyAsync(e->d_ring_h.p, hh.data(), npts * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_ring_o.p, oo.data(), npts * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->ring_cached_id = guber_ring_id(r); e->ring_npts = npts;
    return 0;
}

extern "C" int guber_ring_route_dev(guber_engine_t* e, const guber_ring_t* r, const uint8_t* key_bytes,
                                    const uint32_t* key_off, uint32_t n, uint32_t* owner) {
    if (!e || !r || (n && (!key_bytes || !key_off || !owner))) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n == 0) return GUBER_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = ensure_ring_on_device(e, r);
    if (rc) return rc;
    hipLaunchKernelGGL(k_route, dim3((n + 255) / 256), dim3(256), (size_t)e->ring_npts * 8, e->stream, key_bytes, key_off, n,
                       e->d_ring_h.p, e->d_ring_o.p, e->ring_npts, guber_ring_kind(r), owner);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return GUBER_OK;
}

extern "C" int guber_ring_route_rows_dev(guber_engine_t* e, const guber_ring_t* r, const uint8_t* key_rows, uint32_t stride,
                                         const uint32_t* key_len, uint32_t n, uint32_t* owner) {
    if (!e || !r || (n && (!key_rows || !key_len || !owner))) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n == 0) return GUBER_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = ensure_ring_on_device(e, r);
    if (rc) return rc;
    hipLaunchKernelGGL(k_route_rows, dim3((n + 255) / 256), dim3(256), (size_t)e->ring_npts * 8, e->stream, key_rows, stride, key_len, n,
                       e->d_ring_h.p, e->d_ring_o.p, e->ring_npts, guber_ring_kind(r), owner);
    HIPCHK(hipGetLastError());
    return GUBER_OK;
}

extern "C" int guber_global_pending(guber_engine_t* e, uint32_t* n_out) {
    if (!e || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    *n_out = 0;
    if (!e->T.gpend) return fail(GUBER_E_INVALID_ARG, "engine created without GUBER_FLAG_GLOBAL");
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (c.gdirty_overflow) return fail(GUBER_E_NOMEM, "GLOBAL dirty list overflowed");
    *n_out = c.gdirty_n;
    return GUBER_OK;
}

// guber_global_take with the rows left in HBM, in caller-provided device arrays (cap rows each, key rows of
// out->key_stride bytes).  The rows feed guber_ring_route_rows_dev, an RCCL exchange and guber_eval_batch_dev /
// guber_add_items_dev without touching the host.
extern "C" int guber_global_take_dev(guber_engine_t* e, uint32_t role_mask, const guber_global_rows_dev_t* out, uint32_t* n_out) {
    if (!e || !out || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    *n_out = 0;
    if (!e->T.gpend) return fail(GUBER_E_INVALID_ARG, "engine created without GUBER_FLAG_GLOBAL");
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (c.gdirty_overflow) return fail(GUBER_E_NOMEM, "GLOBAL dirty list overflowed");
    const uint32_t n = c.gdirty_n;
    if (n == 0) return GUBER_OK;
    if (n > out->cap) { *n_out = n; return fail(GUBER_E_NOMEM, "row arrays too small"); }
    if (out->key_stride < e->max_key || !out->key_bytes || !out->key_len || !out->hits || !out->limit || !out->duration || !out->burst ||
        !out->created_at || !out->behavior || !out->algorithm || !out->role)
        return fail(GUBER_E_INVALID_ARG, "row arrays missing or key_stride < max_key_bytes");
    GTakeOut O{out->key_bytes, out->key_len, out->hits, out->limit, out->duration, out->burst, out->created_at, out->behavior,
               out->algorithm, out->role, out->key_stride};
    HIPCHK(hipMemsetAsync(e->gtake_ctr.p, 0, 4 * sizeof(uint32_t), e->stream));
    hipLaunchKernelGGL(k_global_take, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, n, role_mask, e->gdirty2.p,
                       (unsigned int*)e->gtake_ctr.p, O);
    unsigned int cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(cnt, e->gtake_ctr.p, sizeof(cnt), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    std::swap(e->gdirty.p, e->gdirty2.p);
    e->T.gdirty = e->gdirty.p;
    HIPCHK(hipMemcpyAsync(&e->ctr.p->gdirty_n, &cnt[1], sizeof(unsigned int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *n_out = cnt[0];
    return GUBER_OK;
}

// LRUCache.Add for device-resident item columns (keys must be distinct within one call: the receiver side of
// UpdatePeerGlobals, where every key comes from exactly one owner).  result[i] (device): 0 / 1 = existed,
// 0xFF = resubmit (in-call 64-bit hash collision or duplicate key), 0xFE = no directory entry.
extern "C" int guber_add_items_dev(guber_engine_t* e, const guber_items_dev_t* it, uint8_t* result) {
    if (!e || !it) return fail(GUBER_E_INVALID_ARG, "null argument");
    const uint32_t n = it->n;
    if (n == 0) return GUBER_OK;
    if (!result || !it->key_bytes || !it->key_off || !it->algorithm || !it->limit || !it->duration || !it->remaining || !it->remaining_f ||
        !it->stamp || !it->expire_at)
        return fail(GUBER_E_INVALID_ARG, "item column missing");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    {
        const int rc = maintain(e, n, e->clock_ms);
        if (rc) return rc;
    }
    note_enqueued(e, n);
    if (e->d_items.ensure(n) || e->d_islots.ensure(n) || e->d_iflags.ensure(n)) return GUBER_E_NOMEM;
    ItemsSoA S{it->key_off, it->algorithm, it->status, it->limit, it->duration, it->remaining, it->remaining_f, it->stamp, it->burst,
               it->expire_at, it->invalid_at};
    hipStream_t st = e->stream;
    hipLaunchKernelGGL(k_items_from_soa, dim3((n + 255) / 256), dim3(256), 0, st, S, n, e->d_items.p);
    hipLaunchKernelGGL(k_items_probe, dim3((n + 255) / 256), dim3(256), 0, st, e->T, e->d_items.p, it->key_bytes, n, e->d_islots.p, e->d_iflags.p);
    hipLaunchKernelGGL(k_items_commit, dim3((n + 255) / 256), dim3(256), 0, st, e->T, e->d_items.p, it->key_bytes, n, e->d_islots.p, e->d_iflags.p, result, take_stamps(e, n));
    HIPCHK(hipGetLastError());
    return GUBER_OK;
}

extern "C" int guber_global_take(guber_engine_t* e, uint32_t role_mask, guber_global_rows_t* out) {
    if (!e || !out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    memset(out, 0, sizeof(*out));
    if (!e->T.gpend) return fail(GUBER_E_INVALID_ARG, "engine created without GUBER_FLAG_GLOBAL");
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (c.gdirty_overflow) return fail(GUBER_E_NOMEM, "GLOBAL dirty list overflowed");
    const uint32_t n = c.gdirty_n;
    const uint32_t stride = (e->max_key + 7u) & ~7u;
    out->key_stride = stride;
    if (n == 0) return GUBER_OK;
    // one device + one pinned arena: keys | 5 x i64 | key_len u32 | behavior u32 | algorithm u8 | role u8
    const size_t o_keys = 0, o_i64 = (size_t)n * stride, o_len = o_i64 + (size_t)n * 40, o_beh = o_len + (size_t)n * 4,
                 o_alg = o_beh + (size_t)n * 4, o_role = o_alg + n, total = o_role + n + 64;
    int rc = e->d_take.ensure(total) | e->h_take.ensure(total);
    if (rc) return GUBER_E_NOMEM;
    uint8_t* d = e->d_take.p;
    GTakeOut O{d + o_keys, (uint32_t*)(d + o_len), (int64_t*)(d + o_i64), (int64_t*)(d + o_i64) + n, (int64_t*)(d + o_i64) + 2 * (size_t)n,
               (int64_t*)(d + o_i64) + 3 * (size_t)n, (int64_t*)(d + o_i64) + 4 * (size_t)n, (uint32_t*)(d + o_beh), d + o_alg, d + o_role, stride};
    HIPCHK(hipMemsetAsync(e->gtake_ctr.p, 0, 4 * sizeof(uint32_t), e->stream));
    hipLaunchKernelGGL(k_global_take, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, n, role_mask, e->gdirty2.p,
                       (unsigned int*)e->gtake_ctr.p, O);
    unsigned int cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(cnt, e->gtake_ctr.p, sizeof(cnt), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // rows that were not asked for stay queued: the kept list becomes the dirty list
    std::swap(e->gdirty.p, e->gdirty2.p);
    e->T.gdirty = e->gdirty.p;
    HIPCHK(hipMemcpyAsync(&e->ctr.p->gdirty_n, &cnt[1], sizeof(unsigned int), hipMemcpyHostToDevice, e->stream));
    const uint32_t m = cnt[0];
    if (m) HIPCHK(hipMemcpyAsync(e->h_take.p, d, total, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const uint8_t* h = e->h_take.p;
    out->n = m; out->key_bytes = h + o_keys; out->key_len = (const uint32_t*)(h + o_len);
    out->hits = (const int64_t*)(h + o_i64); out->limit = out->hits + n; out->duration = out->hits + 2 * (size_t)n;
    out->burst = out->hits + 3 * (size_t)n; out->created_at = out->hits + 4 * (size_t)n;
    out->behavior = (const uint32_t*)(h + o_beh); out->algorithm = h + o_alg; out->role = h + o_role;
    return GUBER_OK;
}

// Rebuild the table keeping only live buckets (and buckets with pending GLOBAL work).  Called explicitly (guber_compact) or
// by maintain() when the directory is above its load limit.
static int compact_table(guber_engine* e, int64_t now_ms) {
    quiesce_all(e);
    DevBuf<DirEntry> ndir; DevBuf<Bucket> nb; DevBuf<uint8_t> narena; DevBuf<GPend> ngp; DevBuf<CompactOut> d_out;
    int rc = ndir.ensure(e->slots) | nb.ensure(e->slots) | narena.ensure(e->T.arena_cap + 64) | d_out.ensure(1);
    if (e->T.gpend) rc |= ngp.ensure(e->slots);
    auto cleanup = [&]() { ndir.release(); nb.release(); narena.release(); ngp.release(); d_out.release(); };
    if (rc) { cleanup(); return GUBER_E_NOMEM; }
    hipError_t he;
    if ((he = hipMemsetAsync(ndir.p, 0, e->slots * sizeof(DirEntry), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(nb.p, 0, e->slots * sizeof(Bucket), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(d_out.p, 0, sizeof(CompactOut), e->stream)) != hipSuccess ||
        (ngp.p && (he = hipMemsetAsync(ngp.p, 0, e->slots * sizeof(GPend), e->stream)) != hipSuccess)) {
        cleanup();
        return fail(GUBER_E_HIP, "compaction", he);
    }
    Table N = e->T;
    N.dir = ndir.p; N.buckets = nb.p; N.arena = narena.p;
    if (e->T.gpend) { N.gpend = ngp.p; N.gdirty = e->gdirty2.p; }
    hipLaunchKernelGGL(k_compact, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, e->stream, e->T, e->slots, N, now_ms, d_out.p);
    CompactOut co{};
    if ((he = hipMemcpyAsync(&co, d_out.p, sizeof(co), hipMemcpyDeviceToHost, e->stream)) != hipSuccess ||
        (he = hipStreamSynchronize(e->stream)) != hipSuccess) {
        cleanup();
        return fail(GUBER_E_HIP, "compaction", he);
    }
    // tags_used = kept entries; the live count is unchanged except for the expired buckets that were dropped: recount it
    // from the kept entries that are live (kept - pending-but-dead is not tracked separately: size := live kept)
    DevCounters c;
    HIPCHK(hipMemcpy(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost));
    std::vector<BlockCounters> bc(e->n_bctr);
    HIPCHK(hipMemcpy(bc.data(), e->bctr.p, e->n_bctr * sizeof(BlockCounters), hipMemcpyDeviceToHost));
    for (auto& x : bc) x.size_delta = 0;
    c.size = (long long)co.live; c.tags_used = co.kept; c.arena_head = co.arena_head;
    if (e->T.gpend) c.gdirty_n = co.gdirty_n;
    HIPCHK(hipMemcpy(e->ctr.p, &c, sizeof(c), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->bctr.p, bc.data(), e->n_bctr * sizeof(BlockCounters), hipMemcpyHostToDevice));
    std::swap(e->dir.p, ndir.p); std::swap(e->buckets.p, nb.p); std::swap(e->arena.p, narena.p);
    std::swap(e->dir.cap, ndir.cap); std::swap(e->buckets.cap, nb.cap); std::swap(e->arena.cap, narena.cap);
    if (e->T.gpend) {
        std::swap(e->gpend.p, ngp.p); std::swap(e->gpend.cap, ngp.cap);
        std::swap(e->gdirty.p, e->gdirty2.p);
        e->T.gpend = e->gpend.p; e->T.gdirty = e->gdirty.p;
    }
    cleanup();
    e->T.dir = e->dir.p; e->T.buckets = e->buckets.p; e->T.arena = e->arena.p;
    e->tags_upper = co.kept; e->size_upper = co.live;
    e->last_ctr.size = (long long)co.live; e->last_ctr.tags_used = co.kept;
    rb_disarm_all(e);
    e->lru_tail_ok = false;                                          // the tail list names slots of the old table
    e->compactions++;
    return 0;
}

// Bring the cache down to cache_size: the least recently used items go, in the list's exact order (lrucache.go:98-100,138-149).
// Batches never leave the cache above its size (their pre-pass evicts as the reference does, in the middle of the batch); this is
// what Add / UpdatePeerGlobals / Load need — adding n items and then dropping the oldest leaves exactly the items the reference's
// item-by-item Add leaves — and the safety net behind everything else.
static int evict_to_size(guber_engine* e, int64_t now_ms) {
    uint32_t st = 0;
    const LruKeys none{};
    const int rc = lru_admit(e, none, 0, now_ms, &st);
    if (!rc) e->evict_passes++;
    return rc;
}

// Keep the cache within cache_size and the directory under its load limit before `incoming` more requests arrive.
// The bounds are upper bounds (every request in flight might create an item).  Near a limit, counter snapshots are kept on their
// way (one riding on every batch) and folded as they complete; the stream is drained only when an eviction / rebuild is really
// due or a HARD limit (physical room) is at stake.
// defer_hard (GUBER_FUSE_EP, launch_group): the caller is holding a k_eval3 back on this stream — anything that would enqueue, synchronise
// or read the counters must wait until that has been launched: *defer_hard = true and NOTHING is done; the caller launches it and calls again
static int maintain(guber_engine* e, uint64_t incoming, int64_t now_ms, bool batch_follows, bool* defer_hard) {
    const uint64_t tag_limit = e->slots - e->slots / 8;   // keep >= 1/8 of the entries free
    const uint64_t hard_size = e->cache_size + std::max<uint64_t>(e->cache_size / 2, 4ull * e->max_batch);
    if (e->rb_ride >= 0 && !batch_follows) {              // a snapshot that was to ride on a batch that never came: launch it now
        if (defer_hard) { *defer_hard = true; return 0; }
        const uint32_t i = (uint32_t)e->rb_ride;
        e->rb_ride = -1;
        rb_launch(e, i);
    }
    if (rb_any_armed(e)) rb_fold_newest(e);                // exact as of the newest completed snapshot + what was enqueued since
    // A call whose size bound reaches cache_size goes through the eviction pre-pass, which synchronises (lru_may_bind / lru_admit),
    // so the bound is tightened EARLY: from 16 calls' worth of requests below a limit on, every batch carries a snapshot, and the
    // bound a decision is taken on is the exact count a few batches ago plus what was enqueued since.
    const uint64_t early = std::min<uint64_t>(16 * incoming, e->cache_size / 2);
    const bool over = e->size_upper > e->cache_size, tags = e->tags_upper + incoming > tag_limit;
    const bool near = e->size_upper + incoming + early > e->cache_size || e->tags_upper + incoming + early > tag_limit;
    if (!over && !tags && !near) return 0;
    const bool sure_over = (uint64_t)std::max<long long>(e->last_ctr.size, 0) > e->cache_size && !rb_any_armed(e);
    const bool hard = (incoming == 0 && (over || tags)) || e->size_upper > hard_size || tags || sure_over;
    if (!hard) {
        if (batch_follows) (void)rb_arm(e, true);            // no launch of its own: the batch's first kernel carries it
        else if (defer_hard) { *defer_hard = true; return 0; }
        else if (!rb_any_armed(e)) { (void)rb_arm(e, false); HIPCHK(hipGetLastError()); }
        return 0;
    }
    if (defer_hard) { *defer_hard = true; return 0; }
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    if ((uint64_t)std::max<long long>(e->last_ctr.size, 0) > e->cache_size) {
        rc = evict_to_size(e, now_ms);
        if (rc) return rc;
    }
    if (e->tags_upper + incoming > tag_limit) {
        // dead entries (expired, removed, evicted) still hold their tags: rebuild without them
        rc = compact_table(e, now_ms);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int guber_compact(guber_engine_t* e, int64_t now_ms) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    if (now_ms > e->clock_ms) e->clock_ms = now_ms;
    return compact_table(e, now_ms);
}

// the engine has no clock of its own: `now` comes with every batch; maintenance between batches (eviction after Add) uses
// the latest value seen, which a caller with a frozen or external clock sets here (clock.Freeze / clock.Advance)
extern "C" int guber_set_clock(guber_engine_t* e, int64_t now_ms) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    e->clock_ms = now_ms;
    return GUBER_OK;
}

extern "C" int guber_profile_enable(guber_engine_t* e, int enable) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    e->profiling = enable != 0;
    return GUBER_OK;
}
extern "C" int guber_profile_read(guber_engine_t* e, guber_kernel_time_t* out, uint32_t cap, uint32_t* n_out) {
    if (!e || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamSynchronize(e->stream));
    {   // one pipeline pass = the spans from a first-stage kernel up to the next first-stage kernel
        // (k_evalpart_multi both ends a pass — the previous group's k_eval3 — and begins the next: a pass that is followed by one
        // lasts until the end of that launch)
        auto first_stage = [](int k) { return k == KT_FRONT || k == KT_FRONT_MULTI || k == KT_PART || k == KT_PART_MULTI || k == KT_RESOLVE || k == KT_EVALPART_MULTI; };
        // (a front's routing kernels run on a stream of their own and belong to no pass: guber_front_latencies times a generation's way)
        std::vector<const guber_engine::Span*> sp;
        for (auto& x : e->spans) if (x.kernel < KT_FR_COUNT || x.kernel > KT_FR_OUT) sp.push_back(&x);
        size_t g0 = 0;
        for (size_t i = 0; i <= sp.size(); ++i) {
            if (i == sp.size() || (i > g0 && first_stage(sp[i]->kernel))) {
                if (i > g0 && first_stage(sp[g0]->kernel)) {
                    float ms = 0.f;
                    const size_t last = i < sp.size() && sp[i]->kernel == KT_EVALPART_MULTI ? i : i - 1;
                    if (hipEventElapsedTime(&ms, sp[g0]->a, sp[last]->b) == hipSuccess) e->group_us.push_back(ms * 1e3f);
                }
                g0 = i;
            }
        }
    }
    for (auto& s : e->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { e->prof_ms[s.kernel] += ms; e->prof_n[s.kernel]++; }
        e->event_pool.push_back(s.a); e->event_pool.push_back(s.b);
    }
    e->spans.clear();
    *n_out = KT_COUNT;
    for (uint32_t k = 0; k < KT_COUNT && k < cap && out; ++k) {
        memset(&out[k], 0, sizeof(out[k]));
        snprintf(out[k].name, sizeof(out[k].name), "%s", kKernelNames[k]);
        out[k].launches = e->prof_n[k]; out[k].total_ms = e->prof_ms[k]; out[k].units = e->prof_units[k];
    }
    if (out) for (int k = 0; k < KT_COUNT; ++k) { e->prof_ms[k] = 0; e->prof_n[k] = 0; e->prof_units[k] = 0; }
    return GUBER_OK;
}

// the process's zone: the host helpers' copy and, on every visible device, the kernels' (guber_table.h g_tz)
extern "C" int guber_set_timezone(const guber_tz_t* tz) {
    // validate first; then every device; the host helpers' copy LAST, and only when every device has the zone — a failure part of the
    // way leaves the devices that were reached in the new zone and says so, the host (and with it guber_gregorian_*) in the old one
    // never ahead of them; the caller's current device is restored on every path
    guber::TzTable t{};
    const int rc = guber_host_build_tz(tz, &t);
    if (rc != GUBER_OK) return fail(rc, "time zone: at most 16 transitions, ascending");
    int ndev = 0, cur = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { guber_host_publish_tz(t); return GUBER_OK; }   // (no device: the helpers still follow the zone)
    (void)hipGetDevice(&cur);
    int failed = -1;
    for (int d = 0; d < ndev && failed < 0; ++d) {
        if (hipSetDevice(d) != hipSuccess) continue;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(guber::g_tz), &t, sizeof(guber::TzTable)) != hipSuccess) failed = d;
    }
    (void)hipSetDevice(cur);
    if (failed >= 0) { (void)hipGetLastError(); return fail(GUBER_E_HIP, "time zone: a device did not take the table (the host helpers keep the zone they had)"); }
    guber_host_publish_tz(t);
    return GUBER_OK;
}

extern "C" int guber_profile_passes(guber_engine_t* e, float* us, uint32_t cap, uint32_t* n_out) {
    if (!e || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    *n_out = (uint32_t)e->group_us.size();
    if (us) {
        for (uint32_t k = 0; k < cap && k < e->group_us.size(); ++k) us[k] = e->group_us[k];
        e->group_us.clear();
    }
    return GUBER_OK;
}

extern "C" const char* guber_last_error(void) { return g_last_error.c_str(); }

#include "guber_global_sync.h"
#include "guber_wire_dev.h"
