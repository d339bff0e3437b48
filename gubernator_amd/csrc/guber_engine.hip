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
#include "guber_test_flags.h"
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
    e->zero_copy = guber_lab_env("GUBER_NO_ZEROCOPY") == nullptr;
    e->fuse = guber_lab_env("GUBER_NO_FUSE") == nullptr;
    e->stage_dma = guber_lab_env("GUBER_NO_STAGE_DMA") == nullptr;
    if (const char* v = guber_lab_env("GUBER_STAGE_COPY_MIN")) e->stage_copy_min = (uint32_t)atoi(v);
    e->cap256 = (e->fast_cap + FT - 1) / FT * FT;
    rc |= e->w_tilemask.ensure((size_t)2 * e->fast_cap * FT_WORDS); rc |= e->w_srec.ensure(e->fast_cap); rc |= e->w_sinv.ensure(e->cap256);
    rc |= e->w_tilerow.ensure((size_t)e->cap256 * FT_MAX_TILES);
    e->force_part = (cfg->flags & GUBER_FLAG_TEST_FORCE_PART) != 0;
    e->use_part = !(cfg->flags & GUBER_FLAG_NO_PART) && !e->force_radix && !e->always_careful;
    // GUBER_PIPELINE: "claims" = never the owner-partitioned pipeline, "part" = also for a batch launched on its own; default: the
    // owner-partitioned pipeline when several tables share the launches (where it is faster: profiles/r04_*), claims otherwise
    if (const char* v = guber_lab_env("GUBER_PIPELINE")) { if (!strcmp(v, "claims")) e->use_part = false; else if (!strcmp(v, "part")) e->part_single = true; }
    if (const char* v = guber_lab_env("GUBER_PART_MIN")) e->part_min = (uint32_t)std::max(257, atoi(v));
    // Inside one guber_eval_batches_routed_dev call a group's k_eval3 shares a launch with the k_part of the same tables' next batches
    // (k_evalpart_multi, guber_kernels_part.h: two launches per pass instead of three; measured in round 5 on the headline, same box,
    // alternating: 8.69 -> 8.99 G decisions/s, profiles/r05_a_fuse_ep_ab.txt).  Another thread's call on one of the group's engines
    // launches the held-back k_eval3 first (guber_engine::held).  GUBER_FUSE_EP=0 is the switch for an A/B on another box.
    if (const char* v = guber_lab_env("GUBER_FUSE_EP")) e->fuse_ep = atoi(v) != 0;
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
    if (const char* v = guber_lab_env("GUBER_PT_BITS")) { const int b = atoi(v); if (b == 7 || b == 8) { pm0[0] = pm0[4] = pm0[5] = (uint32_t)b; pm0[3] = 1u; } }
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
    if (guber_lab_env("GUBER_ENGINE_STATS"))
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

// The rest of the engine, in the order it is compiled (one translation unit: the parts share the file-local helpers above):
#include "engine_batch.inl"      // the bounded cache's pre-pass, a batch's prelude / plans / launches, guber_eval_batch[es]_dev
#include "engine_dispatch.inl"   // one dispatcher for several engines: fused launches, GUBER_FUSE_EP, guber_eval_batches_routed_dev
#include "guber_front.h"         // guber_front_*: a stream in arrival order routed on the device, answered in arrival order
#include "engine_host.inl"       // host-pointer evaluation, the one-launch small path
#include "engine_stages.inl"     // stages (end-to-end path), groups of stages, the routed stage of a pool
#include "engine_items.inl"      // guber_eval_batch, cache operations, dump, ring router, GLOBAL rows
#include "engine_maint.inl"      // compaction, eviction, counters, per-kernel timing, time zone
