// guber_table.h — device-side data layout of the engine (HBM table, batch views, per-batch work arrays) and the helpers
// every kernel shares: key cell compare / store, probing, workgroup reductions, GLOBAL queueing.
// The layout is described at the top of guber_kernels.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "guber_algo.h"

namespace guber {


constexpr int TILE = 1024;              // requests per workgroup in resolve / scatter (16 waves)
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int MAX_PASSES = 3;           // dense ids < 2^24

// the process's time zone for DURATION_IS_GREGORIAN requests whose calendar values the host did not precompute (guber_set_timezone;
// all zero = UTC)
__device__ TzTable g_tz;
__device__ __forceinline__ const TzTable* guber_tz() { return (g_tz.n || g_tz.offset0_s) ? &g_tz : nullptr; }

struct DirEntry { unsigned long long tag; unsigned long long meta; };
struct alignas(64) KeyCell { uint64_t w[8]; };
struct alignas(128) Bucket { KeyCell cell; Rec rec; };
constexpr uint32_t INLINE_KEY = 62;
constexpr unsigned long long META_READY = 1ull << 63;

struct DevCounters {
    unsigned long long over, hits, misses, evictions;   // over/hits/misses: see BlockCounters
    long long size;
    unsigned long long tags_used, arena_head, retries;
    unsigned int gdirty_n, gdirty_overflow;
};
// Event counters are accumulated per workgroup slot (plain read-modify-write by one thread; launches
// on one stream are ordered) instead of hammering three global words with atomics; readers sum them.
struct BlockCounters { unsigned long long over, hits, misses; long long size_delta; };

// Pending GLOBAL work of one bucket (the reference's globalManager queues, global.go:74-140, kept per
// bucket instead of in host maps): on a non-owner the hits of the interval are summed and the FIRST
// queued request is the template (global.go:100-111); on the owner the LAST request is the template of
// the broadcast (global.go:200).
struct alignas(64) GPend {
    int64_t hits;          // non-owner: summed Hits of the interval
    int64_t limit, duration, burst, created_at;
    uint32_t behavior;
    uint8_t algorithm;
    uint8_t queued;        // 0 = nothing pending, 1 = hits for the owner, 2 = owner update to broadcast
    uint16_t pad;
    uint64_t pad2[2];
};
static_assert(sizeof(GPend) == 64, "one pending record per 64-byte sector");

struct Table {
    GPend* gpend;          // null unless the engine was created with GUBER_FLAG_GLOBAL
    uint32_t* gdirty;      // slots with a pending record
    uint32_t gdirty_cap;
    DirEntry* dir; Bucket* buckets; uint8_t* arena;
    uint64_t mask; uint64_t arena_cap; DevCounters* ctr; BlockCounters* bctr;
    uint32_t max_probe; uint32_t max_key;
    uint64_t hash_mask;   // ~0; tests narrow it to force 64-bit-hash collisions through the verify / retry path
};

struct BatchView {
    uint32_t n;
    uint32_t n_cap;   // engine max_batch: stride of the per-batch double-buffered work arrays
    const uint8_t* key_bytes; const uint32_t* key_off;
    const int64_t *hits, *limit, *duration, *burst, *created_at;
    const uint8_t* algorithm; const uint32_t* behavior; const uint8_t* is_owner;
    const int64_t *greg_expire, *greg_duration;
    int64_t now_ms;
    // keys as rows of a matrix instead of a packed buffer (the GLOBAL exchange evaluates received rows in place):
    // key i = key_bytes + i * key_stride, key_len[i] bytes.  0 = packed, key_off[] applies — with key_len[] as well when the
    // requests are not in the order of their keys (a stage routed to several engines: guber_stage_submit_routed).
    uint32_t key_stride; const uint32_t* key_len;
};
struct ResultView { uint8_t* status; int64_t *limit, *remaining, *reset_time; uint8_t* err; };

// Per-batch record of one segment (= one key of the batch) in the two-launch pipeline, indexed by the request index of the
// key's first toucher ("claimer"): the bucket as it was before the batch, where it lives, how many requests the claimer's
// own (key, tile) group has, and the segment's flags — everything k_eval2 needs for a key that only one tile touches, in
// ONE 64-byte sector written by ONE thread.  `flags` is the only word other threads write (CAS, tagged with the 16-bit
// claim epoch so that it never has to be cleared); the claimer stores the 56 bytes before it and leaves it alone.
// CacheItem.InvalidAt does not fit: a bucket that has one (only Store / Loader items do) sets SM_HAS_INVALID and the
// value travels in a side array.
struct alignas(64) SegRec {
    int64_t limit, duration, remaining, stamp, burst, expire_at;
    uint32_t smeta;                  // kind (2 bits) | status << 2 | SM_HAS_INVALID | algorithm << 8 | (claimer group size - 1) << 16
    uint32_t slot;
    unsigned long long flags;        // epoch16 << 48 | SEG_* bits (error code in bits 8..15)
};
static_assert(sizeof(SegRec) == 64, "one segment record = one 64-byte sector");
enum : uint32_t { SM_HAS_INVALID = 8 };
GB_HD uint32_t pack_smeta(const Rec& s, uint32_t group_size) {
    return (rec_kind(s) & 3u) | ((rec_status(s) & 1u) << 2) | (s.invalid_at != 0 ? SM_HAS_INVALID : 0u) | ((rec_algo(s) & 0xffu) << 8) |
           (((group_size - 1u) & 0xffu) << 16);
}
GB_HD uint32_t smeta_group(uint32_t m) { return ((m >> 16) & 0xffu) + 1u; }
GB_HD uint32_t smeta_meta(uint32_t m) { return make_meta(m & 3u, (m >> 2) & 1u, (m >> 8) & 0xffu); }

// request flags written by k_resolve
enum : uint8_t { RF_INSERTED = 1, RF_NEED_VERIFY = 2, RF_ERR = 4 };
// segment flags
enum : uint32_t { SEG_NONUNIFORM = 1, SEG_RETRY = 2, SEG_ERR = 4 /* code in bits 8..15 */, SEG_CREATED_DIFFERS = 8 };

#ifdef GUBER_PHASE_TIMING   // measurement build only (make timing): per-workgroup phase timestamps
#define GB_STAMP(k) do { if (threadIdx.x == 0) W.dbg[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define GB_STAMP2(k) do { if (threadIdx.x == 0) W.dbg[2048 + blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define GB_STAMPW(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); GB_STAMP(k); } while (0)
// the owner-partitioned pipeline's kernels: kern 0 = k_part, 1 = k_own, 2 = k_eval3
#define GP_STAMP(kern, k) do { if (threadIdx.x == 0) W.dbg[4096 + (kern) * 2048 + blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define GP_STAMPW(kern, k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); GP_STAMP(kern, k); } while (0)
#else
#define GP_STAMP(kern, k) do {} while (0)
#define GP_STAMPW(kern, k) do {} while (0)
#define GB_STAMP(k) do {} while (0)
#define GB_STAMP2(k) do {} while (0)
#define GB_STAMPW(k) do {} while (0)
#endif
struct GMsg; struct GRec; struct GRecS;
// the launch's dynamic LDS as a byte array (a macro so that the host emulation of the kernel source, tests/hostsim/fakehip, can give it storage)
#ifndef GUBER_DYN_LDS
#define GUBER_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

struct Work {
    // Store side channel (guber_eval_batch_store): per request, which Store callbacks the reference would issue
    // (EV_ONCHANGE | EV_REMOVE >> 3) and the bucket right after the request.  null = not requested.
    uint8_t* store_flags; Rec* store_after;
    // per-batch segment claims of the two-launch pipeline: an insert-only hash table key hash -> first toucher, 4 x fast_cap
    // cells of (epoch16 << 48 | fingerprint32 << 16 | request index) of which a batch uses 4 per request (cmask), small enough to live in L2 / Infinity Cache, so that
    // the HBM table is not written by k_front at all in steady state (a claim in the directory's meta word would dirty
    // one directory sector per distinct key and batch) and the claim does not wait for the directory lookup.
    unsigned long long* claims; uint32_t cmask; uint32_t epoch16;
#ifdef GUBER_PHASE_TIMING
    unsigned long long* dbg;
#endif
    uint32_t *slot, *did; uint8_t* rflags;   // two-launch pipeline: did[g] = segment id << 16 | head thread << 8 | rank in (segment, tile) group
    uint32_t *keyA, *valA, *keyB, *valB;
    uint32_t *pos, *order, *sdid;
    uint32_t *seg_first, *seg_last, *seg_flags, *seg_rep, *seg_slot;
    Rec* snap;
    uint32_t* hist;        // [MAX_PASSES][tiles][RADIX], raw per-tile digit counts
    uint32_t tiles;        // tiles of this batch
    uint32_t epoch;        // 1 .. 2^31-1
    uint64_t touch;        // the recency stamp of this batch's request 0 (request i carries touch + i: rec_set_stamp by the request that ends a key's run)
    // tile-bitmap grouping (batches of <= FT_MAX_TILES tiles of FT requests): per segment a bitmap of the tiles holding its
    // requests and, per (segment, tile), the group's size and start inside the tile's sorted order.
    // seg_tilemask is double-buffered by batch parity: a batch's eval kernel clears, in the other copy, the words the
    // previous batch's publishers added to, so no memset launch is needed.
    unsigned long long* seg_tilemask;   // [2][cap][FT_WORDS]: (members in these 32 tiles) << 32 | tile bitmap
    uint32_t* did_prev;                 // segment ids of the previous batch (which entries of the other copy to clear)
    uint16_t* tilerow;                  // [cap][FT_MAX_TILES]: members per (segment, tile) — written only for keys that span several tiles of a word
    // request columns copied to HBM by k_front when the batch lives in host memory (zero-copy path): k_eval2 reads the copy,
    // so every request field crosses PCIe once.  null = k_eval2 reads the batch's own arrays.
    int64_t *st_hits, *st_limit, *st_duration, *st_burst, *st_created; uint32_t* st_behavior; uint8_t *st_algorithm, *st_owner;
    SegRec* srec;                       // [cap] segment records of the two-launch pipeline
    int64_t* sinv;                      // [cap] CacheItem.InvalidAt of the segments whose record says SM_HAS_INVALID
    uint32_t careful;                   // 1 = retry round: verify the key before claiming (no speculation)
    uint32_t parity;                    // batch & 1
    uint32_t clear_n;                   // entries of the other copy dirtied by the previous batch
    // counter read-back riding on this batch's k_front (0 = none): its first workgroup copies the engine's counters, as they
    // are before the batch, to device-visible host memory and stamps snap_seq when done (guber_engine.hip maintain())
    uint32_t snap_seq, snap_n;
    DevCounters* snap_c; BlockCounters* snap_b; uint32_t* snap_stamp;
    // owner-partitioned pipeline (guber_kernels_part.h): per (key, tile) group one message from the tile to the key's owner
    // workgroup and one record back; per (tile, owner) where the tile's messages for that owner start and how many there are
    GMsg* gmsg;                         // [cap] tile t's messages: gmsg[t * FT ..], sorted by owner
    uint32_t* gse;                      // [FT_MAX_TILES][PT_PARTS] start | count << 16
    GRec* grec;                         // [cap] the owner's answer to message i: bucket before the batch, slot, flags, rank base, total
    unsigned long long* segtiles;       // [cap][4] tiles holding a segment whose requests are walked serially (all zero between batches)
    uint32_t pshift;                    // owner of a key (256 owners) = its home position >> pshift
    uint32_t pmslot;                    // 0: the batch's owner count is pmode[0]; 1 | 2: pmode[3 + pmslot] (GUBER_FUSE_EP: guber_kernels_part.h pm_bits_of)
    uint32_t* pmode;                    // device words: [0..3] owner bits of the next batch (7 | 8), batches left at 8, rounds that split in this batch, pinned; [4..5] the bits per batch parity (pmslot)
    // 32-byte records in grs[] (guber_kernels_part.h), the 64-byte form (grec[]) only for
    // the groups whose record does not fit
    GRecS* grs;                         // [cap]
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// raise SEG_* bits on a segment record (rare: errors, hash collisions, requests of one key that differ)
__device__ __forceinline__ void seg_raise(SegRec* r, uint32_t e16, uint32_t bits) {
    unsigned long long* p = &r->flags;
    unsigned long long cur = ld_agent(p);
    for (;;) {
        const unsigned long long base = (uint32_t)(cur >> 48) == e16 ? cur : ((unsigned long long)e16 << 48);
        const unsigned long long want = base | bits;
        if (want == cur) return;
        const unsigned long long old = atomicCAS(p, cur, want);
        if (old == cur) return;
        cur = old;
    }
}
__device__ __forceinline__ uint32_t seg_flags_of(unsigned long long w, uint32_t e16) {
    return (uint32_t)(w >> 48) == e16 ? (uint32_t)w : 0u;
}
__device__ __forceinline__ uint64_t ld_key_word(const uint8_t* p) {
    uint64_t v; __builtin_memcpy(&v, p, 8); return v;
}
__device__ __forceinline__ uint64_t tail_mask(uint32_t nbytes) {  // nbytes in 1..8
    return nbytes >= 8 ? ~0ull : ((1ull << (8 * nbytes)) - 1ull);
}

__device__ __forceinline__ uint32_t key_off_of(const BatchView& B, uint32_t i) { return B.key_stride ? i * B.key_stride : B.key_off[i]; }
__device__ __forceinline__ uint32_t key_len_of(const BatchView& B, uint32_t i, uint32_t off) {
    return (B.key_stride || B.key_len) ? B.key_len[i] : B.key_off[i + 1] - off;
}
__device__ __forceinline__ Req load_req(const BatchView& B, uint32_t i) {
    Req r;
    r.hits = B.hits[i]; r.limit = B.limit[i]; r.duration = B.duration[i];
    r.burst = B.burst ? B.burst[i] : 0;
    r.created_at = B.created_at ? B.created_at[i] : B.now_ms;
    r.greg_expire = B.greg_expire ? B.greg_expire[i] : 0;
    r.greg_duration = B.greg_duration ? B.greg_duration[i] : 0;
    r.behavior = B.behavior ? B.behavior[i] : 0;
    r.algorithm = B.algorithm ? B.algorithm[i] : 0;
    r.is_owner = B.is_owner ? B.is_owner[i] : 1;
    // DURATION_IS_GREGORIAN without host-precomputed values: the calendar interval of the batch clock (interval.go:84-148, UTC)
    if ((r.behavior & BH_GREGORIAN) && !(B.greg_expire && B.greg_duration)) greg_fill(B.now_ms, r.duration, r.greg_expire, r.greg_duration, guber_tz());
    return r;
}
// the request without the calendar values (only the general path reads them: the closed forms decline GREGORIAN requests)
__device__ __forceinline__ Req load_req_nogreg(const BatchView& B, uint32_t i) {
    Req r;
    r.hits = B.hits[i]; r.limit = B.limit[i]; r.duration = B.duration[i];
    r.burst = B.burst ? B.burst[i] : 0;
    r.created_at = B.created_at ? B.created_at[i] : B.now_ms;
    r.greg_expire = 0; r.greg_duration = 0;
    r.behavior = B.behavior ? B.behavior[i] : 0;
    r.algorithm = B.algorithm ? B.algorithm[i] : 0;
    r.is_owner = B.is_owner ? B.is_owner[i] : 1;
    return r;
}
__device__ __forceinline__ void store_resp(const ResultView& R, uint32_t i, const Resp& o) {
    R.status[i] = o.status; R.limit[i] = o.limit; R.remaining[i] = o.remaining;
    R.reset_time[i] = o.reset_time; R.err[i] = o.err;
}
__device__ __forceinline__ void store_events(const Work& W, uint32_t i, uint32_t ev, const Rec& after) {
    if (!W.store_flags) return;
    W.store_flags[i] = (uint8_t)((ev >> 3) & 3u);
    if (ev & EV_ONCHANGE) W.store_after[i] = after;
}
__device__ __forceinline__ void store_err(const ResultView& R, uint32_t i, uint8_t code) {
    R.status[i] = 0; R.limit[i] = 0; R.remaining[i] = 0; R.reset_time[i] = 0; R.err[i] = code;
}

// exact key comparison against the key stored for `slot`
__device__ __forceinline__ bool key_equal(const Table& T, uint64_t slot, const uint8_t* key, uint32_t len) {
    const KeyCell* c = &T.buckets[slot].cell;
    uint64_t w7 = c->w[7];
    if ((uint32_t)(w7 >> 48) != len) return false;
    const uint8_t* stored = nullptr;
    if (len > INLINE_KEY) stored = T.arena + c->w[0];
    uint32_t nw = (len + 7) >> 3;
    for (uint32_t w = 0; w < nw; ++w) {
        uint64_t kv = ld_key_word(key + 8 * w);
        uint64_t cv;
        if (stored) cv = ld_key_word(stored + 8 * w);   // arena allocations are 8-byte padded
        else { cv = c->w[w]; if (w == 7) cv &= 0x0000ffffffffffffull; }
        if (w == nw - 1) { uint64_t m = tail_mask(len - 8 * w); kv &= m; cv &= m; }
        if (kv != cv) return false;
    }
    return true;
}

// store the key of a freshly claimed slot; false = key arena exhausted
__device__ __forceinline__ bool key_store(const Table& T, uint64_t slot, const uint8_t* key, uint32_t len) {
    KeyCell* c = &T.buckets[slot].cell;
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = 0;
    if (len <= INLINE_KEY) {
        uint32_t nw = (len + 7) >> 3;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            if (i < nw) {
                uint64_t kv = ld_key_word(key + 8 * i);
                if (i == nw - 1) kv &= tail_mask(len - 8 * i);
                w[i] = kv;
            }
        }
    } else {
        uint64_t need = ((uint64_t)len + 7) & ~7ull;
        uint64_t off = atomicAdd(&T.ctr->arena_head, (unsigned long long)need);
        if (off + need > T.arena_cap) {   // poison the cell: length 0xFFFF never equals a legal key length
#pragma unroll
            for (int i = 0; i < 7; ++i) c->w[i] = 0;
            c->w[7] = 0xffffull << 48;
            return false;
        }
        for (uint64_t b = 0; b < need; b += 8) {
            uint64_t kv = ld_key_word(key + b);
            if (b + 8 > len) kv &= tail_mask(len - (uint32_t)b);
            *(uint64_t*)(T.arena + off + b) = kv;
        }
        w[0] = off;
    }
    w[7] = (w[7] & 0x0000ffffffffffffull) | ((uint64_t)len << 48);
#pragma unroll
    for (int i = 0; i < 8; ++i) c->w[i] = w[i];
    return true;
}

enum : uint32_t { PR_FOUND = 1, PR_INSERTED = 2, PR_NEED_VERIFY = 4, PR_FULL = 8, PR_MISSING = 16 };

// Find the directory entry of `key`, inserting it when absent (insert = true).
//  - tags are write-once, so a stale "empty" read is resolved by the CAS;
//  - an entry without META_READY was inserted during THIS launch by another thread whose key bytes
//    may not be visible yet: the match is tentative (PR_NEED_VERIFY) and checked in the next launch.
// The result packs the PR_* flags (low 32 bits) and the slot (high 32 bits): a by-reference out-parameter was
// observed to come back as 0 from the inlined function with this compiler (ROCm 7.0.2 hipcc, gfx950) after an
// unrelated layout change, so the slot travels in the return value.
__device__ __forceinline__ uint64_t probe_packed(const Table& T, const uint8_t* key, uint32_t len, uint64_t h, bool insert) {
    h &= T.hash_mask;
    unsigned long long tag = h ? h : 1ull;
    uint64_t pos = (h >> 7) & T.mask;
    for (uint32_t step = 0; step < T.max_probe; ++step, pos = (pos + 1) & T.mask) {
        unsigned long long t = ld_agent(&T.dir[pos].tag);
        if (t == 0ull) {
            if (!insert) return PR_MISSING;
            unsigned long long old = atomicCAS(&T.dir[pos].tag, 0ull, tag);
            if (old == 0ull) {
                if (!key_store(T, pos, key, len)) return (pos << 32) | PR_FULL | PR_INSERTED;
                return (pos << 32) | PR_INSERTED;
            }
            t = old;
        }
        if (t == tag) {
            unsigned long long m = ld_agent(&T.dir[pos].meta);
            if (m & META_READY) {
                if (key_equal(T, pos, key, len)) return (pos << 32) | PR_FOUND;
            } else {
                return (pos << 32) | PR_NEED_VERIFY;
            }
        }
    }
    return PR_FULL;
}
__device__ __forceinline__ uint32_t probe(const Table& T, const uint8_t* key, uint32_t len, uint64_t h, bool insert, uint32_t& slot_out) {
    const uint64_t r = probe_packed(T, key, len, h, insert);
    slot_out = (uint32_t)(r >> 32);
    return (uint32_t)r;
}

// Inclusive prefix sum over the 64 lanes of a wave, ALL of them active: the DPP forms read neighbouring lanes' registers whether those
// lanes are enabled or not, so a partial EXEC mask would drop partial sums silently and readlane(63) would return a stale value.  Every
// caller is at a workgroup-uniform point of a kernel whose workgroup is a multiple of 64 threads (block_sum / block_sum_lds, the scans of
// k_part / k_own / k_eval3 / k_fr_scan / the wire decoder): keep it that way — a divergent caller must take its own shuffle loop.
// row_bcast exists on gfx9 (gfx90a / gfx94x / gfx950) only; this library is built for gfx950 alone.
// Inclusive prefix sum over the 64 lanes of a wave, ALL of them active.  On the device: six DPP adds (row_shr 1 / 2 / 4 / 8 inside the
// rows of 16 lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 — the gfx9 scan) instead of six
// ds_bpermute round trips with their index arithmetic (round 5's instruction diet: a scan was ~36 instructions, now 6 + hazards).
// The host build of the kernel source (tests/hostsim/fakehip) takes the shuffle form.
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 (lane 15 of the row before -> rows 1, 3)
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 (lane 31 -> rows 2, 3)
    return v;
#else
    const int lane = (int)(threadIdx.x & 63);
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, (unsigned)o, 64); if (lane >= o) v += t; }
    return v;
#endif
}
__device__ __forceinline__ int wave_sum(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readlane(wave_incl_scan_i32(v), 63);
#else
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#endif
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global
// load, store and atomic of the wave (s_waitcnt vmcnt(0)); k_front / k_eval2 exchange data between threads
// through LDS only, so their global traffic may stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ int block_sum_lds(int v, int* red) {   // block_sum with LDS-only barriers
    v = wave_sum(v);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    lds_barrier();
    if (lane == 0) red[wave] = v;
    lds_barrier();
    int t = 0;
    if (threadIdx.x == 0) for (uint32_t w = 0; w < nw; ++w) t += red[w];
    return t;
}
// Sum v over the workgroup (<= 16 waves); result valid in thread 0.  `red` = 16 ints of LDS.
__device__ __forceinline__ int block_sum(int v, int* red) {
    v = wave_sum(v);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    int t = 0;
    if (threadIdx.x == 0) for (uint32_t w = 0; w < nw; ++w) t += red[w];
    return t;
}
// Queue GLOBAL work for a bucket after a segment of n successful identical requests `r` (or, from the
// serial walk, one request at a time with n = 1).  Called by ONE thread per bucket per batch.
//   non-owner request (V1Instance.getGlobalRateLimit -> QueueHit, gubernator.go:395-421, global.go:74-78)
//   owner request     (getLocalRateLimit -> QueueUpdate, gubernator.go:604-606, global.go:80-84)
__device__ __forceinline__ void queue_global(const Table& T, uint32_t slot, const Req& r, uint64_t n) {
    if (!T.gpend || !(r.behavior & BH_GLOBAL) || r.hits == 0 || n == 0) return;
    GPend p = T.gpend[slot];
    const bool was_queued = p.queued != 0;
    if (r.is_owner) {
        p.queued = 2; p.hits = 0;
        p.limit = r.limit; p.duration = r.duration; p.burst = r.burst; p.created_at = r.created_at;
        p.behavior = r.behavior; p.algorithm = r.algorithm;                      // last request wins
    } else if (p.queued == 1) {
        p.hits = wadd(p.hits, wmul(r.hits, (int64_t)n));                        // hits[key].Hits += r.Hits
        p.behavior |= (r.behavior & BH_RESET_REMAINING);                        // global.go:105-107
    } else {
        p.queued = 1; p.hits = wmul(r.hits, (int64_t)n);
        p.limit = r.limit; p.duration = r.duration; p.burst = r.burst; p.created_at = r.created_at;
        p.behavior = r.behavior; p.algorithm = r.algorithm;                      // first request is the template
    }
    T.gpend[slot] = p;
    if (!was_queued) {
        const uint32_t k = atomicAdd(&T.ctr->gdirty_n, 1u);
        if (k < T.gdirty_cap) T.gdirty[k] = slot; else atomicAdd(&T.ctr->gdirty_overflow, 1u);
    }
}

}  // namespace guber
