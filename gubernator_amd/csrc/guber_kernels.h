// guber_kernels.h — device-side data layout and the gfx950 kernels of the batched rate-limit path.
//
// HBM layout (one engine = one GPU = one shard of the key space):
//   dir[slots]      16 B  {tag = XXH64(key) (0 = empty), meta = READY | epoch | segment id}  probe target
//   buckets[slots] 128 B  key cell (key bytes <= 62 inline, longer keys in the arena, u16 length) +
//                         guber::Rec (CacheItem + Token/LeakyBucketItem): two adjacent 64-byte sectors,
//                         fetched together in one round trip
// `slots` is a power of two >= 2 x cache_size; linear probing; tags are write-once (a removed bucket
// keeps its tag and key, its record becomes K_ABSENT) so probing never needs tombstone handling.
//
// One batch (n requests, any number of duplicates of a key, reference order semantics
// gubernator.go:203 + workers.go:190-258) is evaluated by a fixed sequence of launches on one stream.
//
// n <= 65 536 (the headline configuration) — TWO launches, described in detail above k_front:
//   k_front        one workgroup per tile of 256 requests: hash, find-or-insert the directory entry (home bucket
//                  fetched speculatively in the same round trip), one claim CAS per (workgroup, key) after an LDS
//                  leader election, key verification, bucket snapshot, grouping of the tile by segment id
//                  through an LDS hash table, one packed atomic per (key, tile) group
//   k_eval2        rank of every request inside its key's segment from the tile bitmap + per-tile counts, then
//                  every request computes ITS OWN response from (snapshot, rank) with guber::eval_uniform_rank —
//                  no atomics on bucket state, no serial chain for hot keys; the last request of a segment
//                  writes the bucket back.  Heterogeneous segments are walked in request order by one thread.
//
// n > 65 536 — a global stable radix sort of the requests by segment id:
//   k_resolve      hash each key, find-or-insert its directory entry, give every distinct key of
//                  the batch a segment id (= request index of the first toucher, claimed with one
//                  64-bit CAS on the entry's meta word), per-tile digit histogram for the sort
//   k_scatter x P  stable LSD radix passes on the segment id (P = ceil(log256 n)); every workgroup
//                  derives its digit bases from the raw per-tile histograms itself (no scan launch);
//                  the first pass also verifies in-batch inserts, publishes READY, snapshots each
//                  touched bucket into a dense array and flags segments whose requests differ
//   k_hist         per-tile digit histogram of the next pass (only between passes)
//   k_heads        segment (= same key) boundaries in the sorted order
//   k_eval         as k_eval2, with the rank taken from the sorted position
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "guber_algo.h"

namespace guber {

constexpr int TILE = 1024;              // requests per workgroup in resolve / scatter (16 waves)
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int MAX_PASSES = 3;           // dense ids < 2^24

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
};
struct ResultView { uint8_t* status; int64_t *limit, *remaining, *reset_time; uint8_t* err; };

// request flags written by k_resolve
enum : uint8_t { RF_INSERTED = 1, RF_NEED_VERIFY = 2, RF_ERR = 4 };
// segment flags
enum : uint32_t { SEG_NONUNIFORM = 1, SEG_RETRY = 2, SEG_ERR = 4 /* code in bits 8..15 */, SEG_CREATED_DIFFERS = 8 };

#ifdef GUBER_PHASE_TIMING   // measurement build only (make timing): per-workgroup phase timestamps
#define GB_STAMP(k) do { if (threadIdx.x == 0) W.dbg[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define GB_STAMP2(k) do { if (threadIdx.x == 0) W.dbg[2048 + blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define GB_STAMPW(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); GB_STAMP(k); } while (0)
#else
#define GB_STAMP(k) do {} while (0)
#define GB_STAMP2(k) do {} while (0)
#define GB_STAMPW(k) do {} while (0)
#endif
struct Work {
    // Store side channel (guber_eval_batch_store): per request, which Store callbacks the reference would issue
    // (EV_ONCHANGE | EV_REMOVE >> 3) and the bucket right after the request.  null = not requested.
    uint8_t* store_flags; Rec* store_after;
    // per-batch segment claims of the two-launch pipeline: an insert-only hash table slot -> first toucher, 2 x fast_cap
    // cells of (epoch16 << 48 | slot << 16 | request index), small enough to live in L2 / Infinity Cache, so that the
    // HBM table is not written by k_front at all in steady state (a claim in the directory's meta word would dirty one
    // directory sector per distinct key and batch).  null = claim in the directory entry instead (GUBER_FLAG_DIR_CLAIMS).
    unsigned long long* claims; uint32_t cmask; uint32_t epoch16;
#ifdef GUBER_PHASE_TIMING
    unsigned long long* dbg;
#endif
    uint32_t *slot, *did; uint8_t* rflags;
    uint32_t *keyA, *valA, *keyB, *valB;
    uint32_t *pos, *order, *sdid;
    uint32_t *seg_first, *seg_last, *seg_flags, *seg_rep, *seg_slot;
    Rec* snap;
    uint32_t* hist;        // [MAX_PASSES][tiles][RADIX], raw per-tile digit counts
    uint32_t tiles;        // tiles of this batch
    uint32_t epoch;        // 1 .. 2^31-1
    // tile-bitmap grouping (batches of <= FT_MAX_TILES tiles of FT requests): per segment a bitmap of the tiles holding its
    // requests and, per (segment, tile), the group's size and start inside the tile's sorted order.
    // seg_flags / seg_tilemask are double-buffered by batch parity: a batch's eval kernel clears the
    // other copy for the next batch, so no memset launch is needed.
    unsigned long long* seg_tilemask;   // [2][cap][FT_WORDS]: (members in these 32 tiles) << 32 | tile bitmap
    uint32_t* did_prev;                 // segment ids of the previous batch (which entries of the other copy to clear)
    uint32_t* seg_flags2;               // [2][cap]
    uint32_t* tilerow;                  // [cap][FT_MAX_TILES]: members per (segment, tile) — written only for keys that span several tiles of a word
    uint16_t* lrank;                    // [max_batch] rank of a request inside its (segment, tile) group
    uint32_t careful;                   // 1 = retry round: verify the key before claiming (no speculation)
    uint32_t parity;                    // batch & 1
    uint32_t clear_n;                   // entries of the other copy dirtied by the previous batch
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_key_word(const uint8_t* p) {
    uint64_t v; __builtin_memcpy(&v, p, 8); return v;
}
__device__ __forceinline__ uint64_t tail_mask(uint32_t nbytes) {  // nbytes in 1..8
    return nbytes >= 8 ? ~0ull : ((1ull << (8 * nbytes)) - 1ull);
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

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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

// lanes of this wave that hold the same 8-bit digit as the caller (among `valid` lanes)
__device__ __forceinline__ unsigned long long digit_peers(uint32_t digit, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

// ---------------------------------------------------------------------------------------------
// k_resolve: one thread per request.
__global__ __launch_bounds__(TILE) void k_resolve(Table T, BatchView B, Work W) {
    __shared__ uint32_t lhist[RADIX];
    __shared__ int red[TILE / 64];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    const uint32_t i = tile * TILE + tid;
    const bool valid = i < B.n;
    if (tid < RADIX) lhist[tid] = 0;
    __syncthreads();
    uint32_t d = 0;
    int inserted = 0;
    if (valid) {
        const uint32_t off = B.key_off[i];
        const uint32_t len = B.key_off[i + 1] - off;
        const uint8_t* key = B.key_bytes + off;
        uint32_t slot = 0;
        uint8_t rf = 0;
        uint32_t errcode = 0;
        if (len == 0) errcode = IE_EMPTY_KEY;
        else if (len > T.max_key) errcode = 7;  // GUBER_ITEM_E_KEY_TOO_LONG
        uint32_t pr = 0;
        if (!errcode) {
            uint64_t h = xxhash64(key, len, 0);
            pr = probe(T, key, len, h, true, slot);
            if (pr & PR_FULL) errcode = 6;      // GUBER_ITEM_E_TABLE_FULL
        }
        inserted = (pr & PR_INSERTED) ? 1 : 0;
        if (errcode) {
            d = i;                              // a solo segment that only carries the error
            W.seg_flags[d] = SEG_ERR | (errcode << 8);
            W.seg_rep[d] = i; W.seg_slot[d] = 0;
            rf = RF_ERR | (inserted ? RF_INSERTED : 0);
        } else {
            if (inserted) rf |= RF_INSERTED;
            if (pr & PR_NEED_VERIFY) rf |= RF_NEED_VERIFY;
            // segment id of this key within the batch = request index of the first toucher
            unsigned long long* mp = &T.dir[slot].meta;
            unsigned long long m = ld_agent(mp);
            for (;;) {
                if ((uint32_t)((m >> 32) & 0x7fffffffu) == W.epoch) { d = (uint32_t)m; break; }
                const unsigned long long want = (m & META_READY) | ((unsigned long long)W.epoch << 32) | i;
                const unsigned long long old = atomicCAS(mp, m, want);
                if (old == m) {
                    d = i;
                    W.seg_flags[d] = 0; W.seg_rep[d] = i; W.seg_slot[d] = slot;
                    break;
                }
                m = old;
            }
        }
        W.slot[i] = slot; W.did[i] = d; W.rflags[i] = rf;
    }
    // per-tile histogram of the first digit: one LDS add per distinct digit per wave
    const uint32_t digit = d & (RADIX - 1);
    const unsigned long long peers = digit_peers(digit, valid);
    if (valid && (peers & ((1ull << (tid & 63)) - 1ull)) == 0) atomicAdd(&lhist[digit], (uint32_t)__popcll(peers));
    const int ins = block_sum(inserted, red);   // contains the barriers that publish lhist
    if (tid == 0 && ins) atomicAdd(&T.ctr->tags_used, (unsigned long long)ins);
    if (tid < RADIX) W.hist[(size_t)tile * RADIX + tid] = lhist[tid];
}

// k_hist: per-tile digit histogram of pass `pass` over the keys produced by the previous pass.
__global__ __launch_bounds__(TILE) void k_hist(Work W, uint32_t n, int pass, const uint32_t* kin) {
    __shared__ uint32_t lhist[RADIX];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    const uint32_t g = tile * TILE + tid;
    const bool valid = g < n;
    if (tid < RADIX) lhist[tid] = 0;
    __syncthreads();
    const uint32_t digit = valid ? ((kin[g] >> (RADIX_BITS * pass)) & (RADIX - 1)) : 0;
    const unsigned long long peers = digit_peers(digit, valid);
    if (valid && (peers & ((1ull << (tid & 63)) - 1ull)) == 0) atomicAdd(&lhist[digit], (uint32_t)__popcll(peers));
    __syncthreads();
    if (tid < RADIX) W.hist[((size_t)pass * W.tiles + tile) * RADIX + tid] = lhist[tid];
}

// k_scatter: one stable LSD radix pass (8-bit digit `pass`) over (key = segment id, val = request idx).
__global__ __launch_bounds__(TILE) void k_scatter(Table T, BatchView B, Work W, int pass, int first, int last,
                                                  const uint32_t* kin, const uint32_t* vin, uint32_t* kout,
                                                  uint32_t* vout) {
    __shared__ uint32_t whist[TILE / 64][RADIX];
    __shared__ uint32_t dscan[RADIX];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = tile * TILE + tid;
    const bool valid = g < B.n;
    for (uint32_t j = tid; j < (TILE / 64) * RADIX; j += TILE) (&whist[0][0])[j] = 0;

    // digit bases from the raw per-tile histograms: all elements of smaller digits, plus this digit's
    // elements in earlier tiles.  Thread t < 256 owns digit t; the column reads are coalesced.
    uint32_t before = 0, total = 0;
    if (tid < RADIX) {
        const uint32_t* col = W.hist + (size_t)pass * W.tiles * RADIX + tid;
        for (uint32_t t = 0; t < W.tiles; ++t) {
            const uint32_t v = col[(size_t)t * RADIX];
            total += v;
            if (t < tile) before += v;
        }
        dscan[tid] = total;
    }
    __syncthreads();
    for (uint32_t o = 1; o < RADIX; o <<= 1) {          // inclusive scan of the 256 digit totals
        uint32_t v = 0;
        if (tid < RADIX && tid >= o) v = dscan[tid - o];
        __syncthreads();
        if (tid < RADIX) dscan[tid] += v;
        __syncthreads();
    }
    const uint32_t my_base = tid < RADIX ? dscan[tid] - total + before : 0;

    uint32_t key = 0, val = 0;
    if (valid) { key = first ? W.did[g] : kin[g]; val = first ? g : vin[g]; }
    const uint32_t digit = (key >> (RADIX_BITS * pass)) & (RADIX - 1);

    if (first && valid) {
        // deferred work of the resolve stage, in request order (needs every k_resolve write)
        const uint8_t rf = W.rflags[g];
        const uint32_t d = key;
        if (!(rf & RF_ERR)) {
            const uint32_t slot = W.slot[g];
            if (rf & RF_NEED_VERIFY) {
                const uint32_t off = B.key_off[g];
                if (!key_equal(T, slot, B.key_bytes + off, B.key_off[g + 1] - off)) atomicOr(&W.seg_flags[d], SEG_RETRY);
            }
            if (rf & RF_INSERTED) atomicOr(&T.dir[slot].meta, META_READY);
            if (d == g) {
                W.snap[d] = T.buckets[slot].rec;
            } else {
                Req a = load_req(B, g), b = load_req(B, d);
                if (!req_eq(a, b)) atomicOr(&W.seg_flags[d], req_eq_but_created(a, b) ? SEG_CREATED_DIFFERS : SEG_NONUNIFORM);
            }
        } else if (rf & RF_INSERTED) {
            atomicOr(&T.dir[W.slot[g]].meta, META_READY);
        }
    }

    const unsigned long long peers = digit_peers(digit, valid);
    const uint32_t rank_in_wave = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank_in_wave == 0) whist[wave][digit] = __popcll(peers);
    __syncthreads();
    if (tid < RADIX) {
        uint32_t run = my_base;
#pragma unroll
        for (int w = 0; w < TILE / 64; ++w) { uint32_t c = whist[w][tid]; whist[w][tid] = run; run += c; }
    }
    __syncthreads();
    if (valid) {
        const uint32_t dst = whist[wave][digit] + rank_in_wave;
        if (last) { W.sdid[dst] = key; W.order[dst] = val; W.pos[val] = dst; }
        else { kout[dst] = key; vout[dst] = val; }
    }
}

// k_heads: segment boundaries in sorted order.
__global__ __launch_bounds__(256) void k_heads(Work W, uint32_t n) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t d = W.sdid[p];
    if (p == 0 || W.sdid[p - 1] != d) W.seg_first[d] = p;
    if (p == n - 1 || W.sdid[p + 1] != d) W.seg_last[d] = p;
}

// k_eval: one thread per request, request order (coalesced inputs and outputs).
__global__ __launch_bounds__(256) void k_eval(Table T, BatchView B, ResultView R, Work W) {
    __shared__ int red[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    int c_over = 0, c_hit = 0, c_miss = 0, c_size = 0;
    if (i < B.n) {
        const uint32_t d = W.did[i];
        const uint32_t sf = W.seg_flags[d];
        if (sf & SEG_ERR) {
            store_err(R, i, (uint8_t)(sf >> 8));
        } else if (sf & SEG_RETRY) {
            store_err(R, i, IE_RETRY);
            atomicAdd(&T.ctr->retries, 1ull);
        } else {
            const uint32_t first = W.seg_first[d], last = W.seg_last[d];
            const uint32_t rank = W.pos[i] - first;
            const uint32_t slot = W.seg_slot[d];
            const Req r = load_req(B, i);
            const Rec s0 = W.snap[d];
            // requests differing only in created_at still take the parallel path when created_at is never read
            const bool parallel = !(sf & SEG_NONUNIFORM) &&
                                  (!(sf & SEG_CREATED_DIFFERS) ||
                                   (created_at_irrelevant(s0, r, B.now_ms) && !(T.gpend && (r.behavior & BH_GLOBAL))));
            if (parallel) {
                Rec after; Resp out;
                const uint32_t ev = eval_uniform_rank(s0, r, B.now_ms, rank, out, after);
                store_resp(R, i, out);
                store_events(W, i, ev, after);
                c_over = (ev & EV_OVER) ? 1 : 0; c_hit = (ev & EV_HIT) ? 1 : 0; c_miss = (ev & EV_MISS) ? 1 : 0;
                if (rank == last - first) {
                    T.buckets[slot].rec = after;
                    c_size = (int)(rec_kind(after) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
                    if (out.err == 0) queue_global(T, slot, r, (uint64_t)rank + 1);
                }
            } else if (rank == 0) {
                // requests to this key differ: apply them one by one in request order
                Rec s = s0;
                for (uint32_t q = first; q <= last; ++q) {
                    const uint32_t j = W.order[q];
                    const Req rj = load_req(B, j);
                    Resp out;
                    const uint32_t ev = apply(s, rj, B.now_ms, out);
                    store_resp(R, j, out);
                    store_events(W, j, ev, s);
                    if (out.err == 0) queue_global(T, slot, rj, 1);
                    c_over += (ev & EV_OVER) ? 1 : 0; c_hit += (ev & EV_HIT) ? 1 : 0; c_miss += (ev & EV_MISS) ? 1 : 0;
                }
                T.buckets[slot].rec = s;
                c_size = (int)(rec_kind(s) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
            }
        }
    }
    const int t_over = block_sum(c_over, red), t_hit = block_sum(c_hit, red), t_miss = block_sum(c_miss, red),
              t_size = block_sum(c_size, red);
    if (threadIdx.x == 0 && (t_over | t_hit | t_miss | t_size)) {
        BlockCounters* bc = &T.bctr[blockIdx.x];
        bc->over += (unsigned long long)t_over; bc->hits += (unsigned long long)t_hit;
        bc->misses += (unsigned long long)t_miss; bc->size_delta += t_size;
    }
}

// ---------------------------------------------------------------------------------------------
// Pipeline for batches of <= FT_MAX_TILES tiles of FT requests (65536 requests): TWO launches.
//
// Measured on MI355X (tools/microbench.hip, profiles/): the random-access part (16 B directory entry, 128 B
// bucket, claim CAS for 65536 requests) takes ~5 us with 256-thread workgroups spread over all 256 CUs and
// ~15 us with 1024-thread workgroups on 64 CUs (per-CU L1 request rate), and atomics on ONE address serialise
// at ~12 ns each.  Hence: 256-request tiles, one claim CAS per (workgroup, key) — the threads of a workgroup
// that hold the same key elect a leader through LDS — plain (L1-cacheable) directory loads, and no per-key
// atomics other than one packed add per (key, tile) group.
//
// k_front (one workgroup = one tile of FT requests):
//   A  resolve: key -> directory entry || home bucket (one round trip) -> claim CAS by the leader -> verify.
//      Segment id of a key = request index of its first toucher.  A match on an entry that is not
//      READY (inserted during this launch) is verified against the CLAIMER's request key (input data)
//      instead of the stored key, whose writer may still be in flight; the inserter performs the same
//      comparison, so every member of a segment provably has the key that ends up stored.
//   B  group the tile by segment id through an LDS hash table with per-wave member bitmaps: a request's sorted
//      position, its rank inside its (segment, tile) group and the group size come from four popcounts.
//   C  group heads publish size / start for (segment, tile) and add (size << 32 | tile bit) to the segment's word.
// k_eval2 (request order): rank = members in earlier tiles (bitmap + per-tile counts) + rank in tile.
constexpr int FT = 256;                 // requests per tile in the two-launch pipeline
constexpr int FT_MAX_TILES = 256;       // bitmap bits per segment
constexpr int FT_WORDS = FT_MAX_TILES / 32;   // per segment: 8 x u64, each = members << 32 | bitmap of 32 tiles

__device__ __forceinline__ bool req_key_equal(const BatchView& B, uint32_t a, uint32_t b) {
    const uint32_t oa = B.key_off[a], ob = B.key_off[b];
    const uint32_t la = B.key_off[a + 1] - oa, lb = B.key_off[b + 1] - ob;
    if (la != lb) return false;
    const uint8_t* pa = B.key_bytes + oa; const uint8_t* pb = B.key_bytes + ob;
    const uint32_t nw = (la + 7) >> 3;
    for (uint32_t w = 0; w < nw; ++w) {
        uint64_t x = ld_key_word(pa + 8 * w), y = ld_key_word(pb + 8 * w);
        if (w == nw - 1) { const uint64_t m = tail_mask(la - 8 * w); x &= m; y &= m; }
        if (x != y) return false;
    }
    return true;
}

// claim the segment id of a directory entry for this batch; m = last known meta value
__device__ __forceinline__ uint32_t claim_segment(unsigned long long* mp, unsigned long long m, uint32_t epoch, uint32_t g,
                                                 bool& claimed) {
    for (;;) {
        if ((uint32_t)((m >> 32) & 0x7fffffffu) == epoch) return (uint32_t)m;
        const unsigned long long want = (m & META_READY) | ((unsigned long long)epoch << 32) | g;
        const unsigned long long old = atomicCAS(mp, m, want);
        if (old == m) { claimed = true; return g; }
        m = old;
    }
}

// claim the segment id of bucket `slot` for this batch in the claims table (see Work::claims).  Insert-only open
// addressing: cells of older epochs count as empty; everybody scans from the same home cell in the same order, so a
// slot is inserted at most once and later arrivers find it.
__device__ __forceinline__ uint32_t claim_cell(uint32_t slot, uint32_t cmask) { return ((slot * 0x9E3779B1u) >> 11) & cmask; }
__device__ __forceinline__ uint32_t claim_slot(unsigned long long* claims, uint32_t cmask, uint32_t slot, uint32_t e16,
                                               uint32_t g, bool& claimed) {
    uint32_t h = claim_cell(slot, cmask);
    const unsigned long long want = ((unsigned long long)e16 << 48) | ((unsigned long long)slot << 16) | g;
    for (;;) {
        // a fresh (L1-bypassing) look first: for a hot key all but the first workgroup find the cell taken and issue no
        // CAS at all — requesting the cell early with the directory entry and CAS-ing on that value was measured and lost
        // (failed CASes on the hot cells; profiles/r01_claims_ab.txt)
        unsigned long long cur = ld_agent(&claims[h]);
        for (;;) {
            if ((uint32_t)(cur >> 48) == e16) {
                if ((uint32_t)(cur >> 16) == slot) return (uint32_t)(cur & 0xffffull);
                break;                                          // another bucket's cell: next
            }
            const unsigned long long old = atomicCAS(&claims[h], cur, want);
            if (old == cur) { claimed = true; return g; }
            cur = old;                                          // lost the race for this cell: look at the winner
        }
        h = (h + 1) & cmask;
    }
}

__global__ __launch_bounds__(FT) void k_front(Table T, BatchView B, Work W) {
    __shared__ uint32_t skey[FT];       // stage 2: candidate slot per thread; phase B: segment id per thread
    __shared__ uint32_t sd[FT];         // stage 2: segment id obtained by each leader
    __shared__ uint32_t ltab[2 * FT];   // stage 2: slot-hash -> some thread holding that slot
    __shared__ int red[FT / 64];
    constexpr int GT_BITS = 9, GT = 1 << GT_BITS;                 // grouping table: 2 x FT entries
    __shared__ uint32_t gkey[GT];
    __shared__ unsigned long long gbits[FT / 64][GT];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    const uint32_t g = tile * FT + tid;
    const bool valid = g < B.n;
    uint32_t* seg_flags = W.seg_flags2 + (size_t)W.parity * B.n_cap;
    unsigned long long* seg_mask = W.seg_tilemask + (size_t)W.parity * B.n_cap * FT_WORDS;

    GB_STAMP(0);
    // ---- phase A, stage 1: find (or insert) the directory entry; start fetching its bucket -----------
    uint32_t d = 0xffffffffu, slot = 0, errcode = 0, len = 0;
    int inserted = 0;
    bool fresh = false, claimed = false, cand = false, ready = false;
    const uint8_t* key = nullptr;
    Rec rec; rec_clear(rec);
    uint4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long meta0 = 0ull;
    if (valid) {
        const uint32_t off = B.key_off[g];
        len = B.key_off[g + 1] - off;
        key = B.key_bytes + off;
        if (len == 0) errcode = IE_EMPTY_KEY;
        else if (len > T.max_key) errcode = 7;
        GB_STAMPW(6);
        if (!errcode && W.careful) {
            // retry round: verify the stored key BEFORE claiming (no speculation, no dedup)
            uint32_t cslot = 0;
            const uint32_t pr = probe(T, key, len, xxhash64(key, len, 0), true, cslot);
            slot = cslot;
            inserted = (pr & PR_INSERTED) ? 1 : 0;
            if (pr & PR_FULL) errcode = 6;
            else {
                fresh = (pr & (PR_INSERTED | PR_NEED_VERIFY)) != 0;
                bool cl = false;
                if (W.claims) d = claim_slot(W.claims, W.cmask, cslot, W.epoch16, g, cl);
                else { unsigned long long* mp = &T.dir[cslot].meta; d = claim_segment(mp, ld_agent(mp), W.epoch, g, cl); }
                claimed = cl;
                rec = T.buckets[cslot].rec;
            }
        } else if (!errcode) {
            const uint64_t h = xxhash64(key, len, 0) & T.hash_mask;
            const unsigned long long tag = h ? h : 1ull;
            uint64_t pos = (h >> 7) & T.mask;
            GB_STAMPW(7);
            // speculation: the key's bucket is at its home position for most resident keys (load <= 0.5), so the
            // home bucket is requested together with the home directory entry — one round trip instead of two.
            // Plain loads: L1 may serve a line that is stale within this launch, which is safe here — a stale
            // "empty" tag is corrected by the insert CAS, a stale meta by the claim CAS (only one leader per
            // workgroup and key claims), READY never changes during k_front — and it keeps the thousands of
            // re-reads of a hot key's entry out of L2 / the memory-side atomics' way.
            const uint32_t home = (uint32_t)pos;
            const ulonglong2 de0 = *(const ulonglong2*)&T.dir[home];
            {
                const Bucket* hb = &T.buckets[home];
                const uint4* cw = (const uint4*)&hb->cell; c0 = cw[0]; c1 = cw[1]; c2 = cw[2]; c3 = cw[3];
                rec = hb->rec;
            }
            for (uint32_t step = 0; step < T.max_probe; ++step, pos = (pos + 1) & T.mask) {
                ulonglong2 de = de0;
                if (step) de = *(const ulonglong2*)&T.dir[pos];
                unsigned long long t = de.x, m = de.y;
                if (t == 0ull) {
                    const unsigned long long old = atomicCAS(&T.dir[pos].tag, 0ull, tag);
                    if (old == 0ull) {                               // new key: this thread inserts it
                        slot = (uint32_t)pos; inserted = 1; fresh = true; cand = true;
                        if (!key_store(T, pos, key, len)) { errcode = 6; cand = false; }
                        break;
                    }
                    t = old;
                    m = ld_agent(&T.dir[pos].meta);
                }
                if (t == tag) { slot = (uint32_t)pos; cand = true; meta0 = m; ready = (m & META_READY) != 0; fresh = !ready; break; }
            }
            if (!cand && !errcode) errcode = 6;                      // probe bound exceeded: table full
            if (cand && slot != home) {
                const Bucket* bk = &T.buckets[slot];
                if (ready) { const uint4* cw = (const uint4*)&bk->cell; c0 = cw[0]; c1 = cw[1]; c2 = cw[2]; c3 = cw[3]; }
                rec = bk->rec;                                       // (zero for a bucket never used)
            }
        }
    }
    // ---- stage 2: one claim per (workgroup, slot).  Threads holding the same slot elect a leader
    // through an LDS table; only leaders touch the entry's meta word (atomics on one address serialise
    // at ~12 ns each, and a hot key shows up thousands of times in a batch).
    GB_STAMP(1);
    const bool fast = cand && !W.careful;
    const uint32_t hidx = (slot * 0x9E3779B1u) >> 23;                // 9 bits
    skey[tid] = fast ? slot : 0xffffffffu;
    if (fast) ltab[hidx] = tid;
    lds_barrier();
    uint32_t lead = tid;
    if (fast) { const uint32_t l = ltab[hidx]; if (skey[l] == slot) lead = l; }
    if (fast && lead == tid) {
        d = W.claims ? claim_slot(W.claims, W.cmask, slot, W.epoch16, g, claimed)
                     : claim_segment(&T.dir[slot].meta, meta0, W.epoch, g, claimed);   // GUBER_FLAG_DIR_CLAIMS
        sd[tid] = d;
    }
    lds_barrier();
    if (fast && lead != tid) d = sd[lead];
    lds_barrier();
    GB_STAMP(2);

    // ---- stage 3: verify, flag, snapshot -----------------------------------------------------------
    if (valid) {
        uint8_t rf = 0;
        if (!errcode && fast && ready) {
            const uint64_t cell[8] = {((uint64_t)c0.y << 32) | c0.x, ((uint64_t)c0.w << 32) | c0.z,
                                      ((uint64_t)c1.y << 32) | c1.x, ((uint64_t)c1.w << 32) | c1.z,
                                      ((uint64_t)c2.y << 32) | c2.x, ((uint64_t)c2.w << 32) | c2.z,
                                      ((uint64_t)c3.y << 32) | c3.x, ((uint64_t)c3.w << 32) | c3.z};
            bool eq = (uint32_t)(cell[7] >> 48) == len;
            if (eq) {
                if (len <= INLINE_KEY) {
                    const uint32_t nw = (len + 7) >> 3;
#pragma unroll
                    for (uint32_t w = 0; w < 8; ++w) {
                        if (w < nw) {
                            uint64_t kv = ld_key_word(key + 8 * w), cv = cell[w];
                            if (w == 7) cv &= 0x0000ffffffffffffull;
                            if (w == nw - 1) { const uint64_t mk = tail_mask(len - 8 * w); kv &= mk; cv &= mk; }
                            eq = eq && (kv == cv);
                        }
                    }
                } else {
                    eq = key_equal(T, slot, key, len);
                }
            }
            // the claim was issued before this comparison (speculation).  A mismatch = a 64-bit hash
            // collision with a resident key: I joined a foreign segment, so everybody in it is answered
            // RETRY and re-run in careful mode.
            if (!eq) atomicOr(&seg_flags[d], SEG_RETRY);
        }
        if (errcode) {
            d = g;
            atomicOr(&seg_flags[d], SEG_ERR | (errcode << 8));
            rf = RF_ERR | (inserted ? RF_INSERTED : 0);
        } else {
            if (inserted) rf |= RF_INSERTED;
            if (claimed) {
                rec.pad = slot;                                // the segment's slot rides in the snapshot's spare word
                W.snap[d] = rec;                               // claimer snapshots the bucket
            } else {
                // entry created during this launch: prove key equality against the claimer's request
                if (fresh && !req_key_equal(B, g, d)) atomicOr(&seg_flags[d], SEG_RETRY);
                const Req a = load_req(B, g), b = load_req(B, d);
                if (!req_eq(a, b)) {
                    // created_at-only differences keep the parallel path when created_at cannot matter: decided in
                    // k_eval2 for token buckets; a leaky request must leak nothing (checked here against the bucket
                    // as it is before the batch; the claimer's own created_at is checked in k_eval2)
                    bool soft = req_eq_but_created(a, b);
                    if (soft && a.algorithm == ALGO_LEAKY) soft = leaky_created_harmless(rec, a, B.now_ms);
                    atomicOr(&seg_flags[d], soft ? SEG_CREATED_DIFFERS : SEG_NONUNIFORM);
                }
            }
        }
        W.did[g] = d; W.rflags[g] = rf;
        if (inserted) W.slot[g] = slot;
    }
    const int ins = block_sum_lds(inserted, red);
    if (tid == 0 && ins) atomicAdd(&T.ctr->tags_used, (unsigned long long)ins);
    GB_STAMP(3);

    // ---- phase B: group the tile's FT segment ids through an LDS hash table ------------------------
    // Every distinct id gets one table entry (open addressing, CAS on the key); each wave ORs its lane
    // into the entry's per-wave 64-bit member bitmap.  A request's rank inside its (segment, tile) group,
    // the group size and the group's first thread then come from four popcounts — O(1) per request
    // instead of comparing against all 256 ids.
    const uint32_t lane = tid & 63, wave = tid >> 6;
    for (uint32_t j = tid; j < GT; j += FT) {
        gkey[j] = 0xffffffffu;
#pragma unroll
        for (int w = 0; w < FT / 64; ++w) gbits[w][j] = 0ull;
    }
    lds_barrier();
    uint32_t gh = 0;
    if (valid) {
        gh = (d * 0x9E3779B1u) >> (32 - GT_BITS);
        for (;;) {
            const uint32_t old = atomicCAS(&gkey[gh], 0xffffffffu, d);
            if (old == 0xffffffffu || old == d) break;
            gh = (gh + 1) & (GT - 1);
        }
        atomicOr(&gbits[wave][gh], 1ull << lane);
    }
    lds_barrier();
    uint32_t eq_before = 0, eq_total = 0, head_tid = tid;
    if (valid) {
        bool found_head = false;
#pragma unroll
        for (uint32_t w = 0; w < FT / 64; ++w) {
            const unsigned long long bw = gbits[w][gh];
            const uint32_t c = __popcll(bw);
            eq_total += c;
            if (w < wave) eq_before += c;
            else if (w == wave) eq_before += __popcll(bw & ((1ull << lane) - 1ull));
            if (!found_head && bw) { head_tid = w * 64 + (uint32_t)__ffsll((unsigned long long)bw) - 1; found_head = true; }
        }
    }
    GB_STAMP(4);
    // ---- phase C: publish groups -------------------------------------------------------------------
    if (valid) {
        W.lrank[g] = (uint16_t)(eq_before | (head_tid << 8));   // rank in group | tid of the group's head
        if (eq_before == 0) {
            // ONE atomic per (segment, tile) group: set the tile's bit and add the group size (bits are set once
            // each, so the add never carries into the count).  Its return value tells whether other tiles of this
            // 32-tile word already hold the key — only then are per-tile counts needed (k_eval2 ranks a request by
            // the members in earlier tiles), so the scattered count is written only for keys spanning several tiles:
            // every arriver but the first writes its own, and the second also writes the first's (= the word's
            // count so far, the first having been alone).
            const unsigned long long old = atomicAdd(&seg_mask[(size_t)d * FT_WORDS + (tile >> 5)],
                                                     ((unsigned long long)eq_total << 32) | (1ull << (tile & 31)));
            const uint32_t ob = (uint32_t)old;
            if (ob) {
                uint32_t* row = W.tilerow + (size_t)d * FT_MAX_TILES;
                row[tile] = eq_total;
                if ((ob & (ob - 1u)) == 0u) row[(tile & ~31u) + (uint32_t)__ffs((int)ob) - 1u] = (uint32_t)(old >> 32);
            }
        }
    }
    GB_STAMP(5);
}

#ifndef GUBER_EVAL2_BLOCKS
#define GUBER_EVAL2_BLOCKS 1
#endif
__global__ __launch_bounds__(256, GUBER_EVAL2_BLOCKS) void k_eval2(Table T, BatchView B, ResultView R, Work W) {
    __shared__ int red[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t* seg_flags = W.seg_flags2 + (size_t)W.parity * B.n_cap;
    unsigned long long* seg_mask = W.seg_tilemask + (size_t)W.parity * B.n_cap * FT_WORDS;
    GB_STAMP2(0);
    {   // clear, for the next batch, the entries of the other copy that the previous batch used
        uint32_t* of = W.seg_flags2 + (size_t)(W.parity ^ 1u) * B.n_cap;
        uint4* om = (uint4*)(W.seg_tilemask + (size_t)(W.parity ^ 1u) * B.n_cap * FT_WORDS);
        const uint4 z = {0, 0, 0, 0};
        for (uint32_t j = i; j < W.clear_n; j += gridDim.x * 256) {
            if (W.did_prev[j] == j) {
                of[j] = 0;
#pragma unroll
                for (int q = 0; q < FT_WORDS / 2; ++q) om[(size_t)j * (FT_WORDS / 2) + q] = z;
            }
        }
    }
    // pre-pass: the head of every (segment, tile) group computes the group's base = members of the
    // segment in earlier tiles, and the segment's total, from the bitmap and the per-tile counts; the
    // other members pick both up from LDS (eval workgroup == tile, FT == 256).
    __shared__ uint32_t sbase[FT], stotal[FT];
    const bool live = i < B.n;
    const uint32_t lr = live ? W.lrank[i] : 0u;
    const uint32_t d = live ? W.did[i] : 0u;
    // everything that depends only on (i, d) is requested now, so that these loads are in flight together
    // with the heads' bitmap loads below instead of after the barrier
    uint32_t sf = 0; uint8_t rf = 0; Req r; Rec s0;
    if (live) { sf = seg_flags[d]; rf = W.rflags[i]; r = load_req(B, i); s0 = W.snap[d]; }
    if (live && (lr & 0xffu) == 0u) {
        const uint32_t t = i / FT;
        const uint4* wp = (const uint4*)(seg_mask + (size_t)d * FT_WORDS);     // 64 bytes, one round trip
        unsigned long long sw[FT_WORDS];
#pragma unroll
        for (int q = 0; q < FT_WORDS / 2; ++q) {
            const uint4 v = wp[q];
            sw[2 * q] = ((unsigned long long)v.y << 32) | v.x; sw[2 * q + 1] = ((unsigned long long)v.w << 32) | v.z;
        }
        const uint32_t mw = t >> 5, mb = t & 31;
        uint32_t base = 0, total = 0, below = 0;
#pragma unroll
        for (uint32_t w = 0; w < FT_WORDS; ++w) {
            const uint32_t c = (uint32_t)(sw[w] >> 32);
            total += c;
            base += w < mw ? c : 0u;
            if (w == mw) below = (uint32_t)sw[w] & ((1u << mb) - 1u);
        }
        if (below) {
            // members in earlier tiles of my own 32-tile word: their per-tile counts, 64 bytes, masked by the bitmap
            const uint4* r4 = (const uint4*)(W.tilerow + (size_t)d * FT_MAX_TILES + mw * 32);
            uint4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = r4[q];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t w4[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) base += ((below >> (q * 4 + e)) & 1u) ? (w4[e] & 0xffffu) : 0u;
            }
        }
        sbase[threadIdx.x] = base; stotal[threadIdx.x] = total;
    }
    GB_STAMP2(1);
    lds_barrier();
    GB_STAMP2(2);
    int c_over = 0, c_hit = 0, c_miss = 0, c_size = 0;
    if (live) {
        if (rf & RF_INSERTED) atomicOr(&T.dir[W.slot[i]].meta, META_READY);   // publish this batch's inserts
        if (sf & SEG_ERR) {
            store_err(R, i, (uint8_t)(sf >> 8));
        } else if (sf & SEG_RETRY) {
            store_err(R, i, IE_RETRY);
            atomicAdd(&T.ctr->retries, 1ull);
        } else {
            const uint32_t base = sbase[lr >> 8], total = stotal[lr >> 8];
            const uint32_t rank = base + (lr & 0xffu);
            const uint32_t slot = s0.pad;
            s0.pad = 0;
            // requests differing only in created_at still take the parallel path when created_at cannot matter: live
            // token bucket (never read), or live leaky bucket where no request of the run leaks (the other members
            // were checked in k_front; the claimer's created_at is checked here, identically by every member)
            bool parallel = !(sf & SEG_NONUNIFORM);
            if (parallel && (sf & SEG_CREATED_DIFFERS)) {
                parallel = !(T.gpend && (r.behavior & BH_GLOBAL));
                if (parallel && r.algorithm == ALGO_LEAKY) {
                    Req rc = r;
                    rc.created_at = B.created_at ? B.created_at[d] : B.now_ms;
                    parallel = leaky_created_harmless(s0, rc, B.now_ms) && leaky_created_harmless(s0, r, B.now_ms);
                } else if (parallel) {
                    parallel = created_at_irrelevant(s0, r, B.now_ms);
                }
            }
            if (parallel) {
                Rec after; Resp out;
                const uint32_t ev = eval_uniform_rank(s0, r, B.now_ms, rank, out, after);
                store_resp(R, i, out);
                store_events(W, i, ev, after);
                c_over = (ev & EV_OVER) ? 1 : 0; c_hit = (ev & EV_HIT) ? 1 : 0; c_miss = (ev & EV_MISS) ? 1 : 0;
                if (rank == total - 1) {
                    T.buckets[slot].rec = after;
                    c_size = (int)(rec_kind(after) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
                    if (out.err == 0) queue_global(T, slot, r, (uint64_t)rank + 1);
                }
            } else if (rank == 0) {
                // requests to this key differ: apply them one by one in request order — tiles in order (bitmap), and
                // inside a tile the requests whose segment id is d, found by scanning the tile's 256 ids
                Rec s = s0;
                for (int w = 0; w < FT_WORDS; ++w) {
                    uint32_t mm = (uint32_t)seg_mask[(size_t)d * FT_WORDS + w];
                    while (mm) {
                        const uint32_t tt = w * 32 + (uint32_t)__ffs((int)mm) - 1;
                        mm &= mm - 1u;
                        const uint4* ids = (const uint4*)(W.did + (size_t)tt * FT);
                        for (uint32_t q4 = 0; q4 < FT / 4 && tt * FT + q4 * 4 < B.n; ++q4) {
                          const uint4 v = ids[q4];
                          const uint32_t four[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                          for (uint32_t e4 = 0; e4 < 4; ++e4) {
                            const uint32_t j = tt * FT + q4 * 4 + e4;
                            if (four[e4] != d || j >= B.n) continue;
                            const Req rj = load_req(B, j);
                            Resp out;
                            const uint32_t ev = apply(s, rj, B.now_ms, out);
                            store_resp(R, j, out);
                            store_events(W, j, ev, s);
                            if (out.err == 0) queue_global(T, slot, rj, 1);
                            c_over += (ev & EV_OVER) ? 1 : 0; c_hit += (ev & EV_HIT) ? 1 : 0; c_miss += (ev & EV_MISS) ? 1 : 0;
                          }
                        }
                    }
                }
                T.buckets[slot].rec = s;
                c_size = (int)(rec_kind(s) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
            }
        }
    }
    GB_STAMP2(3);
    const int t_over = block_sum_lds(c_over, red), t_hit = block_sum_lds(c_hit, red), t_miss = block_sum_lds(c_miss, red),
              t_size = block_sum_lds(c_size, red);
    if (threadIdx.x == 0 && (t_over | t_hit | t_miss | t_size)) {
        BlockCounters* bc = &T.bctr[blockIdx.x];
        bc->over += (unsigned long long)t_over; bc->hits += (unsigned long long)t_hit;
        bc->misses += (unsigned long long)t_miss; bc->size_delta += t_size;
    }
    GB_STAMP2(4);
}

// ---------------------------------------------------------------------------------------------
// maintenance kernels: AddCacheItem / GetCacheItem / Remove / Each
struct ItemIn {   // device image of guber_item_t with the key referenced by offset
    Rec rec; uint32_t key_off, key_len;
};

// phase A: find-or-insert the directory entry (flags as in k_resolve)
__global__ __launch_bounds__(256) void k_items_probe(Table T, const ItemIn* items, const uint8_t* keys, uint32_t n,
                                                     uint32_t* slots, uint8_t* flags) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* key = keys + items[i].key_off;
    const uint32_t len = items[i].key_len;
    uint32_t slot = 0; uint8_t f = 0;
    if (len == 0 || len > T.max_key) f = RF_ERR;
    else {
        uint32_t pr = probe(T, key, len, xxhash64(key, len, 0), true, slot);
        if (pr & PR_FULL) f = RF_ERR;
        if (pr & PR_INSERTED) { f |= RF_INSERTED; atomicAdd(&T.ctr->tags_used, 1ull); }
        if (pr & PR_NEED_VERIFY) f |= RF_NEED_VERIFY;
    }
    slots[i] = slot; flags[i] = f;
}
// phase B: verify tentative matches, publish READY, LRUCache.Add (lrucache.go:88-103): replace the
// value when the key is resident (existed = 1), insert otherwise.  result: 0/1 existed, 0xFF retry, 0xFE error
__global__ __launch_bounds__(256) void k_items_commit(Table T, const ItemIn* items, const uint8_t* keys, uint32_t n,
                                                      const uint32_t* slots, const uint8_t* flags, uint8_t* result) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t f = flags[i];
    const uint32_t slot = slots[i];
    if (f & RF_INSERTED) atomicOr(&T.dir[slot].meta, META_READY);
    if (f & RF_ERR) { result[i] = 0xFE; return; }
    if ((f & RF_NEED_VERIFY) && !key_equal(T, slot, keys + items[i].key_off, items[i].key_len)) { result[i] = 0xFF; return; }
    const bool existed = rec_kind(T.buckets[slot].rec) != K_ABSENT;
    T.buckets[slot].rec = items[i].rec;
    if (!existed) atomicAdd((unsigned long long*)&T.ctr->size, 1ull);
    result[i] = existed ? 1 : 0;
}

// LRUCache.GetItem (lrucache.go:111-128) / Remove (:131-135) for one key. mode 0 = get, 1 = remove
__global__ void k_item_lookup(Table T, const uint8_t* key, uint32_t len, int64_t now, int mode, Rec* out, int* found) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    *found = 0;
    uint32_t slot;
    if (len == 0 || len > T.max_key) return;
    uint32_t pr = probe(T, key, len, xxhash64(key, len, 0), false, slot);
    Rec s; rec_clear(s);
    if (pr & PR_FOUND) s = T.buckets[slot].rec;
    if (rec_kind(s) == K_ABSENT) { if (mode == 0) atomicAdd(&T.ctr->misses, 1ull); return; }
    if (mode == 1 || rec_expired(s, now)) {
        Rec z; rec_clear(z);
        T.buckets[slot].rec = z;
        atomicAdd((unsigned long long*)&T.ctr->size, (unsigned long long)(long long)-1);
        if (mode == 0) atomicAdd(&T.ctr->misses, 1ull);
        return;
    }
    if (mode == 0) atomicAdd(&T.ctr->hits, 1ull);
    *out = s; *found = 1;
}

// Read-only residency test per request key: 1 = absent or expired at `now` (what LRUCache.GetItem would report as
// a miss, lrucache.go:111-128) — the keys a configured Store has to be asked for (algorithms.go:45-51).
__global__ __launch_bounds__(256) void k_probe_missing(Table T, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t n,
                                                       int64_t now, uint8_t* missing) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t off = key_off[i], len = key_off[i + 1] - off;
    uint8_t m = 1;
    if (len != 0 && len <= T.max_key) {
        uint32_t slot = 0;
        const uint32_t pr = probe(T, key_bytes + off, len, xxhash64(key_bytes + off, len, 0), false, slot);
        if (pr & PR_FOUND) {
            const Rec s = T.buckets[slot].rec;
            m = (rec_kind(s) == K_ABSENT || rec_expired(s, now)) ? 1 : 0;
        }
    }
    missing[i] = m;
}

// LRUCache.Each (lrucache.go:76-85): compact every resident bucket (+ its key cell) into out arrays
__global__ __launch_bounds__(256) void k_dump(Table T, uint64_t slots, Rec* out_recs, KeyCell* out_cells, uint64_t cap,
                                              unsigned long long* count) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= slots) return;
    if (T.dir[s].tag == 0ull) return;
    Rec r = T.buckets[s].rec;
    if (rec_kind(r) == K_ABSENT) return;
    unsigned long long idx = atomicAdd(count, 1ull);
    if (idx < cap) { out_recs[idx] = r; out_cells[idx] = T.buckets[s].cell; }
}

// globalManager flush (global.go:114-139 / 200-215): turn every pending record into one request row
// (key bytes from the bucket's key cell, summed hits / template fields) and clear it.
struct GTakeOut {
    uint8_t* key_bytes; uint32_t* key_len;      // key i occupies key_bytes[i*stride .. +key_len[i])
    int64_t *hits, *limit, *duration, *burst, *created_at;
    uint32_t* behavior; uint8_t* algorithm; uint8_t* role;   // role 1 = hits for the owner, 2 = owner update
    uint32_t stride;
};
__global__ __launch_bounds__(256) void k_global_take(Table T, uint32_t n, uint32_t role_mask, uint32_t* keep_list,
                                                     unsigned int* counters /* [0] rows out, [1] kept */, GTakeOut O) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t slot = T.gdirty[j];
    GPend p = T.gpend[slot];
    if (!((role_mask >> p.queued) & 1u)) {            // not asked for: stays pending
        keep_list[atomicAdd(&counters[1], 1u)] = slot;
        return;
    }
    const uint32_t i = atomicAdd(&counters[0], 1u);
    const KeyCell* c = &T.buckets[slot].cell;
    const uint32_t len = (uint32_t)(c->w[7] >> 48);
    const uint8_t* src = len <= INLINE_KEY ? (const uint8_t*)c->w : T.arena + c->w[0];
    uint8_t* dst = O.key_bytes + (size_t)i * O.stride;
    for (uint32_t b = 0; b < O.stride; ++b) dst[b] = b < len ? src[b] : 0;
    O.key_len[i] = len;
    O.hits[i] = p.hits; O.limit[i] = p.limit; O.duration[i] = p.duration; O.burst[i] = p.burst;
    O.created_at[i] = p.created_at; O.behavior[i] = p.behavior; O.algorithm[i] = p.algorithm; O.role[i] = p.queued;
    GPend z; __builtin_memset(&z, 0, sizeof(z));
    T.gpend[slot] = z;
}

// Table compaction: re-insert every LIVE bucket (present and not expired at `now`) of the old table into a
// fresh one.  Expired buckets are indistinguishable from absent ones for the algorithm (lrucache.go:115-119
// removes them on access), removed buckets (K_ABSENT) only kept their tag for probing; both are dropped, which
// frees their directory entries — the stand-in for the reference's bounded LRU (lrucache.go:98-100,138-149).
__global__ __launch_bounds__(256) void k_compact(Table Old, uint64_t old_slots, Table New, int64_t now,
                                                 unsigned long long* kept) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= old_slots) return;
    const unsigned long long tag = Old.dir[s].tag;
    if (tag == 0ull) return;
    const Bucket b = Old.buckets[s];
    if (rec_kind(b.rec) == K_ABSENT || rec_expired(b.rec, now)) return;
    uint64_t pos = ((tag == 1ull ? 0ull : tag) >> 7) & New.mask;     // same home position rule as probe()
    for (uint64_t step = 0; step <= New.mask; ++step, pos = (pos + 1) & New.mask) {
        if (atomicCAS(&New.dir[pos].tag, 0ull, tag) == 0ull) {
            New.dir[pos].meta = META_READY;
            New.buckets[pos] = b;                                     // long keys keep their arena offset
            atomicAdd(kept, 1ull);
            return;
        }
    }
}

// wrap of the 31-bit batch epoch: forget every dense-id claim
__global__ __launch_bounds__(256) void k_clear_claims(Table T, uint64_t slots) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < slots) T.dir[s].meta &= META_READY;
}

// ReplicatedConsistentHash.Get (replicated_hash.go:104-119): owner of each key on a sorted ring.
__global__ __launch_bounds__(256) void k_route(const uint8_t* key_bytes, const uint32_t* key_off, uint32_t n,
                                               const uint64_t* ring_hash, const uint32_t* ring_owner, uint32_t npts,
                                               int kind, uint32_t* owner) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* lh = (uint64_t*)smem;
    for (uint32_t j = threadIdx.x; j < npts; j += 256) lh[j] = ring_hash[j];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* k = key_bytes + key_off[i];
    const uint32_t len = key_off[i + 1] - key_off[i];
    const uint64_t h = kind == 1 ? fnv1a_64(k, len) : fnv1_64(k, len);
    uint32_t lo = 0, hi = npts;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (lh[mid] >= h) hi = mid; else lo = mid + 1; }
    if (lo == npts) lo = 0;
    owner[i] = ring_owner[lo];
}

// the same for keys stored as rows of a [n][stride] matrix with explicit lengths (guber_global_take_dev rows)
__global__ __launch_bounds__(256) void k_route_rows(const uint8_t* key_rows, uint32_t stride, const uint32_t* key_len, uint32_t n,
                                                    const uint64_t* ring_hash, const uint32_t* ring_owner, uint32_t npts,
                                                    int kind, uint32_t* owner) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* lh = (uint64_t*)smem;
    for (uint32_t j = threadIdx.x; j < npts; j += 256) lh[j] = ring_hash[j];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* k = key_rows + (size_t)i * stride;
    const uint32_t len = key_len[i];
    const uint64_t h = kind == 1 ? fnv1a_64(k, len) : fnv1_64(k, len);
    uint32_t lo = 0, hi = npts;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (lh[mid] >= h) hi = mid; else lo = mid + 1; }
    if (lo == npts) lo = 0;
    owner[i] = ring_owner[lo];
}

// guber_add_items_dev: build the ItemIn image (bucket record + key reference) of device-resident item columns —
// the device twin of rec_from_item() in guber_engine.hip (UpdatePeerGlobals / Loader items, gubernator.go:425-459)
struct ItemsSoA {
    const uint32_t* key_off; const uint8_t *algorithm, *status;
    const int64_t *limit, *duration, *remaining; const double* remaining_f;
    const int64_t *stamp, *burst, *expire_at, *invalid_at;
};
__global__ __launch_bounds__(256) void k_items_from_soa(ItemsSoA S, uint32_t n, ItemIn* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Rec s; rec_clear(s);
    const uint8_t algo = S.algorithm[i];
    s.limit = S.limit[i]; s.duration = S.duration[i]; s.stamp = S.stamp[i]; s.burst = S.burst ? S.burst[i] : 0;
    s.expire_at = S.expire_at[i]; s.invalid_at = S.invalid_at ? S.invalid_at[i] : 0;
    if (algo == ALGO_TOKEN) { s.remaining = S.remaining[i]; s.burst = 0; s.meta = make_meta(K_TOKEN, S.status ? S.status[i] : 0, ALGO_TOKEN); }
    else if (algo == ALGO_LEAKY) { s.remaining = f2bits(S.remaining_f[i]); s.meta = make_meta(K_LEAKY, 0, ALGO_LEAKY); }
    else s.meta = make_meta(K_NIL, 0, algo);
    ItemIn o; o.rec = s; o.key_off = S.key_off[i]; o.key_len = S.key_off[i + 1] - S.key_off[i];
    out[i] = o;
}

}  // namespace guber
