// guber_kernels_ops.h — everything that is not batch evaluation: cache operations (AddCacheItem / GetCacheItem / Remove /
// Each), Store residency probe, GLOBAL queue flush, table compaction, consistent-hash routers, device item columns.
#pragma once
#include "guber_table.h"

namespace guber {

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
constexpr uint64_t ITEMS_KEEP_STAMP = ~0ull;
__global__ __launch_bounds__(256) void k_items_commit(Table T, const ItemIn* items, const uint8_t* keys, uint32_t n,
                                                      const uint32_t* slots, const uint8_t* flags, uint8_t* result, uint64_t touch) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t f = flags[i];
    const uint32_t slot = slots[i];
    if (f & RF_INSERTED) atomicOr(&T.dir[slot].meta, META_READY);
    if (f & RF_ERR) { result[i] = 0xFE; return; }
    if ((f & RF_NEED_VERIFY) && !key_equal(T, slot, keys + items[i].key_off, items[i].key_len)) { result[i] = 0xFF; return; }
    const bool existed = rec_kind(T.buckets[slot].rec) != K_ABSENT;
    Rec nr = items[i].rec;
    // LRUCache.Add pushes / moves the item to the front (lrucache.go:91,96); item i of the call after item i - 1.  ITEMS_KEEP_STAMP: the
    // host has numbered the items itself (guber_add_items applies a call with duplicate keys in several launches: the numbers follow
    // the items' places in the CALL, so the recency order among the call's keys is the reference's whatever the launches were)
    if (touch != ITEMS_KEEP_STAMP) rec_set_stamp(nr, touch + i);
    T.buckets[slot].rec = nr;
    if (!existed) atomicAdd((unsigned long long*)&T.ctr->size, 1ull);
    result[i] = existed ? 1 : 0;
}

// Take the buckets of up to n keys out of a table BY KEY HASH (the directory tag is the key's XXH64): the item and its key
// bytes go to a staging image that k_items_probe / k_items_commit of ANOTHER table on the same device consume (a hot key moves
// to another logical shard: GPUWorkerPool, guber_move_items_by_hash).  A hash that names no live bucket yields key_len 0 (the
// probe skips it).  One thread per hash; `stride` bytes of key room each.
__global__ __launch_bounds__(64) void k_items_take_by_hash(Table T, const uint64_t* hashes, uint32_t n, uint32_t stride, ItemIn* items, uint8_t* keys) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    ItemIn out; rec_clear(out.rec); out.key_off = i * stride; out.key_len = 0;
    const uint64_t h = hashes[i] & T.hash_mask;
    const unsigned long long tag = h ? h : 1ull;
    uint64_t pos = (h >> 7) & T.mask;
    for (uint32_t step = 0; step < T.max_probe; ++step, pos = (pos + 1) & T.mask) {
        const unsigned long long t = ld_agent(&T.dir[pos].tag);
        if (t == 0ull) break;
        if (t != tag || !(ld_agent(&T.dir[pos].meta) & META_READY)) continue;
        const Rec s = T.buckets[pos].rec;
        if (rec_kind(s) == K_ABSENT) break;
        const KeyCell* c = &T.buckets[pos].cell;
        const uint32_t len = (uint32_t)(c->w[7] >> 48);
        if (len == 0 || len > stride || len > T.max_key) break;
        const uint8_t* src = len > INLINE_KEY ? T.arena + c->w[0] : (const uint8_t*)c->w;
        for (uint32_t b = 0; b < len; ++b) keys[(size_t)i * stride + b] = src[b];
        for (uint32_t b = len; b < ((len + 15u) & ~7u) && b < stride; ++b) keys[(size_t)i * stride + b] = 0;
        out.rec = s; out.key_len = len;
        Rec z; rec_clear(z);
        T.buckets[pos].rec = z;                                  // (the tag stays, as for any removed bucket)
        atomicAdd((unsigned long long*)&T.ctr->size, (unsigned long long)(long long)-1);
        break;
    }
    items[i] = out;
}

// Undo k_items_take_by_hash for the items the destination table did not take (result >= 2: table full, a colliding key, or the
// whole commit never ran): the bucket goes back where it was — its tag and key bytes never left — so a failed migration loses nothing.
// result[i] becomes 0xFD for a bucket that went back.
__global__ __launch_bounds__(64) void k_items_restore(Table T, const ItemIn* items, const uint8_t* keys, uint32_t n, uint8_t* result) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n || result[i] < 2 || items[i].key_len == 0) return;
    const uint8_t* key = keys + items[i].key_off;
    uint32_t slot = 0;
    const uint32_t pr = probe(T, key, items[i].key_len, xxhash64(key, items[i].key_len, 0), false, slot);
    if (!(pr & PR_FOUND)) return;
    T.buckets[slot].rec = items[i].rec;
    atomicAdd((unsigned long long*)&T.ctr->size, 1ull);
    result[i] = 0xFD;
}

// LRUCache.GetItem (lrucache.go:111-128) / Remove (:131-135) for one key. mode 0 = get, 1 = remove
__global__ void k_item_lookup(Table T, const uint8_t* key, uint32_t len, int64_t now, int mode, Rec* out, int* found, uint64_t touch) {
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
    rec_set_stamp(s, touch); T.buckets[slot].rec = s;   // GetItem moves the item to the front (lrucache.go:123)
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

// Table compaction: re-insert every bucket that holds an item into a fresh table.  Removed / evicted buckets (K_ABSENT) only kept
// their tag for probing and are dropped, which frees their directory entries.  An item that has expired is still an item of the
// reference's list until somebody asks for it or it reaches the back (lrucache.go:115-119, :138-149) — it takes a place, so it
// moves with the others (`live` = the items moved).  A bucket with a pending GLOBAL record is kept whatever its state and the record moves with it (new dirty list);
// long keys are copied into a fresh arena, so the space of dropped keys is reclaimed.
struct CompactOut { unsigned long long kept, live, arena_head; unsigned int gdirty_n; };
__global__ __launch_bounds__(256) void k_compact(Table Old, uint64_t old_slots, Table New, int64_t now, CompactOut* out) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= old_slots) return;
    const unsigned long long tag = Old.dir[s].tag;
    if (tag == 0ull) return;
    Bucket b = Old.buckets[s];
    const bool pending = Old.gpend && Old.gpend[s].queued != 0;
    const bool dead = rec_kind(b.rec) == K_ABSENT;
    if (dead && !pending) return;
    const uint32_t len = (uint32_t)(b.cell.w[7] >> 48);
    if (len > INLINE_KEY && len != 0xffffu) {
        const uint64_t need = ((uint64_t)len + 7) & ~7ull;
        const uint64_t off = atomicAdd(&out->arena_head, (unsigned long long)need);   // <= the old head: always fits
        for (uint64_t q = 0; q < need; q += 8) *(uint64_t*)(New.arena + off + q) = *(const uint64_t*)(Old.arena + b.cell.w[0] + q);
        b.cell.w[0] = off;
    }
    uint64_t pos = ((tag == 1ull ? 0ull : tag) >> 7) & New.mask;     // same home position rule as probe()
    for (uint64_t step = 0; step <= New.mask; ++step, pos = (pos + 1) & New.mask) {
        if (atomicCAS(&New.dir[pos].tag, 0ull, tag) == 0ull) {
            New.dir[pos].meta = META_READY;
            New.buckets[pos] = b;
            if (pending) {
                New.gpend[pos] = Old.gpend[s];
                const uint32_t k = atomicAdd(&out->gdirty_n, 1u);
                if (k < New.gdirty_cap) New.gdirty[k] = (uint32_t)pos;
            }
            atomicAdd(&out->kept, 1ull);
            if (!dead) atomicAdd(&out->live, 1ull);
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
    GUBER_DYN_LDS(smem);
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
    GUBER_DYN_LDS(smem);
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

// the engine's event counters -> a snapshot in device-visible host memory (stages bracket their batch with two of these:
// one launch each instead of two small copies)
__global__ void k_ctr_snapshot(const DevCounters* ctr, const BlockCounters* bctr, uint32_t n_bctr, DevCounters* out_c, BlockCounters* out_b,
                               uint32_t* seq_out, uint32_t seq) {
    for (uint32_t k = threadIdx.x; k < n_bctr; k += blockDim.x) out_b[k] = bctr[k];
    if (threadIdx.x == 0) *out_c = *ctr;
    __syncthreads();                                             // every store of the workgroup has been issued and drained
    if (threadIdx.x == 0 && seq_out) {
        __threadfence_system();
        __hip_atomic_store(seq_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Request columns of up to MULTI_MEM_MAX stages (+ one more segment list: the fused launches' argument blocks, guber_kernels.h): device-visible host memory -> each stage's HBM mirror (same layout), ONE launch on the
// stream the batches' kernels follow on (the pool's dispatcher: a hipMemcpyAsync costs 40-60 us of host time, a launch 4-5).
// Only what the batch uses is moved: per column the first n entries (a stage's columns are laid out for max_n requests), then
// the key bytes.  16-byte loads over PCIe, fully coalesced; every column starts on a 64-byte boundary.
constexpr int STAGE_SEGS = 10;
struct StageIn { const uint4* src; uint4* dst; uint32_t nseg; uint32_t off16[STAGE_SEGS], n16[STAGE_SEGS]; };
struct MultiStageIn { uint32_t nb, wg_per; StageIn sub[MULTI_MEM_MAX + 1]; };
static_assert(sizeof(MultiStageIn) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void k_stage_in_multi(MultiStageIn A) {
    const uint32_t sb = blockIdx.x / A.wg_per, w = blockIdx.x - sb * A.wg_per;
    const StageIn* s = (const StageIn*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiStageIn, sub)) + sb;
    const uint32_t stride = A.wg_per * 256u;
    for (uint32_t k = 0; k < s->nseg; ++k) {
        const uint4* src = s->src + s->off16[k]; uint4* dst = s->dst + s->off16[k];
        const uint32_t n16 = s->n16[k];
        for (uint32_t i = w * 256u + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    }
}


// A stage whose requests belong to SEVERAL engines (guber_stage_submit_routed: the device-level stage of a pool, filled by the
// callers in arrival order): request i carries dest[i] = engine index << 24 | its rank inside that engine's share, the host
// knows the shares' sizes, so request i's place in HBM is base[engine] + rank — every engine's share ends up contiguous, in
// the order of the ranks, and the two-launch pipeline runs on it as on any batch.  ONE launch, three kinds of workgroups:
// [0, nb_req) scatter the fixed-width columns (coalesced reads over PCIe, 4/8-byte writes to HBM) and record where each
// request went (fwd: where its answer will be); [nb_req, nb_req + nb_key) copy the key bytes as they are (keys stay in
// arrival order: the requests carry offset + length); the rest copy the argument blocks of the launches that follow.
// The answers take the way back in a launch of their own (k_stage_out_routed): k_eval2 writes them to HBM next to the shares,
// and they reach the stage's host arrays in arrival order as full, coalesced lines (answers written from the shares' order
// straight over PCIe would be one small transaction each: measured 4x the batch's time on the device).
struct RoutedIn {
    uint32_t n, nb_req, nb_key, nb_arg;
    const uint32_t* dest; uint32_t base[MULTI_MEM_MAX];
    const uint32_t* key_off; const int64_t *hits, *limit, *duration, *burst, *created_at; const uint32_t* behavior; const uint8_t *algorithm, *is_owner;   // stage (host)
    uint32_t *d_key_off, *d_key_len, *d_fwd; int64_t *d_hits, *d_limit, *d_duration, *d_burst, *d_created_at; uint32_t* d_behavior; uint8_t *d_algorithm, *d_is_owner;   // HBM
    const uint4* key_src; uint4* key_dst; uint32_t key_n16;
    const uint4* arg_src; uint4* arg_dst; uint32_t arg_off16[2], arg_n16[2];
};
static_assert(sizeof(RoutedIn) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void k_stage_in_routed(RoutedIn A) {
    const uint32_t b = blockIdx.x;
    if (b < A.nb_req) {
        const uint32_t i = b * 256u + threadIdx.x;
        if (i >= A.n) return;
        const uint32_t dv = A.dest[i], o0 = A.key_off[i], o1 = A.key_off[i + 1];
        const int64_t hits = A.hits[i], limit = A.limit[i], duration = A.duration[i], burst = A.burst[i], created = A.created_at[i];
        const uint32_t beh = A.behavior[i]; const uint8_t algo = A.algorithm[i], owner = A.is_owner[i];
        const uint32_t d = A.base[(dv >> 24) & (MULTI_MEM_MAX - 1)] + (dv & 0xffffffu);
        A.d_fwd[i] = d < A.n ? d : 0u;
        if (d >= A.n) return;                                        // (ranks that are no permutation: nothing is written out of bounds)
        A.d_key_off[d] = o0; A.d_key_len[d] = o1 - o0;
        A.d_hits[d] = hits; A.d_limit[d] = limit; A.d_duration[d] = duration; A.d_burst[d] = burst; A.d_created_at[d] = created;
        A.d_behavior[d] = beh; A.d_algorithm[d] = algo; A.d_is_owner[d] = owner;
    } else if (b < A.nb_req + A.nb_key) {
        const uint32_t stride = A.nb_key * 256u;
        for (uint32_t i = (b - A.nb_req) * 256u + threadIdx.x; i < A.key_n16; i += stride) A.key_dst[i] = A.key_src[i];
    } else {
        const uint32_t stride = A.nb_arg * 256u;
        for (uint32_t k = 0; k < 2; ++k) {
            const uint4* src = A.arg_src + A.arg_off16[k]; uint4* dst = A.arg_dst + A.arg_off16[k];
            for (uint32_t i = (b - A.nb_req - A.nb_key) * 256u + threadIdx.x; i < A.arg_n16[k]; i += stride) dst[i] = src[i];
        }
    }
}

struct RoutedOut {
    uint32_t n; const uint32_t* fwd;
    const uint8_t *d_status, *d_err; const int64_t *d_limit, *d_remaining, *d_reset_time;      // HBM, in the shares' order
    uint8_t *status, *err; int64_t *limit, *remaining, *reset_time;                           // the stage's result arrays (host), arrival order
};
__global__ __launch_bounds__(256) void k_stage_out_routed(RoutedOut A) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= A.n) return;
    const uint32_t d = A.fwd[i];
    const uint8_t st = A.d_status[d], er = A.d_err[d];
    const int64_t l = A.d_limit[d], r = A.d_remaining[d], t = A.d_reset_time[d];
    A.limit[i] = l; A.remaining[i] = r; A.reset_time[i] = t; A.status[i] = st; A.err[i] = er;
}

}  // namespace guber
