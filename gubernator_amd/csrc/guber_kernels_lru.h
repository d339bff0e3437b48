// guber_kernels_lru.h — the bounded cache's EXACT victim order (lrucache.go:88-149) for batches.
//
// The reference keeps a list ordered by last access: Add / GetItem move an item to the front (:91, :96, :123), an insert that
// makes the list longer than cacheSize removes the item at the back (:98-100, :138-149) — at once, in the middle of a stream of
// requests, so a key evicted by request i is a NEW item for request j > i.  A batch evaluates its requests in parallel; what makes
// it exact anyway is that the list's order is a pure function of the access sequence:
//
//   * every bucket carries the sequence number of its last access (rec_stamp: request i of a batch = the batch's first number + i);
//     older stamp = nearer the back;
//   * at the start of a batch of n <= cacheSize requests let pos0(k) be key k's distance from the front.  Until k is accessed its
//     distance grows by one for every DISTINCT key accessed before it that was not in front of it (new, or further back); keys in
//     front of it that are accessed stay in front of it.  k is still in the list at its first access f(k) iff
//         pos0(k) + #{ j : f(j) < f(k), j new or pos0(j) > pos0(k) }  <=  cacheSize - 1
//     (everything in front of k is in the list as long as k is, so the left side is a lower bound of the list's length);
//   * after its first access a key is at the front and cannot be evicted again within n <= cacheSize requests;
//   * the items the batch never touches leave from the back: the oldest len0 + (new keys) - cacheSize of them.
//
// Two kinds of request do not fit these rules, because they change the list's LENGTH where the rules only move keys to the front:
//   * a TOKEN_BUCKET request with RESET_REMAINING for a key that is in the cache REMOVES the item (algorithms.go:78-90) and inserts
//     nothing: the list is one shorter until the key's next request, or for good — one eviction less for whoever inserts next, and
//     everything behind the key has moved up one place;
//   * the FIRST request of a resident key that fails inside the algorithm before c.Add (DURATION_IS_GREGORIAN with a duration that is
//     no interval constant: lru_cannot_insert) is no arrival at the front if the key had been pushed out before (GetItem misses,
//     nothing is inserted: the key re-enters at its first good request, or never) or if its item had expired (GetItem removes it,
//     cache.go:43-57 / lrucache.go:111-128, and nothing takes its place).
// Both are rare (a client resetting a limit; a client's invalid constant), so they are not modelled, they are ISOLATED: the pre-pass
// reports the first such request (LruCtl::split_at, status LRU_SPLIT) and the host evaluates the requests before it as a batch of
// their own — for which the rules hold as they stand —, then that request alone (a batch of one: its key is first, nothing precedes
// it, and one request cannot overflow a cache that was not over its size), then the rest, each with its own pre-pass.  Exact by
// construction, at the price of a pre-pass per such request while the cache binds.  (Round 5 documented the second kind as not
// reproduced and had not noticed the first.)
//
// So before a batch that may overflow the cache (host: live + n > cacheSize) a PRE-PASS decides which resident keys are evicted
// before their first access (their buckets become absent: the batch's pipeline — any of them — then sees a new key, as the
// reference does) and which untouched items go; the pipeline itself is unchanged.  Only the items near the back matter: the
// engine keeps the TAIL LIST — (stamp, slot) of the live items sorted by stamp, built by one table scan + sort and used for many
// batches; an entry is valid as long as its bucket still carries that stamp (any access gives a bucket a newer stamp, eviction and
// Remove make it absent), so nothing has to maintain the list — and per batch looks at a WINDOW of it: the oldest valid entries,
// as many as the batch can reach (len0 + keys of the batch - cacheSize).  Items inserted after the list was built are newer than
// every entry.  A batch larger than cacheSize is cut into pieces of cacheSize requests by the host (launch_batch).
//
// Launch sequence (guber_engine.hip lru_admit; the host reads LruCtl::status afterwards):
//   k_lru_begin  k_lru_probe  k_lru_keys  k_lru_win_flag  k_lru_scan_u32  k_lru_win_emit  k_lru_check
//   k_lru_risk  k_lru_scan_u8 (x2)  k_lru_decide  k_lru_evict  k_lru_end
// k_lru_gather feeds the sort that builds the tail list.  The same source runs on the CPU in tests/hostsim/devsim.cpp.
#pragma once
#include "guber_table.h"

namespace guber {

enum : uint32_t { LRU_NONE = 1,      // the batch cannot overflow the cache: nothing to do
                  LRU_APPLIED = 2,   // evictions decided and applied
                  LRU_MORE = 3,      // the window holds too few valid entries and the tail list goes on: look at a longer window
                  LRU_REBUILD = 4,   // the tail list is used up: build a new one
                  LRU_CUT = 5,       // evictions are due and the batch is larger than the cache: evaluate it in pieces
                  LRU_SPLIT = 6 };   // evictions are due and a RESIDENT key's first request cannot insert (LruCtl::split_at): that request is evaluated on its own

struct LruCtl {
    unsigned long long cursor, tail_n;             // the tail list: first entry that may still be valid, entries (persistent)
    long long len0;                                // items in the cache before the batch
    unsigned long long evicted, unexpired;         // this call: buckets made absent, of which not yet expired (lrucache.go:142-146)
    uint32_t m_new, m_res, n_risk, win_len, win_valid, zone, evict_untouched, first_left, status, split_at;
};

// where the keys of the requests (or of the items of an Add) are
struct LruKeys {
    const uint8_t* bytes;
    const uint8_t* off_p; uint32_t off_stride;     // u32 offset of key i at off_p + i * off_stride (null: i * key_stride)
    const uint8_t* len_p; uint32_t len_stride;     // u32 length likewise (null: off[i + 1] - off[i])
    uint32_t key_stride;
    const uint8_t* algorithm;                      // requests only: an invalid algorithm never reaches the cache (workers.go:317-321)
    // requests only: DURATION_IS_GREGORIAN with a duration that is no interval constant fails in tokenBucketNewItem / leakyBucketNewItem
    // BEFORE c.Add (algorithms.go; interval.go:93,107,125,148): for a key that is not in the cache such a request is a GetItem miss and
    // nothing else — it inserts nothing (for a resident key it is an access like any other: GetItem has moved the item to the front)
    const uint32_t* behavior; const int64_t* duration; const int64_t* greg_duration;   // (greg_duration: host-precomputed, < 0 = the error)
};
__device__ __forceinline__ bool lru_cannot_insert(const LruKeys& K, uint32_t i) {
    if (!K.behavior || !(K.behavior[i] & BH_GREGORIAN)) return false;
    if (K.greg_duration) return K.greg_duration[i] < 0;
    const int64_t d = K.duration ? K.duration[i] : 0;
    return d < 0 || d > 5 || d == 3;                 // GregorianWeeks: "not supported" (interval.go:97,130)
}
__device__ __forceinline__ uint32_t lru_u32(const uint8_t* p, uint32_t stride, uint32_t i) { return *(const uint32_t*)(p + (size_t)i * stride); }

// the keys of the batch, grouped: an insert-only hash table id -> first request index (cells = pow2 >= 2 n, all bits set = empty).
// id: bit 63 set = the key has a directory entry (slot in the low 32 bits; bit 62 = its bucket is live), else 63 bits of a
// second, independent hash of the key bytes (a key the table has never seen)
// first = the key's first request, first_ok = its first request that can insert it (all bits set: none — see lru_cannot_insert)
// first_reset = its first TOKEN_BUCKET request with RESET_REMAINING (all bits set: none)
struct LruGroups { unsigned long long* id; uint32_t* first; uint32_t* first_ok; uint32_t* first_reset; uint32_t mask; };
struct LruRes { uint32_t* first; uint32_t* slot; unsigned long long* stamp; };           // resident keys of the batch
struct LruWin { unsigned long long* stamp; uint32_t* slot; uint32_t* widx; };             // the window's valid entries, oldest first
struct LruRisk { uint32_t* first; uint32_t* rank; uint32_t* slot; };                     // resident keys inside the zone

__global__ __launch_bounds__(256) void k_lru_begin(Table T, LruCtl* C, uint32_t n_bctr) {
    __shared__ long long part[256];
    long long s = 0;
    for (uint32_t b = threadIdx.x; b < n_bctr; b += 256) s += T.bctr[b].size_delta;
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = T.ctr->size;
        for (int i = 0; i < 256; ++i) t += part[i];
        C->len0 = t < 0 ? 0 : t;
        C->evicted = C->unexpired = 0ull;
        C->m_new = C->m_res = C->n_risk = C->win_len = C->win_valid = C->zone = C->evict_untouched = 0u;
        C->first_left = 0xffffffffu; C->status = 0u; C->split_at = 0xffffffffu;
    }
}

__global__ __launch_bounds__(256) void k_lru_probe(Table T, LruKeys K, uint32_t n, LruGroups G) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (K.algorithm && K.algorithm[i] > ALGO_LEAKY) return;
    const uint32_t off = K.off_p ? lru_u32(K.off_p, K.off_stride, i) : i * K.key_stride;
    const uint32_t len = K.len_p ? lru_u32(K.len_p, K.len_stride, i) : lru_u32(K.off_p, K.off_stride, i + 1) - off;
    if (len == 0 || len > T.max_key) return;
    const uint8_t* key = K.bytes + off;
    const uint64_t h = xxhash64(key, len, 0);
    uint32_t slot = 0;
    const uint32_t pr = probe(T, key, len, h, false, slot);
    unsigned long long id;
    if (pr & PR_FOUND) {
        const bool live = rec_kind(T.buckets[slot].rec) != K_ABSENT;
        id = (1ull << 63) | (live ? 1ull << 62 : 0ull) | slot;
    } else {
        id = xxhash64(key, len, 0x9e3779b97f4a7c15ull) >> 1;
    }
    uint32_t c = (uint32_t)((id * 0x9e3779b97f4a7c15ull) >> 40) & G.mask;
    for (;;) {
        unsigned long long cur = G.id[c];
        if (cur == ~0ull) { const unsigned long long old = atomicCAS(&G.id[c], ~0ull, id); cur = old == ~0ull ? id : old; }
        if (cur == id) {
            atomicMin(&G.first[c], i);
            if (!lru_cannot_insert(K, i)) atomicMin(&G.first_ok[c], i);
            if (K.behavior && (K.behavior[i] & BH_RESET_REMAINING) && (!K.algorithm || K.algorithm[i] == ALGO_TOKEN)) atomicMin(&G.first_reset[c], i);
            return;
        }
        c = (c + 1) & G.mask;
    }
}

// one thread per cell: the distinct keys.  New keys mark the request that inserts them; resident ones are listed with their stamp.
__global__ __launch_bounds__(256) void k_lru_keys(Table T, LruGroups G, LruCtl* C, uint8_t* isnew_at, LruRes R) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c > G.mask) return;
    const unsigned long long id = G.id[c];
    if (id == ~0ull) return;
    const uint32_t f = G.first[c];
    // requests that change the list's length are evaluated on their own (file header): a token RESET of a key that is in the cache by then
    // (resident, or inserted by an earlier request of this batch; a RESET that is a new key's first request is an insert like any other —
    // it goes alone all the same, so that the key's next RESET finds it resident), and a resident key's first request that cannot insert
    if (G.first_reset[c] != 0xffffffffu) atomicMin(&C->split_at, G.first_reset[c]);
    if ((id >> 62) == 3ull) {
        const uint32_t slot = (uint32_t)id;
        const uint32_t k = atomicAdd(&C->m_res, 1u);
        R.first[k] = f; R.slot[k] = slot; R.stamp[k] = rec_stamp(T.buckets[slot].rec);
        if (G.first_ok[c] != f) atomicMin(&C->split_at, f);
    } else {
        const uint32_t fo = G.first_ok[c];               // a new key is inserted by its first request that gets as far as c.Add
        if (fo == 0xffffffffu) return;                   // (none does: the key never enters the list)
        atomicAdd(&C->m_new, 1u);
        isnew_at[fo] = 1;
    }
}

// ---- the window: entries [cursor, cursor + w_len) of the tail list, valid ones flagged and counted per workgroup ----
__global__ __launch_bounds__(256) void k_lru_win_flag(Table T, const unsigned long long* t_stamp, const uint32_t* t_slot, LruCtl* C, uint32_t w_len,
                                                      uint8_t* wflag, uint32_t* blockcnt) {
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0u;
    __syncthreads();
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long idx = C->cursor + w;
    bool valid = false;
    if (w < w_len && idx < C->tail_n) {
        const Rec r = T.buckets[t_slot[idx]].rec;
        valid = rec_kind(r) != K_ABSENT && rec_stamp(r) == t_stamp[idx];
    }
    if (w < w_len) wflag[w] = valid ? 1 : 0;
    if (valid) atomicAdd(&cnt, 1u);
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = cnt;
}
// exclusive prefix sums by ONE workgroup (the pre-pass is not the hot path): v[0 .. n) in place, the total to *total
__global__ __launch_bounds__(1024) void k_lru_scan_u32(uint32_t* v, uint32_t n, uint32_t* total) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint32_t t = part[i]; part[i] = run; run += t; }
        if (total) *total = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t t = v[i]; v[i] = run; run += t; }
}
__global__ __launch_bounds__(1024) void k_lru_scan_u8(const uint8_t* in, uint32_t n, uint32_t* out) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += in[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint32_t t = part[i]; part[i] = run; run += t; }
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { out[i] = run; run += in[i]; }
}
__global__ __launch_bounds__(256) void k_lru_win_emit(const unsigned long long* t_stamp, const uint32_t* t_slot, const LruCtl* C, uint32_t w_len,
                                                      const uint8_t* wflag, const uint32_t* blockoff, LruWin Z) {
    __shared__ uint32_t fl[256];
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    const bool valid = w < w_len && wflag[w];
    fl[threadIdx.x] = valid ? 1u : 0u;
    __syncthreads();
    if (!valid) return;
    uint32_t before = 0;
    for (uint32_t t = 0; t < threadIdx.x; ++t) before += fl[t];
    const uint32_t z = blockoff[blockIdx.x] + before;
    const unsigned long long idx = C->cursor + w;
    Z.stamp[z] = t_stamp[idx]; Z.slot[z] = t_slot[idx]; Z.widx[z] = w;
}

// what this batch needs: the zone (how far from the back it can reach), the untouched items that leave, and whether the window
// covers it.  w_len = the window looked at, n = requests of the batch, cache_size = cacheSize, trim = 1: no batch, the cache is
// only brought down to cache_size (after a configuration change / as a safety net).
__global__ void k_lru_check(LruCtl* C, uint32_t w_len, uint32_t n, uint64_t cache_size) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long N = (long long)cache_size, len0 = C->len0;
    const long long m_new = C->m_new, m = m_new + C->m_res;
    C->win_len = (uint32_t)((C->cursor + w_len <= C->tail_n) ? w_len : (C->tail_n > C->cursor ? C->tail_n - C->cursor : 0ull));
    if (len0 + m_new <= N) { C->status = LRU_NONE; return; }
    if ((long long)n > N) { C->status = LRU_CUT; return; }
    if (C->split_at != 0xffffffffu && n > 1) { C->status = LRU_SPLIT; return; }
    long long zone = len0 + m - N;                       // <= len0 because m <= n <= N
    if (zone > len0) zone = len0;
    C->zone = (uint32_t)zone;
    C->evict_untouched = (uint32_t)(len0 + m_new - N);
    if ((long long)C->win_valid < zone) { C->status = (C->cursor + w_len < C->tail_n) ? LRU_MORE : LRU_REBUILD; return; }
    C->status = LRU_APPLIED;
}

// resident keys of the batch that lie inside the zone: their rank from the back, and the window entry marked as touched
__global__ __launch_bounds__(256) void k_lru_risk(const LruCtl* C, LruRes R, LruWin Z, uint8_t* ztouched, LruRisk Q, uint32_t* n_risk) {
    if (C->status != LRU_APPLIED) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= C->m_res) return;
    const uint32_t zone = C->zone;
    if (zone == 0) return;
    const unsigned long long st = R.stamp[k];
    if (st > Z.stamp[zone - 1]) return;
    uint32_t lo = 0, hi = zone;
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (Z.stamp[mid] < st) lo = mid + 1; else hi = mid; }
    if (lo >= zone || Z.stamp[lo] != st) return;          // (an item that is older than the zone's end but not in the list cannot exist)
    ztouched[lo] = 1;
    const uint32_t q = atomicAdd(n_risk, 1u);
    Q.first[q] = R.first[k]; Q.rank[q] = lo; Q.slot[q] = R.slot[k];
}

__device__ __forceinline__ void lru_drop(Table& T, LruCtl* C, uint32_t slot, int64_t now) {
    const Rec r = T.buckets[slot].rec;
    Rec z; rec_clear(z);
    T.buckets[slot].rec = z;
    atomicAdd(&C->evicted, 1ull);
    if (now < r.expire_at) atomicAdd(&C->unexpired, 1ull);          // lrucache.go:142-144
}

// a resident key inside the zone is gone before its first access iff its distance from the front has reached cacheSize by then
__global__ __launch_bounds__(256) void k_lru_decide(Table T, LruCtl* C, uint64_t cache_size, LruRisk Q, const uint32_t* n_risk, const uint32_t* new_before,
                                                    int64_t now) {
    if (C->status != LRU_APPLIED) return;
    const uint32_t nr = *n_risk;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nr) return;
    const uint32_t f = Q.first[q], r = Q.rank[q];
    long long c = new_before[f];
    for (uint32_t j = 0; j < nr; ++j) c += (Q.rank[j] < r && Q.first[j] < f) ? 1 : 0;
    const long long pos0 = C->len0 - 1 - (long long)r;
    if (pos0 + c > (long long)cache_size - 1) lru_drop(T, C, Q.slot[q], now);
}

// the untouched items that leave: the first evict_untouched valid, untouched entries of the window
__global__ __launch_bounds__(256) void k_lru_evict(Table T, LruCtl* C, LruWin Z, const uint8_t* ztouched, const uint32_t* touched_before, int64_t now) {
    if (C->status != LRU_APPLIED) return;
    const uint32_t z = blockIdx.x * 256 + threadIdx.x;
    if (z >= C->zone) return;
    if (ztouched[z]) return;
    if (z - touched_before[z] < C->evict_untouched) lru_drop(T, C, Z.slot[z], now);
    else atomicMin(&C->first_left, z);
}

__global__ void k_lru_end(Table T, LruCtl* C, LruWin Z) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (C->status != LRU_APPLIED) return;
    atomicAdd((unsigned long long*)&T.ctr->size, (unsigned long long)(-(long long)C->evicted));
    atomicAdd(&T.ctr->evictions, C->unexpired);
    // the list's head moves to the oldest entry that is still valid after this batch
    if (C->first_left != 0xffffffffu) C->cursor += Z.widx[C->first_left];
    else if (C->zone < C->win_valid) C->cursor += Z.widx[C->zone];
    else C->cursor += C->win_len;
}

// ---- building the tail list: every live bucket's (stamp, slot), unordered; the host sorts by stamp ----
__global__ __launch_bounds__(256) void k_lru_gather(Table T, uint64_t slots, unsigned long long* stamp, uint32_t* slot, uint64_t cap, unsigned long long* count) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= slots || T.dir[s].tag == 0ull) return;
    const Rec r = T.buckets[s].rec;
    if (rec_kind(r) == K_ABSENT) return;
    const unsigned long long k = atomicAdd(count, 1ull);
    if (k < cap) { stamp[k] = rec_stamp(r); slot[k] = (uint32_t)s; }
}

}  // namespace guber
