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
#include <stddef.h>
#include <stdint.h>

#include "guber_table.h"

#include "guber_kernels_radix.h"

namespace guber {

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
//   0  key -> 64-bit hash; the request's own fields go to LDS.
//   1  group the tile by hash through an LDS hash table with per-wave member bitmaps: a request's rank inside its
//      (key, tile) group, the group size and the group's head come from four popcounts.  Only heads go on.
//   2  claim: a fresh look at the key's cell of the per-batch claim table, a CAS only when the cell is free.  Segment id of
//      a key = request index of its first toucher (the claimer).  A head that finds its key claimed by another tile is
//      done with the table (no directory entry, no bucket) and publishes its group: ONE atomic (size << 32 | tile bit) on
//      the segment's word; the claimer's own group travels in the segment record.
//   3  one key, one request shape per segment: members against their head (LDS), heads against the claimer's request.
//   4  the claimer: directory entry || home bucket in one trip, displaced bucket if needed, insert if absent, exact key
//      verification, and the segment record (bucket before the batch, slot, group size) in ONE 64-byte sector.  An entry
//      that is not READY (inserted during this launch) is trusted because every head compared its key with the claimer's.
// k_eval2 (request order): rank = members in earlier tiles (claimer's group, bitmap + per-tile counts) + rank in tile.
constexpr int FT = 256;                 // requests per tile in the two-launch pipeline
constexpr int FT_MAX_TILES = 256;       // bitmap bits per segment
constexpr int FT_WORDS = FT_MAX_TILES / 32;   // per segment: 8 x u64, each = members << 32 | bitmap of 32 tiles

__device__ __forceinline__ bool req_key_equal(const BatchView& B, uint32_t a, uint32_t b) {
    const uint32_t oa = key_off_of(B, a), ob = key_off_of(B, b);
    const uint32_t la = key_len_of(B, a, oa), lb = key_len_of(B, b, ob);
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

// claim the segment id of one key for this batch in the claims table (see Work::claims).  Insert-only open addressing:
// cells of older epochs count as empty; everybody scans from the same home cell in the same order, so a key is
// inserted at most once and later arrivers find it.  Cell = epoch16 << 48 | fp32 << 16 | request index of the first
// toucher.  The table is keyed by the 64-bit KEY HASH (home cell from bits 40.., fingerprint = low 32 bits), so the claim
// does not depend on the directory lookup and runs concurrently with it; two keys that agree in home-cell chain and
// fingerprint fall into one segment and are told apart by the exact key comparison against the claimer's request
// (SEG_RETRY -> careful round).  The careful round keys the same table by the verified bucket slot (fp32 = slot,
// exact).
__device__ __forceinline__ uint32_t claim_home_slot(uint32_t slot, uint32_t cmask) { return ((slot * 0x9E3779B1u) >> 11) & cmask; }
__device__ __forceinline__ uint32_t claim_home_hash(uint64_t h, uint32_t cmask) { return (uint32_t)(h >> 40) & cmask; }
// `first` = what the caller's CAS(0 -> want) on the home cell returned.  The part of the table a batch uses is zeroed again by its
// own k_eval2 (whole sectors, sequentially), so the expected value of a free cell is known without looking first: the claim
// and the speculative table fetch of a head travel in ONE round trip.  A cell that is neither zero nor of this epoch (left
// behind by a batch that was not cleaned up: an aborted launch pair, a wrapped epoch) is still taken over correctly, one
// trip later.  hcell ends up as the cell that holds the key.
__device__ __forceinline__ uint32_t claim_finish(unsigned long long* claims, uint32_t cmask, uint32_t& hcell, unsigned long long first,
                                                 uint32_t fp, uint32_t e16, uint32_t g, bool& claimed) {
    const unsigned long long want = ((unsigned long long)e16 << 48) | ((unsigned long long)fp << 16) | g;
    unsigned long long cur = first;
    for (;;) {
        if (cur == 0ull) { claimed = true; return g; }              // the CAS went through
        if ((uint32_t)(cur >> 48) == e16) {
            if ((uint32_t)(cur >> 16) == fp) return (uint32_t)(cur & 0xffffull);
            hcell = (hcell + 1) & cmask;                             // another key's cell: next
            cur = atomicCAS(&claims[hcell], 0ull, want);
            continue;
        }
        const unsigned long long old = atomicCAS(&claims[hcell], cur, want);   // a stale cell
        if (old == cur) { claimed = true; return g; }
        cur = old;                                                   // lost the race for it: look at the winner
    }
}

// per-request word handed from k_front to k_eval2: segment id | thread of the (segment, tile) group's head | rank in group
__device__ __forceinline__ uint32_t pack_dl(uint32_t d, uint32_t head_tid, uint32_t rank) { return (d << 16) | (head_tid << 8) | rank; }

// exact comparison of two request keys of the batch given their offsets and lengths (keys <= 64 bytes: every load is
// issued before any is consumed; longer keys loop)
__device__ __forceinline__ bool req_key_equal_at(const BatchView& B, uint32_t oa, uint32_t la, uint32_t ob, uint32_t lb) {
    if (la != lb) return false;
    const uint8_t* pa = B.key_bytes + oa; const uint8_t* pb = B.key_bytes + ob;
    const uint32_t nw = (la + 7) >> 3;
    if (nw <= 8) {
        uint64_t diff = 0;
#pragma unroll
        for (uint32_t w = 0; w < 8; ++w) {
            if (w < nw) {
                uint64_t x = ld_key_word(pa + 8 * w) ^ ld_key_word(pb + 8 * w);
                if (w == nw - 1) x &= tail_mask(la - 8 * w);
                diff |= x;
            }
        }
        return diff == 0;
    }
    for (uint32_t w = 0; w < nw; ++w) {
        uint64_t x = ld_key_word(pa + 8 * w) ^ ld_key_word(pb + 8 * w);
        if (w == nw - 1) x &= tail_mask(la - 8 * w);
        if (x) return false;
    }
    return true;
}
// SEG_* bits a request raises on its segment when its fields differ from the reference request's; soft = the requests
// differ only in created_at on a leaky bucket (decided against the bucket state by the caller)
__device__ __forceinline__ uint32_t req_diff_flags(const BatchView& B, uint32_t ia, uint32_t ib, const Req& a, const Req& b, bool& soft_leaky) {
    // a / b carry no calendar values (never read unless DURATION_IS_GREGORIAN is set): compare those only when it is
    // (without host-precomputed values they derive from the batch clock and `duration`, which is compared below)
    if ((a.behavior & BH_GREGORIAN) && a.behavior == b.behavior && B.greg_expire && B.greg_duration &&
        (B.greg_expire[ia] != B.greg_expire[ib] || B.greg_duration[ia] != B.greg_duration[ib]))
        return SEG_NONUNIFORM;
    if (req_eq(a, b)) return 0u;
    if (!req_eq_but_created(a, b)) return SEG_NONUNIFORM;
    if (a.algorithm == ALGO_LEAKY) { soft_leaky = true; return 0u; }
    return SEG_CREATED_DIFFERS;
}

// the tile's requests in LDS (structure of arrays, one 8-byte column per field): members compare themselves with
// their head without touching global memory
struct TileReqs {
    int64_t hits[FT], limit[FT], duration[FT], burst[FT], created_at[FT];
    unsigned long long misc[FT];          // behavior | algorithm << 32 | is_owner << 40
    // (the calendar values of DURATION_IS_GREGORIAN requests stay in global memory: they are compared only when the bit is set)
};
__device__ __forceinline__ void tile_put(TileReqs& t, uint32_t i, const Req& r) {
    t.hits[i] = r.hits; t.limit[i] = r.limit; t.duration[i] = r.duration; t.burst[i] = r.burst; t.created_at[i] = r.created_at;
    t.misc[i] = (unsigned long long)r.behavior | ((unsigned long long)r.algorithm << 32) | ((unsigned long long)r.is_owner << 40);
}
__device__ __forceinline__ Req tile_get(const TileReqs& t, uint32_t i) {
    Req r;
    r.hits = t.hits[i]; r.limit = t.limit[i]; r.duration = t.duration[i]; r.burst = t.burst[i]; r.created_at = t.created_at[i];
    r.greg_expire = 0; r.greg_duration = 0;
    const unsigned long long m = t.misc[i];
    r.behavior = (uint32_t)m; r.algorithm = (uint8_t)(m >> 32); r.is_owner = (uint8_t)(m >> 40);
    return r;
}

__device__ __forceinline__ void front_body(const Table& T, const BatchView& B, const Work& W, const uint32_t tile) {
    constexpr int GT_BITS = 9, GT = 1 << GT_BITS;                 // grouping table: 2 x FT entries
    __shared__ alignas(16) unsigned long long gkey[GT];           // grouping key (0 = free); after the grouping (same bytes): the heads' key words
    ulonglong2* const hkw = (ulonglong2*)gkey;                    // head thread -> {key bytes 0..7, 8..15} of a key of <= 16 bytes, zero padded
    static_assert(GT * 8 >= FT * 16, "the heads' key words fit where the hash table was");
    __shared__ unsigned long long gbits[FT / 64][GT];             // per wave: lanes holding the entry's key
    __shared__ uint32_t sd[FT];                                   // head -> segment id
    __shared__ uint32_t sslot[FT];                                // head -> bucket slot (for the rare member that needs the bucket)
    __shared__ uint32_t soff[FT], slen[FT];                       // where each request's key lives (members fetch their head's key)
    __shared__ TileReqs sreq;
    __shared__ int red[FT / 64];
    __shared__ uint32_t soft_any, ins_any;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = tile * FT + tid;
    const bool valid = g < B.n;
    unsigned long long* seg_mask = W.seg_tilemask + (size_t)W.parity * B.n_cap * FT_WORDS;
    const uint32_t e16 = W.epoch16;

    GB_STAMP(0);
    if (tile == 0 && W.snap_seq) {                                  // a counter read-back rides on this launch (Work::snap_*)
        for (uint32_t k = tid; k < W.snap_n; k += FT) W.snap_b[k] = T.bctr[k];
        if (tid == 0) *W.snap_c = *T.ctr;
        __syncthreads();                                             // every store of the workgroup issued and drained
        if (tid == 0) {
            __threadfence_system();                                  // (measured in round 6: a relaxed stamp instead changes nothing, profiles/r06_snapshot_release_ab.txt)
            __hip_atomic_store(W.snap_stamp, W.snap_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (tid == 0) { soft_any = 0u; ins_any = 0u; }
    for (uint32_t j = tid; j < GT; j += FT) {
        gkey[j] = 0ull;
#pragma unroll
        for (int w = 0; w < FT / 64; ++w) gbits[w][j] = 0ull;
    }
    // ---- stage 0: key -> hash.  Grouping key gk: the 64-bit hash; in a careful (retry) round the verified slot.  The
    // request's own fields are requested first: they arrive while the key is fetched and hashed and go to LDS ----
    uint32_t errcode = 0, off = 0, len = 0, slot = 0;
    int inserted = 0;
    bool cand = false, ready = false;
    const uint8_t* key = nullptr;
    unsigned long long gk = 0ull;
    uint64_t h = 0;
    Rec rec; rec_clear(rec);
    unsigned long long k0 = 0ull, k1 = 0ull;                       // a key of <= 16 bytes as two zero-padded words (members against their head, through LDS)
    if (valid) {
        // Fixed-width keys (every front end that formats its keys does): the key's words are requested at the offset the first
        // two keys suggest, together with the request's own offsets instead of after them — one dependent trip less.  The words
        // are used only if the offsets confirm the guess.
        uint64_t kw[4] = {0, 0, 0, 0};
        uint32_t off_g = 0, len_g = 0;
        bool spec = false;
        if (!B.key_stride && !B.key_len && !W.careful && B.n >= 2) {
            const uint32_t o0 = B.key_off[0], o1 = B.key_off[1], oend = B.key_off[B.n];
            len_g = o1 - o0; off_g = o0 + g * len_g;
            if (len_g != 0 && len_g < 32 && (uint64_t)off_g + 32 <= (uint64_t)oend + 8) {   // (the buffer is padded by 8 bytes)
                spec = true;
                const uint8_t* kp = B.key_bytes + off_g;
                kw[0] = ld_key_word(kp); kw[1] = ld_key_word(kp + 8); kw[2] = ld_key_word(kp + 16); kw[3] = ld_key_word(kp + 24);
            }
        }
        const Req mine = load_req_nogreg(B, g);
        off = key_off_of(B, g);
        len = key_len_of(B, g, off);
        key = B.key_bytes + off;
        if (len == 0) errcode = IE_EMPTY_KEY;
        else if (len > T.max_key) errcode = 7;
        GB_STAMPW(6);
        if (!errcode) {
            if (!(spec && off == off_g && len == len_g)) {
                spec = false;
                if (len <= 16) { kw[0] = ld_key_word(key); kw[1] = len > 8 ? ld_key_word(key + 8) : 0ull; }
            }
            h = (spec ? xxhash64_words4(kw, len, 0) : xxhash64(key, len, 0)) & T.hash_mask;
            if (len <= 16) {
                k0 = kw[0]; k1 = len > 8 ? kw[1] : 0ull;
                if (len < 8) k0 &= tail_mask(len); else if (len > 8 && len < 16) k1 &= tail_mask(len - 8);
            }
            if (W.careful) {
                // retry round: every request finds its bucket with a full, verifying probe BEFORE anything is claimed
                uint32_t cslot = 0;
                const uint32_t pr = probe(T, key, len, h, true, cslot);
                slot = cslot;
                inserted = (pr & PR_INSERTED) ? 1 : 0;
                if (pr & PR_FULL) errcode = 6;
                else { cand = true; rec = T.buckets[cslot].rec; gk = 0x8000000000000000ull | cslot; }
            } else {
                gk = h ? h : 1ull;
            }
        }
        tile_put(sreq, tid, mine);
        if (W.st_hits) {                                             // the batch is in host memory: keep a copy for k_eval2
            W.st_hits[g] = mine.hits; W.st_limit[g] = mine.limit; W.st_duration[g] = mine.duration; W.st_burst[g] = mine.burst;
            W.st_created[g] = mine.created_at; W.st_behavior[g] = mine.behavior; W.st_algorithm[g] = mine.algorithm; W.st_owner[g] = mine.is_owner;
        }
        GB_STAMPW(7);
    }
    soff[tid] = off; slen[tid] = len;
    lds_barrier();
    // ---- stage 1: group the tile by gk through an LDS hash table with per-wave member bitmaps: a request's rank
    // inside its (key, tile) group, the group size and the group's first thread come from four popcounts ----
    uint32_t gh = 0;
    if (gk) {
        gh = (uint32_t)((gk * 0x9E3779B97F4A7C15ull) >> (64 - GT_BITS));
        for (;;) {
            const unsigned long long old = atomicCAS(&gkey[gh], 0ull, gk);
            if (old == 0ull || old == gk) break;
            gh = (gh + 1) & (GT - 1);
        }
        atomicOr(&gbits[wave][gh], 1ull << lane);
    }
    lds_barrier();
    uint32_t eq_before = 0, eq_total = 1, head_tid = tid;
    if (gk) {
        bool found_head = false;
        eq_total = 0;
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
    const bool head = valid && eq_before == 0;
    const bool khead = head && gk != 0ull;                           // heads that have a key to resolve
    const bool member = valid && gk != 0ull && eq_before != 0;
    if (khead && len <= 16) hkw[tid] = make_ulonglong2(k0, k1);    // (the hash table is done with: every probe ended before the barrier above)
    GB_STAMP(1);

    // ---- stage 2 (heads): claim the key's cell and fetch directory entry + home bucket in the SAME round trip (most resident
    // keys sit at their home position at load <= 0.5).  A head that turns out not to be the claimer has fetched three sectors for
    // nothing; in exchange nobody waits for a look before the fetch.  Plain table loads: L1 may serve a line that is stale within
    // this launch, which is safe — a stale "empty" tag is corrected by the insert CAS, and READY never changes during k_front.
    uint32_t hcell = 0, fp = 0;
    unsigned long long first = 0ull;
    uint64_t pos = (h >> 7) & T.mask;
    ulonglong2 de0 = {0ull, 0ull};
    uint4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    if (khead) {
        if (W.careful) { hcell = claim_home_slot(slot, W.cmask); fp = slot; }
        else { hcell = claim_home_hash(h, W.cmask); fp = (uint32_t)h; }
        first = atomicCAS(&W.claims[hcell], 0ull, ((unsigned long long)e16 << 48) | ((unsigned long long)fp << 16) | g);
        if (!W.careful) {
            de0 = *(const ulonglong2*)&T.dir[pos];
            const Bucket* hb = &T.buckets[pos];
            const uint4* cw = (const uint4*)&hb->cell; c0 = cw[0]; c1 = cw[1]; c2 = cw[2]; c3 = cw[3];
            rec = hb->rec;
        }
    }
    // One key, one request shape per segment: members are compared with their tile's head, heads with the segment's
    // claimer (equality is transitive), on the exact key bytes and on every request field.  Members: fields from LDS.
    uint32_t my_flags = 0;                                           // SEG_* bits this request raises on its segment
    bool soft_leaky = false;
    if (member) my_flags |= req_diff_flags(B, g, tile * FT + head_tid, tile_get(sreq, tid), tile_get(sreq, head_tid), soft_leaky);
    uint32_t d = g;                                                  // error requests: a solo segment that only carries the code
    bool claimed = false;
    if (khead) {
        d = claim_finish(W.claims, W.cmask, hcell, first, fp, e16, g, claimed);
    }
    if (head) sd[tid] = d;
    // A head that is not the claimer publishes its group: ONE atomic per (segment, tile) — set the tile's bit and add the
    // group size (bits are set once each, so the add never carries into the count) — and the group's size in the segment's
    // per-tile row.  Nothing comes back: no dependent trip.  The claimer's own group is not published: its size travels in
    // the segment record, so a key that only one tile touches costs no atomic here and no bitmap traffic at all.
    if (khead && !claimed) {
        (void)__hip_atomic_fetch_add(&seg_mask[(size_t)d * FT_WORDS + (tile >> 5)], ((unsigned long long)eq_total << 32) | (1ull << (tile & 31)),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        W.tilerow[(size_t)d * FT_MAX_TILES + tile] = (uint16_t)eq_total;
    }
    lds_barrier();
    if (member) d = sd[head_tid];
    GB_STAMP(2);

    // ---- stage 3: exact key comparison (member: its head's key, same tile; head: the claimer's key), and the head's
    // request against the claimer's ----
    if (member) {
        bool keq;                                                  // (two LDS words instead of walking both keys' bytes in memory)
        if (len <= 16) {
            keq = slen[head_tid] == len;
            if (keq) { const ulonglong2 hk = hkw[head_tid]; keq = hk.x == k0 && hk.y == k1; }
        } else keq = req_key_equal_at(B, off, len, soff[head_tid], slen[head_tid]);
        if (!keq) my_flags |= SEG_RETRY;                           // two keys under one hash: careful round
    } else if (khead && d != g) {
        const uint32_t c_off = key_off_of(B, d), c_len = key_len_of(B, d, c_off);
        const Req cq = load_req_nogreg(B, d);
        if (!req_key_equal_at(B, off, len, c_off, c_len)) my_flags |= SEG_RETRY;   // two keys under one claim fingerprint
        my_flags |= req_diff_flags(B, g, d, tile_get(sreq, tid), cq, soft_leaky);
    }
    GB_STAMP(3);

    // ---- stage 4 (heads): finish the directory probe, verify the stored key, snapshot ------------------------------
    const bool resolve = khead && !W.careful && claimed;             // (a claimer saw its home cell free or foreign: its table lines are here)
    if (resolve) {
        const unsigned long long tag = gk;
        const uint32_t home = (uint32_t)pos;
        for (uint32_t step = 0; step < T.max_probe; ++step, pos = (pos + 1) & T.mask) {
            ulonglong2 de = de0;
            if (step) de = *(const ulonglong2*)&T.dir[pos];
            unsigned long long t = de.x, m = de.y;
            if (t == 0ull) {
                const unsigned long long old = atomicCAS(&T.dir[pos].tag, 0ull, tag);
                if (old == 0ull) {                                   // new key: this thread inserts it
                    slot = (uint32_t)pos; inserted = 1; cand = true;
                    if (!key_store(T, pos, key, len)) { errcode = 6; cand = false; }
                    break;
                }
                t = old;
                m = ld_agent(&T.dir[pos].meta);
            }
            if (t == tag) { slot = (uint32_t)pos; cand = true; ready = (m & META_READY) != 0; break; }
        }
        if (!cand && !errcode) errcode = 6;                          // probe bound exceeded: table full
        if (cand && slot != home) {
            const Bucket* bk = &T.buckets[slot];
            if (ready) { const uint4* cw = (const uint4*)&bk->cell; c0 = cw[0]; c1 = cw[1]; c2 = cw[2]; c3 = cw[3]; }
            rec = bk->rec;                                           // (zero for a bucket never used)
        }
        if (cand && ready) {
            const uint64_t cell[8] = {((uint64_t)c0.y << 32) | c0.x, ((uint64_t)c0.w << 32) | c0.z,
                                      ((uint64_t)c1.y << 32) | c1.x, ((uint64_t)c1.w << 32) | c1.z,
                                      ((uint64_t)c2.y << 32) | c2.x, ((uint64_t)c2.w << 32) | c2.z,
                                      ((uint64_t)c3.y << 32) | c3.x, ((uint64_t)c3.w << 32) | c3.z};
            bool eq = (uint32_t)(cell[7] >> 48) == len;
            if (eq) {
                if (len <= 16) {
                    // the key's two zero-padded words are in registers since the hash (cells are stored zero-padded: key_store):
                    // no second trip to the key bytes
                    eq = cell[0] == k0 && cell[1] == k1;
                } else if (len <= INLINE_KEY) {
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
            // the claim was issued before this comparison.  A mismatch = a 64-bit hash collision with a resident key:
            // the whole segment is answered RETRY and re-run in careful mode (verify first, claim by slot).
            if (!eq) my_flags |= SEG_RETRY;
        }
        // an entry that is not READY was inserted during this launch; its key bytes may still be in flight, but every
        // head — the inserter included — has compared its key with the claimer's request, so the key that ends up
        // stored is provably the segment's key.
    }
    GB_STAMPW(4);
    if (khead) sslot[tid] = resolve || W.careful ? slot : 0xffffffffu;
    if (soft_leaky) soft_any = 1u;
    if (inserted) ins_any = 1u;
    uint8_t rf = 0;
    if (valid) {
        if (errcode) { my_flags |= SEG_ERR | (errcode << 8); rf = RF_ERR; }
        if (inserted) { rf |= RF_INSERTED; W.slot[g] = slot; }
        if (claimed && !errcode) {
            // the claimer writes the segment record: the bucket before the batch, its slot, the size of the claimer's own
            // group — 56 bytes; the flags word next to them is only ever touched by seg_raise
            SegRec* sr = &W.srec[d];
            ulonglong2* q = (ulonglong2*)sr;
            q[0] = make_ulonglong2((unsigned long long)rec.limit, (unsigned long long)rec.duration);
            q[1] = make_ulonglong2((unsigned long long)rec.remaining, (unsigned long long)rec.stamp);
            q[2] = make_ulonglong2((unsigned long long)rec.burst, (unsigned long long)rec.expire_at);
            *(unsigned long long*)&sr->smeta = (unsigned long long)pack_smeta(rec, eq_total) | ((unsigned long long)slot << 32);
            if (rec.invalid_at != 0) W.sinv[d] = rec.invalid_at;
        }
        W.did[g] = pack_dl(d, head_tid, eq_before);
        W.rflags[g] = rf;
    }
    lds_barrier();
    if (soft_any) {                                                  // rare: requests of a leaky key stamped differently
        if (soft_leaky) {
            const uint32_t bs = sslot[head_tid];
            if (bs == 0xffffffffu) my_flags |= SEG_NONUNIFORM;       // a group that did not resolve its bucket (not the claimer's): the exact serial walk
            else {
                const Rec cur = T.buckets[bs].rec;
                my_flags |= leaky_created_harmless(cur, tile_get(sreq, tid), B.now_ms) ? SEG_CREATED_DIFFERS : SEG_NONUNIFORM;   // (declines GREGORIAN requests)
            }
        }
    }
    if (my_flags) seg_raise(&W.srec[d], e16, my_flags);
    if (ins_any) {                                                   // new keys in this tile (rare in steady state)
        const int ins = block_sum_lds(inserted, red);
        if (tid == 0 && ins) atomicAdd(&T.ctr->tags_used, (unsigned long long)ins);
    }
    GB_STAMPW(5);
}

__global__ __launch_bounds__(FT) void k_front(Table T, BatchView B, Work W) { front_body(T, B, W, blockIdx.x); }

struct EvalArgs { Table T; BatchView B; ResultView R; Work W; };

#ifndef GUBER_EVAL2_WAVES
#define GUBER_EVAL2_WAVES 4      // waves per SIMD the register allocation must allow (<= 128 VGPRs): 4 co-resident workgroups per CU
#endif
__device__ __forceinline__ void eval2_body(const EvalArgs& A, const uint32_t tile, const uint32_t ntiles) {
    const Table& T = A.T; const BatchView& B = A.B; const ResultView& R = A.R; const Work& W = A.W;
    __shared__ unsigned long long cnt[4];
    const uint32_t i = tile * 256 + threadIdx.x;
    unsigned long long* seg_mask = W.seg_tilemask + (size_t)W.parity * B.n_cap * FT_WORDS;
    const uint32_t e16 = W.epoch16;
    GB_STAMP2(0);
    {   // zero this batch's claim cells for the next batch: the part of the claim table the batch used (4 cells per request),
        // as whole sectors written once each — cheaper than one 8-byte store into a random sector per claim
        ulonglong2* cz = (ulonglong2*)W.claims;
        const uint32_t pairs = (W.cmask + 1u) >> 1;
        for (uint32_t j = i; j < pairs; j += ntiles * 256) cz[j] = make_ulonglong2(0ull, 0ull);
    }
    {   // clear, for the next batch, what the previous batch's publishers added to the other copy of the tile bitmaps: a
        // publisher = the head of a (segment, tile) group that is not the segment's claimer; it zeroes the word it added to
        unsigned long long* om = W.seg_tilemask + (size_t)(W.parity ^ 1u) * B.n_cap * FT_WORDS;
        for (uint32_t j = i; j < W.clear_n; j += ntiles * 256) {
            const uint32_t pd = W.did_prev[j];
            if ((pd & 0xffu) == 0u && (pd >> 16) != j) om[(size_t)(pd >> 16) * FT_WORDS + (j / FT >> 5)] = 0ull;
        }
    }
    // pre-pass: the head of every (segment, tile) group computes the group's base = members of the segment in earlier
    // tiles, and the segment's total, from the claimer's group (segment record), the bitmap and the per-tile counts of the
    // published groups; the other members pick both up from LDS (eval workgroup == tile, FT == 256).
    __shared__ uint32_t sbase[FT], stotal[FT];
    const bool live = i < B.n;
    const uint32_t dl = live ? W.did[i] : 0u;
    const uint32_t d = dl >> 16, lr = dl & 0xffffu;
    // everything that depends only on (i, d) is requested now, so that these loads are in flight together
    // with the heads' bitmap loads below instead of after the barrier
    uint32_t sf = 0, slot = 0, smeta = 0; uint8_t rf = 0; Req r; Rec s0;
    rec_clear(s0);
    if (live) {
        rf = W.rflags[i];
        r = load_req_nogreg(B, i);
        const ulonglong2* q = (const ulonglong2*)&W.srec[d];             // one 64-byte sector: bucket, slot, flags
        const ulonglong2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        s0.limit = (int64_t)q0.x; s0.duration = (int64_t)q0.y; s0.remaining = (int64_t)q1.x; s0.stamp = (int64_t)q1.y;
        s0.burst = (int64_t)q2.x; s0.expire_at = (int64_t)q2.y;
        smeta = (uint32_t)q3.x; slot = (uint32_t)(q3.x >> 32);
        s0.meta = smeta_meta(smeta);
        sf = seg_flags_of(q3.y, e16);
    }
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
        const uint32_t ct = d / FT, cc = smeta_group(smeta);               // the claimer's tile and its group (not in the bitmap)
        uint32_t base = ct < t ? cc : 0u, total = cc, below = 0, wcount = 0, wbits = 0;
#pragma unroll
        for (uint32_t w = 0; w < FT_WORDS; ++w) {
            const uint32_t c = (uint32_t)(sw[w] >> 32);
            total += c;
            base += w < mw ? c : 0u;
            if (w == mw) { wbits = (uint32_t)sw[w]; wcount = c; below = wbits & ((1u << mb) - 1u); }
        }
        if (below) {
            if ((wbits & (wbits - 1u)) == 0u) {
                base += wcount;                                            // the word holds one published tile: the count is its
            } else {
                // several published tiles in my 32-tile word: their per-tile counts (u16, 64 bytes), masked by the bitmap
                const uint4* r4 = (const uint4*)(W.tilerow + (size_t)d * FT_MAX_TILES + mw * 32);
                uint4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = r4[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t w4[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        base += ((below >> (q * 8 + e * 2)) & 1u) ? (w4[e] & 0xffffu) : 0u;
                        base += ((below >> (q * 8 + e * 2 + 1)) & 1u) ? (w4[e] >> 16) : 0u;
                    }
                }
            }
        }
        sbase[threadIdx.x] = base; stotal[threadIdx.x] = total;
    }
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0ull;
    GB_STAMP2(1);
    lds_barrier();
    GB_STAMP2(2);
    int c_over = 0, c_hit = 0, c_miss = 0, c_size = 0;
    // the wave in which every request is the common case runs straight through the closed form (as k_eval3's does,
    // guber_kernels_part.h: one ballot instead of the general path's cascade of divergent sections; the same steps, so the same results)
    const bool plain = !live || (!(rf & RF_INSERTED) && sf == 0u && !(smeta & SM_HAS_INVALID) && token_fast_ok(s0, r, B.now_ms));
    const bool plain_wave = !W.store_flags && !T.gpend && __ballot(!plain) == 0ull;
    if (plain_wave) {
        if (live) {
            const uint32_t total = stotal[lr >> 8], rank = sbase[lr >> 8] + (lr & 0xffu);
            Rec after; Resp out;
            const uint32_t ev = token_fast(s0, r, rank, out, after);
            store_resp(R, i, out);
            c_over = (ev & EV_OVER) ? 1 : 0; c_hit = (ev & EV_HIT) ? 1 : 0; c_miss = (ev & EV_MISS) ? 1 : 0;
            if (rank == total - 1) {
                rec_set_stamp(after, W.touch + i);                    // the key's place in the recency order: its last request (lrucache.go:111-128)
                T.buckets[slot].rec = after;
                c_size = (int)(rec_kind(after) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
            }
        }
    } else if (live) {
        if (rf & RF_INSERTED) atomicOr(&T.dir[W.slot[i]].meta, META_READY);   // publish this batch's inserts
        if (sf & SEG_ERR) {
            store_err(R, i, (uint8_t)(sf >> 8));
        } else if (sf & SEG_RETRY) {
            store_err(R, i, IE_RETRY);
            atomicAdd(&T.ctr->retries, 1ull);
        } else {
            if (smeta & SM_HAS_INVALID) s0.invalid_at = W.sinv[d];
            const uint32_t base = sbase[lr >> 8], total = stotal[lr >> 8];
            const uint32_t rank = base + (lr & 0xffu);
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
            // The requests real traffic consists of — a live bucket, a request that does not reconfigure it — are answered by
            // the closed forms of guber_algo.h (a few dozen instructions).  Everything else goes through ONE inlined
            // apply() site that serves both the rank-stepping of a uniform run (eval_uniform_rank_1x) and the serial walk
            // of a heterogeneous segment, so the kernel carries the general state machine once, not four times.
            Rec after; Resp out;
            uint32_t ev = 0;
            bool done = false;
            if (parallel) {
                if (token_fast_ok(s0, r, B.now_ms)) { ev = token_fast(s0, r, rank, out, after); done = true; }
                else if (leaky_fast(s0, r, B.now_ms, rank, out, after, ev)) done = true;
            }
            const bool walk = !parallel && rank == 0;
            uint32_t lastj = walk ? 0xffffffffu : i;                     // the run's last request (walk: the last one walked that reached the cache)
            if ((parallel && !done) || walk) {
                Req cur = r;
                if (parallel) {                                          // the calendar values are loaded only here
                    if (B.greg_expire && B.greg_duration) { cur.greg_expire = B.greg_expire[i]; cur.greg_duration = B.greg_duration[i]; }
                    else if (cur.behavior & BH_GREGORIAN) greg_fill(B.now_ms, cur.duration, cur.greg_expire, cur.greg_duration, guber_tz());
                }
                after = s0;
                uint64_t k = rank;
                Rec prev2; rec_clear(prev2);
                bool have_prev2 = false;
                // walk iterator: tiles holding the segment in order (bitmap), inside a tile the packed words whose id is d
                uint32_t wv = 0, mm = 0, tt = 0, q = FT;
                for (;;) {
                    uint32_t j = i;
                    if (walk) {
                        bool found = false, end = false;
                        while (!found && !end) {
                            if (q < FT && tt * FT + q < B.n) {
                                const uint32_t id = W.did[(size_t)tt * FT + q];
                                if ((id >> 16) == d) { j = tt * FT + q; found = true; }
                                q++;
                            } else {
                                while (mm == 0u && wv < FT_WORDS) {
                                    mm = (uint32_t)seg_mask[(size_t)d * FT_WORDS + wv];
                                    if (wv == (d / FT >> 5)) mm |= 1u << (d / FT & 31);      // the claimer's tile is not published
                                    tt = wv * 32; wv++;
                                }
                                if (mm == 0u) end = true;
                                else { const uint32_t bpos = (uint32_t)__ffs((int)mm) - 1u; mm &= mm - 1u; tt = (tt & ~31u) + bpos; q = 0; }
                            }
                        }
                        if (end) break;
                        cur = load_req(B, j);
                        if (cur.algorithm <= ALGO_LEAKY) lastj = j;
                    }
                    const Rec before = after;
                    const uint32_t e1 = apply(after, cur, B.now_ms, out);
                    if (walk) {
                        store_resp(R, j, out);
                        store_events(W, j, e1, after);
                        if (out.err == 0) queue_global(T, slot, cur, 1);
                        c_over += (e1 & EV_OVER) ? 1 : 0; c_hit += (e1 & EV_HIT) ? 1 : 0; c_miss += (e1 & EV_MISS) ? 1 : 0;
                        continue;
                    }
                    if (k == 0) { ev = e1; break; }
                    k--;
                    if (k == 0) continue;
                    if (rec_eq(after, before)) { k = 0; continue; }                       // fixed point
                    if (have_prev2 && rec_eq(after, prev2)) {                             // period 2
                        if (k & 1) after = before;
                        k = 0;
                        continue;
                    }
                    prev2 = before; have_prev2 = true;
                    if (pure_subtract(before, after, cur, B.now_ms)) {
                        const uint32_t kind = rec_kind(after);
                        const int64_t n = kind == K_TOKEN ? after.remaining : go_f2i(bits2f(after.remaining));
                        if (n > 0) {
                            const uint64_t m = (uint64_t)(n - 1) / (uint64_t)cur.hits;
                            const uint64_t jj = m < k ? m : k;
                            if (jj > 0) {
                                const int64_t dec = (int64_t)(jj * (uint64_t)cur.hits);   // <= n-1, exact
                                if (kind == K_TOKEN) after.remaining -= dec;
                                else after.remaining = f2bits(bits2f(after.remaining) - (double)dec);
                                k -= jj;
                                have_prev2 = false;
                            }
                        }
                    }
                }
            }
            if (parallel) {
                store_resp(R, i, out);
                store_events(W, i, ev, after);
                c_over = (ev & EV_OVER) ? 1 : 0; c_hit = (ev & EV_HIT) ? 1 : 0; c_miss = (ev & EV_MISS) ? 1 : 0;
            }
            // (a request with an invalid algorithm never reaches the cache — workers.go:317-321 rejects it before tokenBucket / leakyBucket call
            // GetItem — so it does not move its key in the recency order: a run of such requests writes nothing, a walked segment is
            // stamped with its last request that did reach the cache)
            if ((parallel && rank == total - 1 && r.algorithm <= ALGO_LEAKY) || (walk && lastj != 0xffffffffu)) {
                rec_set_stamp(after, W.touch + lastj);                // the key's place in the recency order: its last request (lrucache.go:111-128)
                T.buckets[slot].rec = after;
                c_size = (int)(rec_kind(after) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
                if (parallel && out.err == 0) queue_global(T, slot, r, (uint64_t)rank + 1);
            }
        }
    }
    GB_STAMP2(3);
    // event counters of the workgroup: wave-level sums, one LDS atomic per wave and counter, one barrier
    {
        const int w_over = wave_sum(c_over), w_hit = wave_sum(c_hit), w_miss = wave_sum(c_miss), w_size = wave_sum(c_size);
        if ((threadIdx.x & 63) == 0 && (w_over | w_hit | w_miss | w_size)) {
            if (w_over) atomicAdd(&cnt[0], (unsigned long long)w_over);
            if (w_hit) atomicAdd(&cnt[1], (unsigned long long)w_hit);
            if (w_miss) atomicAdd(&cnt[2], (unsigned long long)w_miss);
            if (w_size) atomicAdd(&cnt[3], (unsigned long long)(long long)w_size);
        }
        lds_barrier();
        if (threadIdx.x == 0 && (cnt[0] | cnt[1] | cnt[2] | cnt[3])) {
            BlockCounters* bc = &T.bctr[tile];
            bc->over += cnt[0]; bc->hits += cnt[1]; bc->misses += cnt[2]; bc->size_delta += (long long)cnt[3];
        }
    }
    GB_STAMP2(4);
}
// (the arguments are read through the kernel-argument pointer, as k_eval2_multi does: preloading all of them into scalar registers
// cost this kernel four spilled vector registers and a scratch allocation)
__global__ __launch_bounds__(256, GUBER_EVAL2_WAVES) void k_eval2(EvalArgs A) {
    const EvalArgs* a = (const EvalArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    eval2_body(*a, blockIdx.x, gridDim.x);
}

// ---- several engines in one launch ------------------------------------------------------------------------------------
// The logical shards of a GPU (one table each, disjoint keys) have nothing to order between them, and one batch's two
// launches leave most of the chip idle (256 workgroups, one per CU, waiting on dependent memory trips).  A dispatcher that
// has batches for several shards waiting puts up to MULTI_MAX of them — one per engine — into ONE k_front_multi and ONE
// k_eval2_multi: workgroup -> (batch, tile) by a prefix table, every workgroup then runs exactly the single-batch body on
// its engine's table and work arrays.  Same results by construction; what changes is that the batches' memory trips
// overlap inside one launch instead of across streams (where every kernel boundary of every stream costs the others:
// profiles/archive/r02_m_shard_streams.txt).
#ifndef GUBER_MULTI_MAX
#define GUBER_MULTI_MAX 4        // tables per fused launch (the kernel-argument segment holds at most 7: static_assert below)
#endif
constexpr int MULTI_MAX = GUBER_MULTI_MAX;
struct FrontArgs { Table T; BatchView B; Work W; };
// (end_tile[k] = where batch k's workgroups end; entries past the last batch stay at UINT32_MAX — what multi_batch_of relies on)
struct MultiFront {
    uint32_t nb; uint32_t end_tile[MULTI_MAX]; FrontArgs sub[MULTI_MAX];
    MultiFront() { memset((void*)this, 0, sizeof *this); for (int k = 0; k < MULTI_MAX; ++k) end_tile[k] = 0xffffffffu; }
};
struct MultiEval {
    uint32_t nb; uint32_t end_tile[MULTI_MAX]; EvalArgs sub[MULTI_MAX];
    MultiEval() { memset((void*)this, 0, sizeof *this); for (int k = 0; k < MULTI_MAX; ++k) end_tile[k] = 0xffffffffu; }
};
static_assert(sizeof(MultiFront) <= 4096 && sizeof(MultiEval) <= 4096, "kernel arguments are limited to 4 KB");

// which batch of a fused launch a workgroup belongs to: the number of batches that end at or before it (the ends ascend; unused
// entries are UINT32_MAX), without a branch — `ends` is read through the kernel-argument pointer: one wide scalar load and a dozen
// scalar instructions (the chain of conditional updates it replaces was 25-30 with three branches, in every wave of every fused launch)
template <int N>
__device__ __forceinline__ uint32_t multi_batch_of(const uint32_t* ends, uint32_t wg, uint32_t& first) {
    uint32_t e[N];
#pragma unroll
    for (int k = 0; k < N - 1; ++k) e[k] = ends[k];
    uint32_t sb = 0;
#pragma unroll
    for (int k = 0; k < N - 1; ++k) sb += wg >= e[k] ? 1u : 0u;
    first = 0u;
#pragma unroll
    for (int k = 0; k < N - 1; ++k) first = sb == (uint32_t)(k + 1) ? e[k] : first;
    return sb;
}
__global__ __launch_bounds__(FT) void k_front_multi(MultiFront A) {
    const MultiFront* m = (const MultiFront*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t first;
    const uint32_t sb = multi_batch_of<MULTI_MAX>(m->end_tile, blockIdx.x, first);
    const FrontArgs* a = (const FrontArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiFront, sub)) + sb;
    front_body(a->T, a->B, a->W, blockIdx.x - first);
}
__global__ __launch_bounds__(256, GUBER_EVAL2_WAVES) void k_eval2_multi(MultiEval A) {
    const MultiEval* m = (const MultiEval*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t first;
    const uint32_t sb = multi_batch_of<MULTI_MAX>(m->end_tile, blockIdx.x, first);
    const EvalArgs* a = (const EvalArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiEval, sub)) + sb;
    eval2_body(*a, blockIdx.x - first, m->end_tile[sb] - first);
}


// The same two launches for up to MULTI_MEM_MAX batches (a pool dispatcher's whole generation: every shard of a device plus its
// GLOBAL engine): the argument blocks do not fit the 4 KB kernel-argument segment, so they live in device memory — the copy
// kernel that brings the stages' request columns into HBM brings them along (guber_engine.hip launch_stage_group) — and are
// read like kernel arguments: uniform loads from the constant address space (scalar loads, nothing held in vector registers).
constexpr int MULTI_MEM_MAX = 16;
struct MultiFrontMem { uint32_t nb; uint32_t end_tile[MULTI_MEM_MAX]; uint32_t pad_[15]; FrontArgs sub[MULTI_MEM_MAX]; };
struct MultiEvalMem { uint32_t nb; uint32_t end_tile[MULTI_MEM_MAX]; uint32_t pad_[15]; EvalArgs sub[MULTI_MEM_MAX]; };
struct MultiArgsMem { MultiFrontMem F; MultiEvalMem E; };
typedef const __attribute__((address_space(4))) char* const_mem_t;
__device__ __forceinline__ uint32_t multi_mem_batch(const uint32_t* end_tile, uint32_t nb, uint32_t wg, uint32_t& first) {
    uint32_t sb = 0; first = 0;
    for (uint32_t k = 0; k + 1 < nb; ++k)
        if (wg >= end_tile[k]) { first = end_tile[k]; sb = k + 1; }
    return sb;
}
__global__ __launch_bounds__(FT) void k_front_multi_mem(const MultiFrontMem* Ag) {
    const MultiFrontMem* A = (const MultiFrontMem*)(const_mem_t)(uintptr_t)Ag;
    uint32_t first;
    const uint32_t sb = multi_mem_batch(A->end_tile, A->nb, blockIdx.x, first);
    const FrontArgs* a = A->sub + sb;
    front_body(a->T, a->B, a->W, blockIdx.x - first);
}
__global__ __launch_bounds__(256, GUBER_EVAL2_WAVES) void k_eval2_multi_mem(const MultiEvalMem* Ag) {
    const MultiEvalMem* A = (const MultiEvalMem*)(const_mem_t)(uintptr_t)Ag;
    uint32_t first;
    const uint32_t sb = multi_mem_batch(A->end_tile, A->nb, blockIdx.x, first);
    eval2_body(A->sub[sb], blockIdx.x - first, A->end_tile[sb] - first);
}

}  // namespace guber

#include "guber_kernels_part.h"
#include "guber_kernels_route.h"
#include "guber_kernels_front.h"
#ifndef GUBER_KERNELS_PIPELINES_ONLY   // (the host emulation of the batch pipelines, tests/hostsim/devsim.cpp, stops here)
#include "guber_kernels_ops.h"
#include "guber_kernels_small.h"
#endif
