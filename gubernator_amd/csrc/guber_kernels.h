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

}  // namespace guber

#include "guber_kernels_ops.h"
