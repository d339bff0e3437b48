// guber_kernels_part.h — the owner-partitioned pipeline for batches of 257 .. 65 536 requests: THREE launches, every key of a
// batch handled by exactly ONE workgroup, no atomics on device memory in steady state.  (Included at the end of guber_kernels.h.)
//
// Why.  The two-launch pipeline (k_front / k_eval2) lets every tile of 256 requests resolve its own keys and coordinates the
// tiles that share a key through a per-batch claim table: one device-scope CAS per (key, tile) group, one publish atomic and one
// per-tile count for every group that is not the key's first, three speculative table sectors per group head, and 2 MB of claim
// cells zeroed again per batch — 67.8 k atomics and 5.7 MB of coordination writes per 65 536 requests (profiles/r03_*), on a
// path whose rate follows the number of memory transactions per request.  Here the coordination is a partition instead:
//
//   k_part  (request order, one workgroup per tile of 256 requests): key -> XXH64, the tile grouped by hash in LDS exactly as
//           k_front does, members compared with their group's head (fields from LDS, key bytes from L1).  Per (key, tile) GROUP
//           one 64-byte MESSAGE {hash, key bytes (<= 16 inline), hits, limit, duration, burst, packed rest} written into the
//           tile's own region, sorted (LDS counting sort) by the key's OWNER = the top 8 bits of its home position in the table,
//           plus 256 x {start, count} per tile.  Per request one packed word (group, rank inside the group).  Streaming writes only.
//   k_own   (one workgroup per owner, 256 per batch; thread t gathers tile t's run for this owner): the owner's messages in tile
//           order = request order.  Grouped by hash in LDS; per key the exact comparison of every group's key bytes and request
//           shape against the key's first-seen group (from the messages: no gathers from the request columns), the rank base of
//           every group (requests of the key in earlier tiles: per 64-message chunk a wave scan per multi-group key + per-chunk
//           sums) and the total; ONE table lookup per key (directory entry || home bucket in one trip, insert, exact key
//           verification) by the only workgroup that can touch this key in this batch: no claim, no CAS except the insert of a
//           new key's tag.  Per group one 64-byte RECORD back into the tile's region: the bucket as it was before the batch,
//           slot, flags, base, total.  An owner touches one 1/256 of the table (translation locality).
//   k_eval3 (request order): packed word -> the group's record -> every request evaluates its own rank exactly as k_eval2 does
//           (closed forms / one apply() site / serial walk of heterogeneous segments); the last request writes the bucket.
//
// Request order semantics (gubernator.go:203 serial loop, workers.go:190-258 one goroutine per worker) hold as before: the rank
// of a request inside its key's segment is (requests of the key in earlier tiles) + (rank inside its tile's group).
// What this pipeline does not handle itself it reports per item as before (GUBER_ITEM_E_RETRY: two keys under one 64-bit hash;
// the engine re-runs those through the careful round of the two-launch pipeline).
#pragma once

namespace guber {

constexpr int PT_PARTS = 256;            // k_own workgroups launched per batch = the largest number of owners
// HOW MANY OWNERS a batch is split over follows the traffic, on the device, batch by batch (Work::pmode, four words per engine):
// 128 (7 bits) or 256 (8 bits).  With 256 a Zipf batch gives a k_own workgroup 175 messages / 112 keys — half of its lanes idle for
// the whole residency; with 128 it gets 350 / 225 and the pipeline is 8 % faster (profiles/r04_z_owners_ab.txt).  With 128 a batch
// of uniform keys brings 512 keys per owner: every round splits after a wasted gather and the pipeline is 46 % slower.  So: k_own
// counts the rounds that split for their number of keys (pmode[2]); the batch's k_eval3 — after every k_own workgroup, before the
// next batch's k_part — moves to 8 bits for PM_HOLD batches when PM_SPLITS or more did, and back to 7 afterwards (a probe: one slower
// batch in PM_HOLD + 1 if the traffic is still uniform).  pmode[3] != 0 pins the mode (GUBER_PT_BITS=7|8, tests).
constexpr uint32_t PM_SPLITS = 8, PM_HOLD = 255;
GB_HD uint32_t pm_bits(const uint32_t* pm) { return pm[0] == 7u ? 7u : 8u; }
// GUBER_FUSE_EP (k_evalpart_multi below: batch b's k_eval3 and batch b + 1's k_part in ONE launch): the word k_eval3 decides in is
// being written while the next batch's k_part runs, so a batch of such an engine reads its owner count from a slot of its own
// parity instead (Work::pmslot = 1 + (batch & 1) -> pmode[3 + pmslot]; 0 = the word itself), and its k_eval3 leaves the decision
// there for the batch after the next: the same rule, taking effect one batch later.
GB_HD uint32_t pm_bits_of(const Work& W) { return (W.pmslot ? W.pmode[3u + W.pmslot] : W.pmode[0]) == 7u ? 7u : 8u; }

// one (key, tile) group, tile -> owner
struct alignas(64) GMsg {
    unsigned long long hash;             // XXH64(key) & hash_mask, 0 -> 1
    unsigned long long key0, key1;       // key bytes (length <= 16, zero padded) — or, G_LONG: key0 = key_off | len << 32 of the head's key
    long long hits, limit, duration, burst;
    unsigned long long misc;             // gm_*: head thread 8 | members-1 8 | G_* 8 | key length 5 | behavior 6 | algorithm 2 | owner 1 | created_at: min 18 (ms from the batch clock, signed), span 8
};
static_assert(sizeof(GMsg) == 64, "one message = one 64-byte sector");
enum : uint32_t { G_NONUNIFORM = 1, G_RETRY = 2, G_CREATED = 4, G_CFAR = 8, G_ODD = 16, G_LONG = 32 };
//   G_NONUNIFORM  members of the group differ in a request field other than created_at
//   G_RETRY       members of the group differ in their key bytes (one hash, two keys)
//   G_CREATED     members differ in created_at only (the range travels in misc)
//   G_CFAR        the created_at range does not fit misc (further than +-131 s from the batch clock, or wider than 255 ms)
//   G_ODD         a request the packed shape cannot carry exactly (behavior bits above 5, an algorithm other than 0 / 1, calendar
//                 values precomputed by the host): equal to nothing but the members of its own group
//   G_LONG        key longer than 16 bytes: compared through the request's key bytes in memory
//
// The record of a group is 32 bytes {remaining, stamp, expire_at, packed: slot 26 | total 16 | base 16 | burst is zero | algorithm |
// status | kind | 1} whenever the rest of the bucket is what the request itself says (stored limit / duration / burst equal to the
// request's — the steady state —, or the key is new), nothing is flagged, no error, no InvalidAt; every other group gets {.., 0}
// there and the 64-byte record beside it (grec[]).  Two 32-byte records of neighbouring groups — owners of one XCD are neighbours
// in a tile's region — share a sector in that XCD's L2 and leave it as one write; k_eval3 reads half the bytes (+2.3 % on one box,
// profiles/r04_w_forms_ab.txt; 32-byte MESSAGES were measured too: +0.6 %, not kept).
struct alignas(32) GRecS { int64_t remaining, stamp, expire_at; unsigned long long pk; };
static_assert(sizeof(GRecS) == 32, "two per sector");
enum : uint32_t { SM_COMPACT_OK = 1u << 24, SM_BURST_ZERO = 1u << 25 };      // GRec::smeta, k_own only: the group's record has the 32-byte form
GB_HD unsigned long long grs_pack(uint32_t kind, uint32_t status, uint32_t algo, bool burst_zero, uint32_t base, uint32_t total, uint32_t slot) {
    return 1ull | ((unsigned long long)(kind & 3u) << 1) | ((unsigned long long)(status & 1u) << 3) | ((unsigned long long)(algo & 1u) << 4) |
           ((unsigned long long)(burst_zero ? 1u : 0u) << 5) | ((unsigned long long)(base & 0xffffu) << 6) |
           ((unsigned long long)((total - 1u) & 0xffffu) << 22) | ((unsigned long long)slot << 38);
}
GB_HD uint32_t gm_head(unsigned long long m) { return (uint32_t)m & 0xffu; }
GB_HD uint32_t gm_cnt(unsigned long long m) { return (((uint32_t)m >> 8) & 0xffu) + 1u; }
GB_HD uint32_t gm_flags(unsigned long long m) { return ((uint32_t)m >> 16) & 0xffu; }
GB_HD uint32_t gm_klen(unsigned long long m) { return ((uint32_t)m >> 24) & 31u; }
GB_HD uint32_t gm_shape(unsigned long long m) { return (uint32_t)(m >> 29) & 0x1ffu; }       // behavior 6 | algorithm 2 | is_owner 1
GB_HD uint32_t gm_algo(unsigned long long m) { return (uint32_t)(m >> 35) & 3u; }
GB_HD uint32_t gm_behavior(unsigned long long m) { return (uint32_t)(m >> 29) & 63u; }
GB_HD int64_t gm_cmin_delta(unsigned long long m) { return (int64_t)((int64_t)(m << 8) >> 46); }   // bits 38..55, sign-extended
GB_HD uint32_t gm_cspan(unsigned long long m) { return (uint32_t)(m >> 56); }
GB_HD unsigned long long gm_pack(uint32_t head, uint32_t cnt, uint32_t flags, uint32_t klen, uint32_t shape, int64_t cmin_delta, uint32_t span) {
    return (unsigned long long)(head & 0xffu) | ((unsigned long long)((cnt - 1u) & 0xffu) << 8) | ((unsigned long long)(flags & 0xffu) << 16) |
           ((unsigned long long)(klen & 31u) << 24) | ((unsigned long long)(shape & 0x1ffu) << 29) |
           (((unsigned long long)cmin_delta & 0x3ffffull) << 38) | ((unsigned long long)(span & 0xffu) << 56);
}

// one (key, tile) group, owner -> tile: everything k_eval3 needs for the group's requests in ONE sector
struct alignas(64) GRec {
    int64_t limit, duration, remaining, stamp, burst, expire_at;     // the bucket as it was before the batch (guber::Rec)
    uint32_t smeta;                      // pack_smeta: kind | status | SM_HAS_INVALID | algorithm
    uint32_t slot;                       // where the bucket lives
    unsigned long long tail;             // SEG_* 8 | item error code 8 | base 16 (requests of the key in earlier tiles) | total-1 16 | segment id 16
};
static_assert(sizeof(GRec) == 64, "one record = one 64-byte sector");
GB_HD unsigned long long gr_tail(uint32_t sf, uint32_t err, uint32_t base, uint32_t total, uint32_t seg) {
    return (unsigned long long)(sf & 0xffu) | ((unsigned long long)(err & 0xffu) << 8) | ((unsigned long long)(base & 0xffffu) << 16) |
           ((unsigned long long)((total - 1u) & 0xffffu) << 32) | ((unsigned long long)(seg & 0xffffu) << 48);
}

// per-request word k_part -> k_eval3: group (position of its message in the tile's region) 8 | rank inside the group 8 | item error 8 | valid << 31
GB_HD uint32_t pd_pack(uint32_t j, uint32_t rank, uint32_t err) { return (j & 0xffu) | ((rank & 0xffu) << 8) | ((err & 0xffu) << 16) | 0x80000000u; }

// owner of a key: the top PT_PARTS bits of its home position — an owner's keys live in one contiguous 1/256 of the table
// (W.pshift = log2(slots) - 8: the shift for 256 owners)
__device__ __forceinline__ uint32_t owner_of(const Table& T, const Work& W, uint64_t h, uint32_t pbits) { return (uint32_t)(((h >> 7) & T.mask) >> (W.pshift + 8u - pbits)) & ((1u << pbits) - 1u); }
// launch order of the owners' workgroups is round-robin over the 8 XCDs: a tile writes the messages of the owners that share an
// XCD next to each other, so that an L2 sees whole sectors of a tile's region
GB_HD uint32_t owner_order(uint32_t p, uint32_t pbits) { return ((p & 7u) << (pbits - 3u)) | (p >> 3); }
GB_HD uint32_t owner_from_order(uint32_t q, uint32_t pbits) { return ((q & ((1u << (pbits - 3u)) - 1u)) << 3) | (q >> (pbits - 3u)); }

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) { return (uint32_t)wave_incl_scan_i32((int)v); }   // (guber_table.h: DPP)

// ---- k_part ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void part_body(const Table& T, const BatchView& B, const Work& W, const uint32_t tile) {
    constexpr int GT_BITS = 9, GT = 1 << GT_BITS;
    __shared__ alignas(16) unsigned long long gkey[GT];       // the grouping's hash table; afterwards (same bytes) the heads' key words
    ulonglong2* const hkw = (ulonglong2*)gkey;                    // head thread -> {key bytes 0..7, 8..15} of a key of <= 16 bytes, zero padded
    static_assert(GT * 8 >= FT * 16, "the heads' key words fit where the hash table was");
    // the per-wave member bitmaps are read once, right after the grouping; the tile's request fields are needed from then on (members
    // against their head): one piece of LDS for both (27 KB per workgroup instead of 39: five workgroups per CU instead of four)
    __shared__ union GbitsOrReqs { unsigned long long gbits[FT / 64][GT]; TileReqs sreq; } gu;
    unsigned long long (&gbits)[FT / 64][GT] = gu.gbits;
    TileReqs& sreq = gu.sreq;
    __shared__ uint32_t sd[FT];                                   // head -> G_* raised by the group's members, then the position of its group's message in the tile's region
    __shared__ uint32_t soff[FT], slen[FT];
    __shared__ int gcmin[FT], gcmax[FT];                          // head -> created_at range of the group (ms from the batch clock, clamped to +-2^17)
    __shared__ uint32_t pc[FT];                                   // groups per owner (in owner_order), then where each owner's run starts
    __shared__ uint32_t wsum[FT / 64];
    uint32_t* const gfl = sd;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = tile * FT + tid;
    const bool valid = g < B.n;
    const uint32_t pbits = pm_bits_of(W);                         // owners of this batch: 1 << pbits
    GP_STAMP(0, 0);

    if (tile == 0 && W.snap_seq) {                                // a counter read-back rides on this launch (Work::snap_*)
        for (uint32_t k = tid; k < W.snap_n; k += FT) W.snap_b[k] = T.bctr[k];
        if (tid == 0) *W.snap_c = *T.ctr;
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();                                  // (measured in round 6: a relaxed stamp instead changes nothing, profiles/r06_snapshot_release_ab.txt)
            __hip_atomic_store(W.snap_stamp, W.snap_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (uint32_t j = tid; j < GT; j += FT) {
        gkey[j] = 0ull;
#pragma unroll
        for (int w = 0; w < FT / 64; ++w) gbits[w][j] = 0ull;
    }
    pc[tid] = 0u;
    // ---- key -> hash (as k_front: for fixed-width keys the key's words are requested together with the offsets) ----
    uint32_t errcode = 0, off = 0, len = 0;
    const uint8_t* key = nullptr;
    unsigned long long gk = 0ull;
    uint64_t h = 0;
    uint64_t kw[4] = {0, 0, 0, 0};
    Req mine;
    mine.hits = mine.limit = mine.duration = mine.burst = mine.created_at = 0; mine.greg_expire = mine.greg_duration = 0;
    mine.behavior = 0; mine.algorithm = 0; mine.is_owner = 0;
    if (valid) {
        uint32_t off_g = 0, len_g = 0;
        bool spec = false;
        if (!B.key_stride && !B.key_len && B.n >= 2) {
            const uint32_t o0 = B.key_off[0], o1 = B.key_off[1], oend = B.key_off[B.n];
            len_g = o1 - o0; off_g = o0 + g * len_g;
            if (len_g != 0 && len_g < 32 && (uint64_t)off_g + 32 <= (uint64_t)oend + 8) {   // (the buffer is padded by 8 bytes)
                spec = true;
                const uint8_t* kp = B.key_bytes + off_g;
                kw[0] = ld_key_word(kp); kw[1] = ld_key_word(kp + 8); kw[2] = ld_key_word(kp + 16); kw[3] = ld_key_word(kp + 24);
            }
        }
        mine = load_req_nogreg(B, g);
        off = key_off_of(B, g);
        len = key_len_of(B, g, off);
        key = B.key_bytes + off;
        if (len == 0) errcode = IE_EMPTY_KEY;
        else if (len > T.max_key) errcode = 7;
        if (!errcode) {
            if (!(spec && off == off_g && len == len_g)) {
                spec = false;
                if (len <= 16) { kw[0] = ld_key_word(key); kw[1] = len > 8 ? ld_key_word(key + 8) : 0ull; }
            }
            h = (spec ? xxhash64_words4(kw, len, 0) : xxhash64(key, len, 0)) & T.hash_mask;
            gk = h ? h : 1ull;
        }
    }
    soff[tid] = off; slen[tid] = len;
    GP_STAMPW(0, 1);
    lds_barrier();
    // ---- group the tile by hash (LDS table with per-wave member bitmaps, as k_front) ----
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
    const bool khead = valid && gk != 0ull && eq_before == 0;
    const bool member = valid && gk != 0ull && eq_before != 0;
    lds_barrier();                                                // every bitmap has been read: the requests' fields take their place
    if (valid) tile_put(sreq, tid, mine);
    // the key's bytes as the message carries them (<= 16 bytes: two zero-padded words).  A head leaves them where the hash table was:
    // its members compare their own words with them — two LDS words instead of walking both keys' bytes in memory
    unsigned long long k0 = kw[0], k1 = len > 8 ? kw[1] : 0ull;
    if (len < 8) k0 &= tail_mask(len); else if (len > 8 && len < 16) k1 &= tail_mask(len - 8);
    if (khead && len <= 16) hkw[tid] = make_ulonglong2(k0, k1);
    GP_STAMP(0, 2);
    // created_at as the messages carry it: milliseconds from the batch clock
    int cd = 0; bool cfar = false;
    {
        const int64_t d64 = wsub(mine.created_at, B.now_ms);
        cfar = d64 < -131072 || d64 > 131071;
        cd = cfar ? (d64 < 0 ? -131072 : 131071) : (int)d64;
    }
    // heads: the group's accumulators, and a place among the tile's messages for this owner
    uint32_t q = 0, qr = 0;
    if (khead) {
        uint32_t f = 0;
        if ((mine.behavior >> 6) != 0u || mine.algorithm > 1u || ((mine.behavior & BH_GREGORIAN) && B.greg_expire && B.greg_duration)) f |= G_ODD;
        if (len > 16) f |= G_LONG;
        if (cfar) f |= G_CFAR;
        gfl[tid] = f; gcmin[tid] = cd; gcmax[tid] = cd;
        q = owner_order(owner_of(T, W, h, pbits), pbits);
        qr = atomicAdd(&pc[q], 1u);
    }
    lds_barrier();
    // members: one key, one request shape per group — compared with the head on the exact key bytes and on every field
    if (member) {
        bool soft_leaky = false;
        uint32_t f = 0;
        const uint32_t df = req_diff_flags(B, g, tile * FT + head_tid, mine, tile_get(sreq, head_tid), soft_leaky);
        if (df & SEG_NONUNIFORM) f |= G_NONUNIFORM;
        if ((df & SEG_CREATED_DIFFERS) || soft_leaky) {
            f |= G_CREATED | (cfar ? G_CFAR : 0u);
            atomicMin(&gcmin[head_tid], cd); atomicMax(&gcmax[head_tid], cd);
        }
        bool keq;
        if (len <= 16) {
            keq = slen[head_tid] == len;
            if (keq) { const ulonglong2 hk = hkw[head_tid]; keq = hk.x == k0 && hk.y == k1; }
        } else keq = req_key_equal_at(B, off, len, soff[head_tid], slen[head_tid]);
        if (!keq) f |= G_RETRY;
        if (f) atomicOr(&gfl[head_tid], f);
    }
    GP_STAMPW(0, 3);
    // exclusive scan of the owners' group counts: where each owner's run starts in the tile's region
    {
        const uint32_t c = pc[tid];
        const uint32_t incl = wave_incl_scan_u32(c);
        if (lane == 63) wsum[wave] = incl;
        lds_barrier();
        uint32_t before = 0;
#pragma unroll
        for (uint32_t w = 0; w < FT / 64; ++w) before += w < wave ? wsum[w] : 0u;
        const uint32_t start = before + incl - c;
        pc[tid] = start;
        if (tid < (1u << pbits)) W.gse[(size_t)tile * PT_PARTS + owner_from_order(tid, pbits)] = start | (c << 16);
    }
    lds_barrier();
    GP_STAMP(0, 4);
    // heads: the message
    if (khead) {
        const uint32_t j = pc[q] + qr;
        uint32_t f = gfl[tid];
        sd[tid] = j;
        const int dmin = gcmin[tid];
        const uint32_t span = (uint32_t)(gcmax[tid] - dmin);
        if (span > 255u) f |= G_CFAR;
        const uint32_t shape = (mine.behavior & 63u) | ((mine.algorithm > 1u ? 2u : (uint32_t)mine.algorithm) << 6) | ((mine.is_owner ? 1u : 0u) << 8);
        if (len > 16) { k0 = (unsigned long long)off | ((unsigned long long)len << 32); k1 = 0ull; }
        GMsg* m = &W.gmsg[(size_t)tile * FT + j];
        ulonglong2* mq = (ulonglong2*)m;
        mq[0] = make_ulonglong2(gk, k0);
        mq[1] = make_ulonglong2(k1, (unsigned long long)mine.hits);
        mq[2] = make_ulonglong2((unsigned long long)mine.limit, (unsigned long long)mine.duration);
        mq[3] = make_ulonglong2((unsigned long long)mine.burst,
                                gm_pack(tid, eq_total, f, len <= 16 ? len : 31u, shape, (f & G_CFAR) ? 0 : (int64_t)dmin, (f & G_CFAR) ? 0u : span));
    }
    lds_barrier();
    if (valid) {
        if (errcode) W.did[g] = pd_pack(0, 0, errcode);
        else W.did[g] = pd_pack(sd[head_tid], eq_before, 0);
    }
    GP_STAMPW(0, 5);
}

#ifndef GUBER_PART_WAVES
#define GUBER_PART_WAVES 4               // waves per SIMD the register allocation of k_part must allow (LDS allows 6 workgroups per CU)
#endif
__global__ __launch_bounds__(FT, GUBER_PART_WAVES) void k_part(Table T, BatchView B, Work W) { part_body(T, B, W, blockIdx.x); }

// ---- k_own ----------------------------------------------------------------------------------------------------------------
#ifndef GUBER_OWN_EPT
#define GUBER_OWN_EPT 3
#endif
constexpr int OW_EPT = GUBER_OWN_EPT;    // messages per thread and round
constexpr int OW_MCAP = 256 * OW_EPT;    // messages of one round of an owner (more: the round splits by further bits of the home position)
#ifndef GUBER_OWN_KCAP
#define GUBER_OWN_KCAP 320
#endif
constexpr int OW_KCAP = GUBER_OWN_KCAP;             // distinct keys of one round (uniform keys: 256 +- 16 per owner)
constexpr int OW_HT = 512;               // LDS hash table of the round's keys
constexpr int OW_CH = OW_MCAP / 64;      // 64-message chunks
#ifndef GUBER_OWN_WAVES
#define GUBER_OWN_WAVES 3                // waves per SIMD the register allocation must allow: 3 co-resident workgroups per CU (LDS: 51 KB each)
#endif

// SEG_* bits of a key from the G_* bits its groups raised, the number of groups and the created_at range over the groups
GB_HD uint32_t own_seg_flags(uint32_t f, uint32_t groups, bool created_range) {
    uint32_t sf = 0;
    if (f & G_RETRY) sf |= SEG_RETRY;
    if (f & G_NONUNIFORM) sf |= SEG_NONUNIFORM;
    if ((f & G_ODD) && groups > 1) sf |= SEG_NONUNIFORM;
    if ((f & G_CFAR) && (groups > 1 || (f & G_CREATED))) sf |= SEG_NONUNIFORM;
    if ((f & G_CREATED) || created_range) sf |= SEG_CREATED_DIFFERS;
    return sf;
}

// store the (<= 16-byte, zero-padded) key of a freshly claimed slot from its words
__device__ __forceinline__ void key_store_words(const Table& T, uint64_t slot, unsigned long long k0, unsigned long long k1, uint32_t len) {
    KeyCell* c = &T.buckets[slot].cell;
    c->w[0] = k0; c->w[1] = k1;
#pragma unroll
    for (int i = 2; i < 7; ++i) c->w[i] = 0;
    c->w[7] = (uint64_t)len << 48;
}

// the request shape a message carries, exactly (everything of the request but created_at, whose range travels beside it)
__device__ __forceinline__ bool msg_same_request(const ulonglong2& a1, const ulonglong2& a2, const ulonglong2& a3,
                                                 const ulonglong2& b1, const ulonglong2& b2, const ulonglong2& b3) {
    return a1.y == b1.y && a2.x == b2.x && a2.y == b2.y && a3.x == b3.x && gm_shape(a3.y) == gm_shape(b3.y);
}

// the hash of message i of the batch (either form)
__device__ __forceinline__ unsigned long long msg_hash_at(const Work& W, size_t i) {
    return W.gmsg[i].hash;
}

__device__ __forceinline__ void own_body(const Table& T, const BatchView& B, const Work& W, const uint32_t p, const uint32_t ntiles) {
    __shared__ alignas(16) unsigned long long ktab[OW_HT];          // the round's keys: hash (0 = free)
    __shared__ uint16_t kidOf[OW_HT];                   // table slot -> key id
    __shared__ alignas(16) uint32_t csum[OW_CH * (OW_KCAP / 2)];    // requests per (64-message chunk, key), two u16 per word
    __shared__ uint8_t etile[OW_MCAP];                  // position in the owner's list -> the tile the message comes from
    __shared__ GMsg kref[OW_KCAP];                      // key id -> the message that installed the key; then (same thread) the key's record
    __shared__ uint16_t kwin[OW_KCAP];                  // key id -> list index of that message
    __shared__ uint32_t kfl[OW_KCAP];                   // G_* over the key's groups
    __shared__ uint32_t ktot[OW_KCAP];                  // requests | groups << 20
    __shared__ int kcmin[OW_KCAP], kcmax[OW_KCAP];      // created_at range over the key's groups (ms from the batch clock)
    __shared__ uint16_t tpos[256 + 1], tstart[256];     // per tile: where its run starts in the owner's list, and in the tile's region
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t nkeys, ins_n;
    __shared__ uint32_t stk[40];                        // rounds to do: log2(split) << 24 | residue of the home position
    __shared__ int sp;
    GRec* const krec = (GRec*)kref;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (p >= (1u << pm_bits_of(W))) {                // (256 workgroups are launched per batch whatever the batch's owner count)
#ifdef GUBER_PHASE_TIMING
        if (t == 0) W.dbg[4096 + 2048 + blockIdx.x * 8] = 0ull;       // "did not run": the fold of the stamps skips this workgroup
#endif
        return;
    }

    uint32_t start = 0, c = 0;
    if (t < ntiles) {
        const uint32_t se = W.gse[(size_t)t * PT_PARTS + p]; start = se & 0xffffu; c = se >> 16;
    }
    const size_t mbase = (size_t)t * FT + start;
    if (t == 0) { stk[0] = 0u; sp = 1; ins_n = 0u; }
    tstart[t] = (uint16_t)start;
    GP_STAMPW(1, 0);
    for (;;) {
        lds_barrier();
        if (sp == 0) break;
        const uint32_t cur = stk[sp - 1];
        const uint32_t lg = cur >> 24, res = cur & 0xffffffu, smask = (1u << lg) - 1u;
        lds_barrier();
        if (t == 0) { sp--; nkeys = 0u; }
        static_assert(OW_HT % 2 == 0 && (OW_CH * (OW_KCAP / 2)) % 4 == 0, "cleared sixteen bytes at a time");
        for (uint32_t j = t; j < OW_HT / 2; j += 256) ((uint4*)ktab)[j] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t j = t; j < OW_CH * (OW_KCAP / 2) / 4; j += 256) ((uint4*)csum)[j] = make_uint4(0u, 0u, 0u, 0u);
        // my tile's messages of this round
        uint32_t cm = c;
        if (lg) { cm = 0; for (uint32_t r = 0; r < c; ++r) cm += ((uint32_t)((msg_hash_at(W, mbase + r) >> 7) & T.mask) & smask) == res ? 1u : 0u; }
        uint32_t M;
        {
            const uint32_t incl = wave_incl_scan_u32(cm);
            if (lane == 63) wsum[wave] = incl;
            lds_barrier();                                            // (also: tables cleared, sp / nkeys settled)
            uint32_t before = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) before += w < wave ? wsum[w] : 0u;
            M = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            const uint32_t pos = before + incl - cm;
            tpos[t] = (uint16_t)(pos < 0xffffu ? pos : 0xffffu);
            if (t == 255) tpos[256] = (uint16_t)(M < 0xffffu ? M : 0xffffu);
        }
        if (M == 0) continue;
        GP_STAMP(1, 1);
        bool split = M > OW_MCAP;
        // The owner's list is dealt out evenly: thread t takes messages t, t + 256, t + 512 of the list (whatever tile they come
        // from: binary search over the tiles' list positions), so that every message of the round is requested in ONE trip and a
        // hot owner's long list costs every thread the same.  Per message: where it lives (tile << 8 | position) and its content.
        uint32_t esrc[OW_EPT]; ulonglong2 m0[OW_EPT], m1[OW_EPT], m2[OW_EPT], m3[OW_EPT];
        uint32_t eslot[OW_EPT];
#pragma unroll
        for (int k = 0; k < OW_EPT; ++k) { esrc[k] = 0xffffffffu; eslot[k] = 0; m0[k] = m1[k] = m2[k] = m3[k] = make_ulonglong2(0ull, 0ull); }
        if (!split) {
            // which tile a position of the list belongs to: every tile's thread says so for its own run (one or two messages as a
            // rule) — a message then finds its tile with ONE read (round 4: a binary search over the tiles' list positions, eight
            // dependent reads per message)
            {
                const uint32_t pos = tpos[t];
                for (uint32_t r = 0; r < cm; ++r) etile[pos + r] = (uint8_t)t;
            }
            lds_barrier();                                            // tpos, etile complete
#pragma unroll
            for (int k = 0; k < OW_EPT; ++k) {
                const uint32_t e = (uint32_t)k * 256 + t;
                if (e < M) {
                    const uint32_t lo = etile[e];                     // the tile whose run holds position e
                    uint32_t r = e - tpos[lo];
                    uint32_t mi = lo * FT + tstart[lo] + r;
                    if (lg) {                                         // a split round: the r-th message of the tile's run that belongs to this round
                        const uint32_t mb = lo * FT + tstart[lo];
                        uint32_t seen = 0, x = 0;
                        for (;; ++x) { if (((uint32_t)((msg_hash_at(W, mb + x) >> 7) & T.mask) & smask) == res) { if (seen == r) break; ++seen; } }
                        mi = mb + x;
                    }
                    esrc[k] = mi;
                    const ulonglong2* mq = (const ulonglong2*)&W.gmsg[mi];
                    m0[k] = mq[0]; m1[k] = mq[1]; m2[k] = mq[2]; m3[k] = mq[3];
                }
            }
            // ---- hash -> LDS table slot; the first message of a key installs it and leaves its content as the key's reference ----
#pragma unroll
            for (int k = 0; k < OW_EPT; ++k) {
                if (esrc[k] != 0xffffffffu) {
                    const unsigned long long hh = m0[k].x;
                    uint32_t s = (uint32_t)((hh * 0x9E3779B97F4A7C15ull) >> 55) & (OW_HT - 1);
                    bool won = false;
                    // (bounded: a round with more distinct keys than the table has cells — 513 .. OW_MCAP keys of one owner in one batch —
                    // must come out of here and split, not probe a full table for ever; a key that found no cell raises nkeys past OW_KCAP)
                    for (uint32_t probes = 0;; ++probes) {
                        if (probes == (uint32_t)OW_HT) { atomicAdd(&nkeys, (uint32_t)OW_KCAP + 1u); s = 0; break; }
                        const unsigned long long old = atomicCAS(&ktab[s], 0ull, hh);
                        if (old == 0ull) { won = true; break; }
                        if (old == hh) break;
                        s = (s + 1) & (OW_HT - 1);
                    }
                    if (won) {
                        const uint32_t kid = atomicAdd(&nkeys, 1u);
                        if (kid < OW_KCAP) {
                            kidOf[s] = (uint16_t)kid; kwin[kid] = (uint16_t)((uint32_t)k * 256 + t); kfl[kid] = 0u; ktot[kid] = 0u;
                            const bool far = (gm_flags(m3[k].y) & G_CFAR) != 0u;      // the installer's own created_at range starts the key's
                            kcmin[kid] = far ? INT32_MAX : (int)gm_cmin_delta(m3[k].y);
                            kcmax[kid] = far ? INT32_MIN : (int)gm_cmin_delta(m3[k].y) + (int)gm_cspan(m3[k].y);
                            ulonglong2* kq = (ulonglong2*)&kref[kid];
                            kq[0] = m0[k]; kq[1] = m1[k]; kq[2] = m2[k]; kq[3] = m3[k];
                        }
                    }
                    eslot[k] = s;
                }
            }
            lds_barrier();
            split = nkeys > OW_KCAP;
            if (split && t == 0) atomicAdd(&W.pmode[2], 1u);          // a round that split for its number of keys: what the owner count follows
        }
        if (split) {
            // the round does not fit: two rounds on one more bit of the home position (nothing outside LDS was touched yet)
            if (lg < 24) {
                lds_barrier();
                if (t == 0) { stk[sp] = ((lg + 1) << 24) | res; stk[sp + 1] = ((lg + 1) << 24) | (res | (1u << lg)); sp += 2; }
            } else {
                // (more than OW_KCAP keys on one home position: cannot happen below 2^24 slots; answered RETRY, never evaluated)
                for (uint32_t r = 0; r < c; ++r) {
                    if (((uint32_t)((msg_hash_at(W, mbase + r) >> 7) & T.mask) & smask) != res) continue;
                    GRec* o = &W.grec[mbase + r];
                    { ulonglong2* cs = (ulonglong2*)&W.grs[mbase + r]; cs[0] = make_ulonglong2(0ull, 0ull); cs[1] = make_ulonglong2(0ull, 0ull); }
                    o->limit = o->duration = o->remaining = o->stamp = o->burst = o->expire_at = 0; o->smeta = 0; o->slot = 0;
                    o->tail = gr_tail(SEG_RETRY, 0, 0, 1, (uint32_t)(mbase + r));
                }
            }
            continue;
        }
        const uint32_t nk = nkeys;
        GP_STAMPW(1, 2);
        // ---- the home bucket of my first key (thread = key id) is requested now and used after the messages have been compared.
        // ONLY the bucket: a resident key that sits at its home position (the usual case at load <= 0.5) is recognised by its stored
        // key bytes alone — keys are inserted once and never move, so "this cell holds my key" is "this is my key's slot" — and the
        // directory is read only for keys that are displaced or new ----
        ulonglong2 de0 = {0ull, 0ull}, de1 = de0;
        uint4 tc0 = {0, 0, 0, 0}, tc3 = tc0;
        Rec trec; rec_clear(trec);
        uint64_t kpos = 0;
        const bool haskey = t < nk;
        if (haskey) {
            kpos = (kref[t].hash >> 7) & T.mask;
            const Bucket* hb = &T.buckets[kpos];
            const uint4* cw = (const uint4*)&hb->cell; tc0 = cw[0]; tc3 = cw[3];
            trec = hb->rec;
        }
        GP_STAMP(1, 3);
        // ---- every message against its key's reference: exact key bytes, exact request shape; the key's totals ----
        uint32_t ekid[OW_EPT], ecnt[OW_EPT], ebase[OW_EPT];
#pragma unroll
        for (int k = 0; k < OW_EPT; ++k) {
            ekid[k] = 0; ecnt[k] = 0; ebase[k] = 0;
            if (esrc[k] != 0xffffffffu) {
                const uint32_t li = (uint32_t)k * 256 + t;
                const unsigned long long misc = m3[k].y;
                const uint32_t kid = kidOf[eslot[k]];
                uint32_t f = gm_flags(misc);
                const uint32_t cnt = gm_cnt(misc);
                const bool far = (f & G_CFAR) != 0u;
                const int cmin = (int)gm_cmin_delta(misc), cmax = cmin + (int)gm_cspan(misc);
                bool range = false;                                   // does this group widen the key's created_at range?
                if (kwin[kid] != li) {
                    const ulonglong2* wq = (const ulonglong2*)&kref[kid];
                    const ulonglong2 b0 = wq[0], b1 = wq[1], b2 = wq[2], b3 = wq[3];
                    bool keq = gm_klen(misc) == gm_klen(b3.y);
                    if (keq) {
                        if (gm_klen(misc) != 31u) keq = m0[k].y == b0.y && m1[k].x == b1.x;
                        else keq = req_key_equal_at(B, (uint32_t)m0[k].y, (uint32_t)(m0[k].y >> 32), (uint32_t)b0.y, (uint32_t)(b0.y >> 32));
                    }
                    if (!keq) f |= G_RETRY;
                    if (!msg_same_request(m1[k], m2[k], m3[k], b1, b2, b3)) f |= G_NONUNIFORM;
                    range = !far && ((gm_flags(b3.y) & G_CFAR) || cmin != (int)gm_cmin_delta(b3.y) || cmax != (int)gm_cmin_delta(b3.y) + (int)gm_cspan(b3.y));
                }
                if (f) atomicOr(&kfl[kid], f);
                atomicAdd(&ktot[kid], cnt | (1u << 20));
                if (range) { atomicMin(&kcmin[kid], cmin); atomicMax(&kcmax[kid], cmax); }
                atomicAdd(&csum[(li >> 6) * (OW_KCAP / 2) + (kid >> 1)], cnt << (16 * (kid & 1)));
                ekid[k] = kid; ecnt[k] = cnt;
            }
        }
        lds_barrier();
        GP_STAMP(1, 4);
        // ---- rank base of every group: requests of its key in earlier tiles = in earlier messages of the list ----
#pragma unroll
        for (int k = 0; k < OW_EPT; ++k) {
            if ((uint32_t)k * 256 < M) {                              // (uniform over the workgroup)
                const uint32_t e = (uint32_t)k * 256 + t, chunk = e >> 6;
                const bool act = esrc[k] != 0xffffffffu;
                const uint32_t kid = ekid[k], cnt = ecnt[k];
                uint32_t ctot = 0;
                if (act) ctot = (csum[chunk * (OW_KCAP / 2) + (kid >> 1)] >> (16 * (kid & 1))) & 0xffffu;
                uint32_t base = 0;
                unsigned long long mm = __ballot(act && ctot != cnt);   // keys with several groups inside this chunk: one wave scan each
                while (mm) {
                    const int Lq = __ffsll(mm) - 1;
                    const uint32_t lk = __shfl(kid, Lq, 64);
                    const bool mine = act && kid == lk;
                    const uint32_t incl = wave_incl_scan_u32(mine ? cnt : 0u);
                    if (mine) base = incl - cnt;
                    mm &= ~__ballot(mine);
                }
                if (act) {
                    const uint32_t treq = ktot[kid] & 0xfffffu;
                    if (treq != ctot)
                        for (uint32_t c2 = 0; c2 < chunk; ++c2) base += (csum[c2 * (OW_KCAP / 2) + (kid >> 1)] >> (16 * (kid & 1))) & 0xffffu;
                    ebase[k] = base;
                }
            }
        }
        GP_STAMP(1, 5);
        // ---- one table lookup per key, by the only workgroup that touches this key in this batch ----
        // (no barrier needed before it: a key's reference is read and then replaced by its record by the same thread, and every
        // comparison against references happened before the barrier above)
        int inserted = 0;
        for (uint32_t kid = t; kid < nk; kid += 256) {
            if (kid != t) {                                           // (more than 256 keys in a round: uniform keys.  Touching this bucket beside the first
                                                                      // key's, so that the second pass finds it in the L2: 6.35 -> 6.30 G/s, profiles/r05_n_*)
                kpos = (kref[kid].hash >> 7) & T.mask;
                const Bucket* hb = &T.buckets[kpos];
                const uint4* cw = (const uint4*)&hb->cell; tc0 = cw[0]; tc3 = cw[3];
                trec = hb->rec;
            }
            const GMsg wm = kref[kid];
            const unsigned long long tag = wm.hash;
            const uint32_t klen = gm_klen(wm.misc);
            const uint32_t home = (uint32_t)kpos;
            uint64_t ppos = kpos;
            uint32_t slot = 0, errcode = 0;
            bool cand = false, fresh = false;
            const uint8_t* lkey = nullptr; uint32_t llen = 0;
            if (klen == 31u) { lkey = B.key_bytes + (uint32_t)wm.key0; llen = (uint32_t)(wm.key0 >> 32); }
            // at home?  (keys of <= 16 bytes: the two key words and the length say it all; longer keys go through the directory)
            const bool at_home = klen != 31u && (((uint64_t)tc0.y << 32) | tc0.x) == wm.key0 && (((uint64_t)tc0.w << 32) | tc0.z) == wm.key1 &&
                                 (tc3.w >> 16) == klen;
            if (at_home) { slot = home; cand = true; }
            else {
                // not there: the directory entries of the home position and of the next one, and the next bucket, in one trip
                const uint64_t npos = (kpos + 1) & T.mask;
                de0 = *(const ulonglong2*)&T.dir[kpos]; de1 = *(const ulonglong2*)&T.dir[npos];
            }
            for (uint32_t step = 0; !at_home && step < T.max_probe; ++step, ppos = (ppos + 1) & T.mask) {
                ulonglong2 de = step == 0 ? de0 : de1;
                if (step >= 1) {
                    // a key that is not at its home position: the bucket behind every further directory entry is requested together
                    // with the entry AFTER it (the first displaced step already has its entry: it came with the home position's)
                    if (step >= 2) de = *(const ulonglong2*)&T.dir[ppos];
                    const Bucket* bk = &T.buckets[ppos];
                    const uint4* cw = (const uint4*)&bk->cell; tc0 = cw[0]; tc3 = cw[3];
                    trec = bk->rec;
                }
                unsigned long long tg = de.x;
                if (tg == 0ull) {
                    const unsigned long long old = atomicCAS(&T.dir[ppos].tag, 0ull, tag);
                    if (old == 0ull) {                                // new key: this thread inserts it
                        slot = (uint32_t)ppos; cand = true; fresh = true; inserted++;
                        if (klen != 31u) key_store_words(T, ppos, wm.key0, wm.key1, klen);
                        else if (!key_store(T, ppos, lkey, llen)) { errcode = 6; cand = false; }
                        T.dir[ppos].meta = META_READY;               // (nobody else can be looking for this tag during this launch)
                        break;
                    }
                    tg = old;
                }
                if (tg == tag) { slot = (uint32_t)ppos; cand = true; break; }
            }
            (void)home;
            if (!cand && !errcode) errcode = 6;                       // probe bound exceeded: table full
            uint32_t sf = 0;
            if (cand && !fresh && !at_home) {
                bool eq;
                if (klen != 31u) {
                    eq = (((uint64_t)tc0.y << 32) | tc0.x) == wm.key0 && (((uint64_t)tc0.w << 32) | tc0.z) == wm.key1 && (tc3.w >> 16) == klen;
                } else {
                    eq = key_equal(T, slot, lkey, llen);
                }
                if (!eq) sf |= SEG_RETRY;                             // a 64-bit hash collision with a resident key: careful round
            }
            if (fresh || !cand) rec_clear(trec);
            const uint32_t tot = ktot[kid];
            const uint32_t groups = tot >> 20;
            const int dmin = kcmin[kid], dmax = kcmax[kid];
            sf |= own_seg_flags(kfl[kid], groups, dmin < dmax);
            if ((sf & SEG_CREATED_DIFFERS) && !(sf & SEG_NONUNIFORM) && gm_algo(wm.misc) == ALGO_LEAKY) {
                // requests of a leaky key stamped differently: the run is uniform only if no request of it leaks and none lets the
                // bucket look expired to the ones behind it (leaky_created_harmless, monotone in created_at while nothing wraps)
                Req rq; rq.hits = 0; rq.limit = wm.limit; rq.duration = wm.duration; rq.burst = 0; rq.greg_expire = rq.greg_duration = 0;
                rq.behavior = gm_behavior(wm.misc); rq.algorithm = ALGO_LEAKY; rq.is_owner = 1;
                const int64_t cmin = wadd(B.now_ms, (int64_t)dmin), cmax = wadd(B.now_ms, (int64_t)dmax);
                bool ok = !(kfl[kid] & G_CFAR) && dmin <= dmax;
                const int64_t d0 = wsub(cmin, trec.stamp), d1 = wsub(cmax, trec.stamp);
                ok = ok && d0 <= d1 && (uint64_t)d1 - (uint64_t)d0 == (uint64_t)cmax - (uint64_t)cmin;
                ok = ok && wadd(cmax, rq.duration) >= wadd(cmin, rq.duration);
                if (ok) { rq.created_at = cmax; ok = leaky_created_harmless(trec, rq, B.now_ms); }
                if (ok) { rq.created_at = cmin; ok = leaky_created_harmless(trec, rq, B.now_ms); }
                if (!ok) sf |= SEG_NONUNIFORM;
            }
            GRec kr;
            kr.limit = trec.limit; kr.duration = trec.duration; kr.remaining = trec.remaining; kr.stamp = trec.stamp; kr.burst = trec.burst;
            kr.expire_at = trec.expire_at;
            kr.smeta = pack_smeta(trec, 1); kr.slot = slot;
            {   // does the 32-byte record carry this key?  (nothing flagged, no error, and the rest of the bucket is what the request says)
                const uint32_t kind = rec_kind(trec);
                bool cok = sf == 0u && errcode == 0u && cand && trec.invalid_at == 0 && slot < (1u << 26) && (kfl[kid] & ~(uint32_t)G_LONG) == 0u;
                if (kind == K_ABSENT)
                    cok = cok && rec_meta(trec) == 0u && (trec.limit | trec.duration | trec.remaining | trec.stamp | trec.burst | trec.expire_at) == 0;
                else
                    cok = cok && kind != K_NIL && rec_algo(trec) <= 1u && trec.limit == wm.limit && trec.duration == wm.duration &&
                          (trec.burst == 0 || trec.burst == wm.burst);
                if (cok) kr.smeta |= SM_COMPACT_OK | (trec.burst == 0 ? SM_BURST_ZERO : 0u);
            }
            const uint32_t wl = kwin[kid];                             // the installing message: list index -> its place in gmsg
            uint32_t seg = 0;
            {   // (the installer's esrc lives in its thread's registers: recomputed from the list index as every thread did)
                const uint32_t lo = etile[wl];
                seg = lo * FT + tstart[lo] + (wl - tpos[lo]);
                if (lg) {
                    const uint32_t mb = lo * FT + tstart[lo];
                    uint32_t seen = 0, x = 0;
                    for (;; ++x) { if (((uint32_t)((msg_hash_at(W, mb + x) >> 7) & T.mask) & smask) == res) { if (seen == wl - tpos[lo]) break; ++seen; } }
                    seg = mb + x;
                }
            }
            kr.tail = gr_tail(sf | (errcode ? SEG_ERR : 0u), errcode, 0, tot & 0xfffffu, seg);
            if (trec.invalid_at != 0) W.sinv[seg] = trec.invalid_at;
            krec[kid] = kr;
        }
        GP_STAMPW(1, 6);
        lds_barrier();
        // ---- every group gets its record: the key's, with the group's base ----
#pragma unroll
        for (int k = 0; k < OW_EPT; ++k) {
            if (esrc[k] != 0xffffffffu) {
                const ulonglong2* kq = (const ulonglong2*)&krec[ekid[k]];
                ulonglong2 q0 = kq[0], q1 = kq[1], q2 = kq[2], q3 = kq[3];
                q3.y |= (unsigned long long)(ebase[k] & 0xffffu) << 16;
                const uint32_t src = esrc[k];
                ulonglong2* cs = (ulonglong2*)&W.grs[src];
                const uint32_t sm = (uint32_t)q3.x;
                if (sm & SM_COMPACT_OK) {
                    cs[0] = q1;                                       // remaining, stamp
                    cs[1] = make_ulonglong2(q2.y, grs_pack(sm & 3u, (sm >> 2) & 1u, (sm >> 8) & 1u, (sm & SM_BURST_ZERO) != 0u, ebase[k],
                                                           ((uint32_t)(q3.y >> 32) & 0xffffu) + 1u, (uint32_t)(q3.x >> 32)));
                } else {
                    cs[0] = make_ulonglong2(0ull, 0ull); cs[1] = make_ulonglong2(0ull, 0ull);        // "see grec[]"
                    ulonglong2* o = (ulonglong2*)&W.grec[src];
                    o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3;
                }
                if ((uint32_t)q3.y & (SEG_NONUNIFORM | SEG_CREATED_DIFFERS)) {   // the walk's map: which tiles hold the segment, and as which group
                    const uint32_t seg = (uint32_t)(q3.y >> 48), tile = src >> 8;
                    atomicOr(&W.segtiles[(size_t)seg * 4 + (tile >> 6)], 1ull << (tile & 63));
                    W.tilerow[(size_t)seg * FT_MAX_TILES + tile] = (uint16_t)(src & 255u);
                }
            }
        }
        if (inserted) atomicAdd(&ins_n, (uint32_t)inserted);
    }
    GP_STAMPW(1, 7);
    if (t == 0 && ins_n) atomicAdd(&T.ctr->tags_used, (unsigned long long)ins_n);
}

__global__ __launch_bounds__(256, GUBER_OWN_WAVES) void k_own(Table T, BatchView B, Work W, uint32_t ntiles) { own_body(T, B, W, blockIdx.x, ntiles); }

// ---- k_eval3 --------------------------------------------------------------------------------------------------------------
// Request order, workgroup = tile.  As k_eval2 from the evaluation on; what differs is where a request learns its segment:
// packed word -> its group's record (ONE sector: bucket, slot, flags, base, total), no bitmaps, no LDS pre-pass, no barrier
// before the evaluation.  The serial walk of a heterogeneous segment follows the segment's tile map (Work::segtiles, tilerow).
// (Measured and not kept, round 4: the closed forms and the rest as two launches — the first at eight waves per SIMD: +1 % / -7 %,
// profiles/r04_split_eval3_ab.txt.)
__device__ __forceinline__ void eval3_body(const EvalArgs& A, const uint32_t tile) {
    const Table& T = A.T; const BatchView& B = A.B; const ResultView& R = A.R; const Work& W = A.W;
    __shared__ unsigned long long cnt[4];
    const uint32_t i = tile * 256 + threadIdx.x;
    const bool live = i < B.n;
    GP_STAMP(2, 0);
    const uint32_t dl = live ? W.did[i] : 0u;
    const uint32_t gj = dl & 0xffu, lr = (dl >> 8) & 0xffu, derr = (dl >> 16) & 0xffu;
    uint32_t sf = 0, slot = 0, smeta = 0, base = 0, total = 1, d = 0, rerr = 0; Req r; Rec s0;
    rec_clear(s0);
    if (live) {
        r = load_req_nogreg(B, i);
        bool full = !derr;
        if (!derr) {
            const ulonglong2* cq = (const ulonglong2*)&W.grs[(size_t)tile * 256 + gj];   // 32 bytes per (key, tile) group
            const ulonglong2 c0 = cq[0], c1 = cq[1];
            const unsigned long long pk = c1.y;
            if (pk & 1ull) {                                          // the usual record: the rest of the bucket is what the request says
                full = false;
                const uint32_t kind = (uint32_t)(pk >> 1) & 3u;
                if (kind != K_ABSENT) {
                    s0.limit = r.limit; s0.duration = r.duration; s0.remaining = (int64_t)c0.x; s0.stamp = (int64_t)c0.y;
                    s0.burst = ((pk >> 5) & 1ull) ? 0 : r.burst; s0.expire_at = (int64_t)c1.x;
                    smeta = kind | (((uint32_t)(pk >> 3) & 1u) << 2) | (((uint32_t)(pk >> 4) & 1u) << 8);
                    s0.meta = smeta_meta(smeta);
                }
                base = (uint32_t)(pk >> 6) & 0xffffu; total = ((uint32_t)(pk >> 22) & 0xffffu) + 1u; slot = (uint32_t)(pk >> 38);
            }
        }
        if (full) {
            const ulonglong2* q = (const ulonglong2*)&W.grec[(size_t)tile * 256 + gj];   // one 64-byte sector per (key, tile) group
            const ulonglong2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            s0.limit = (int64_t)q0.x; s0.duration = (int64_t)q0.y; s0.remaining = (int64_t)q1.x; s0.stamp = (int64_t)q1.y;
            s0.burst = (int64_t)q2.x; s0.expire_at = (int64_t)q2.y;
            smeta = (uint32_t)q3.x; slot = (uint32_t)(q3.x >> 32);
            s0.meta = smeta_meta(smeta);
            sf = (uint32_t)q3.y & 0xffu; rerr = (uint32_t)(q3.y >> 8) & 0xffu;
            base = (uint32_t)(q3.y >> 16) & 0xffffu; total = ((uint32_t)(q3.y >> 32) & 0xffffu) + 1u; d = (uint32_t)(q3.y >> 48);
        }
    }
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0ull;
    GP_STAMPW(2, 1);
    int c_over = 0, c_hit = 0, c_miss = 0, c_size = 0;
    // The wave in which EVERY request is the common case — a live token bucket met by a request that does not reconfigure it, nothing
    // flagged, no Store side channel, no GLOBAL bookkeeping — runs straight through the closed form: one ballot instead of the general
    // path's cascade of divergent branches (error codes, flags, the leaky form, the serial walk, events, queues), each of which costs
    // every wave its exec-mask bookkeeping whether a lane takes it or not.  Exactly the general path's own steps for such a request
    // (token_fast_ok -> token_fast -> store -> the run's last request writes the bucket), so the results are the same by construction.
    const bool plain = !live || (!derr && sf == 0u && !(smeta & SM_HAS_INVALID) && token_fast_ok(s0, r, B.now_ms));
    const bool plain_wave = !W.store_flags && !T.gpend && __ballot(!plain) == 0ull;
    if (plain_wave) {
        if (live) {
            const uint32_t rank = base + lr;
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
        const uint32_t rank = base + lr;
        const bool flagged = (sf & (SEG_NONUNIFORM | SEG_CREATED_DIFFERS)) != 0u;
        if (derr) {
            store_err(R, i, (uint8_t)derr);
        } else if (sf & SEG_ERR) {
            store_err(R, i, (uint8_t)rerr);
        } else if (sf & SEG_RETRY) {
            store_err(R, i, IE_RETRY);
            atomicAdd(&T.ctr->retries, 1ull);
        } else {
            if (smeta & SM_HAS_INVALID) s0.invalid_at = W.sinv[d];
            bool parallel = !(sf & SEG_NONUNIFORM);
            if (parallel && (sf & SEG_CREATED_DIFFERS)) {
                // requests that differ only in created_at: a live token bucket never reads it; for a leaky bucket the key's owner
                // has checked that no request of the run leaks (k_own) — both decisions are the same for every request of the segment
                parallel = !(T.gpend && (r.behavior & BH_GLOBAL));
                if (parallel && r.algorithm != ALGO_LEAKY) parallel = created_at_irrelevant(s0, r, B.now_ms);
            }
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
                // walk iterator: the tiles holding the segment in order (its tile map), inside a tile the requests of the segment's group
                uint32_t wv = 0, tt = 0, q = FT, tj = 0, mm = 0u;          // (the map read as 8 x 32 tiles)
                for (;;) {
                    uint32_t j = i;
                    if (walk) {
                        bool found = false, end = false;
                        while (!found && !end) {
                            if (q < FT && tt * FT + q < B.n) {
                                const uint32_t id = W.did[(size_t)tt * FT + q];
                                if ((id & 0xffu) == tj && ((id >> 16) & 0xffu) == 0u) { j = tt * FT + q; found = true; }
                                q++;
                            } else {
                                while (mm == 0u && wv < 8) { mm = ((const uint32_t*)W.segtiles)[(size_t)d * 8 + wv]; tt = wv * 32; wv++; }
                                if (mm == 0u) end = true;
                                else {
                                    const uint32_t bpos = (uint32_t)__ffs((int)mm) - 1u; mm &= mm - 1u; tt = (tt & ~31u) + bpos; q = 0;
                                    tj = W.tilerow[(size_t)d * FT_MAX_TILES + tt];
                                }
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
        // the first request of a segment that carries a tile map clears it (walked or not): the map is all zero between batches
        if (!derr && flagged && rank == 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w) W.segtiles[(size_t)d * 4 + w] = 0ull;
        }
    }
    GP_STAMP(2, 2);
    {
        const int w_over = wave_sum(c_over), w_hit = wave_sum(c_hit), w_miss = wave_sum(c_miss), w_size = wave_sum(c_size);
        lds_barrier();                                               // (cnt zeroed)
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
    // the owner count of the next batch (all of this batch's k_own workgroups are done, the next batch's k_part has not started)
    if (tile == 0 && threadIdx.x == 0) {
        uint32_t* pm = W.pmode;
        const uint32_t splits = pm[2];
        pm[2] = 0u;
        if (pm[3] == 0u) {
            if (pm[0] == 7u) { if (splits >= PM_SPLITS) { pm[0] = 8u; pm[1] = PM_HOLD; } }
            else if (pm[1] > 1u) pm[1]--;
            else { pm[0] = 7u; pm[1] = 0u; }
        }
        if (W.pmslot) pm[3u + W.pmslot] = pm[0];                  // (the batch after the next reads it: pm_bits_of)
    }
    GP_STAMPW(2, 3);
}
__global__ __launch_bounds__(256, GUBER_EVAL2_WAVES) void k_eval3(EvalArgs A) {
    const EvalArgs* a = (const EvalArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    eval3_body(*a, blockIdx.x);
}

// ---- several engines in one launch (as k_front_multi / k_eval2_multi: workgroup -> (batch, tile) by the prefix table in the
// kernel arguments; k_own_multi: 256 owners per batch, so that an owner's XCD is the same in every batch) ----------------------
__global__ __launch_bounds__(FT, GUBER_PART_WAVES) void k_part_multi(MultiFront A) {
    const MultiFront* m = (const MultiFront*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t first;
    const uint32_t sb = multi_batch_of<MULTI_MAX>(m->end_tile, blockIdx.x, first);
    const FrontArgs* a = (const FrontArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiFront, sub)) + sb;
    part_body(a->T, a->B, a->W, blockIdx.x - first);
}
__global__ __launch_bounds__(256, GUBER_OWN_WAVES) void k_own_multi(MultiFront A) {
    const MultiFront* m = (const MultiFront*)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t sb = blockIdx.x / PT_PARTS;
    const uint32_t ntiles = m->end_tile[sb] - (sb ? m->end_tile[sb - 1] : 0u);
    const FrontArgs* a = m->sub + sb;
    own_body(a->T, a->B, a->W, blockIdx.x % PT_PARTS, ntiles);
}
__global__ __launch_bounds__(256, GUBER_EVAL2_WAVES) void k_eval3_multi(MultiEval A) {
    const MultiEval* m = (const MultiEval*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t first;
    const uint32_t sb = multi_batch_of<MULTI_MAX>(m->end_tile, blockIdx.x, first);
    const EvalArgs* a = (const EvalArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiEval, sub)) + sb;
    eval3_body(*a, blockIdx.x - first);
}

// ---- batch b's k_eval3 and batch b + 1's k_part of the same tables in ONE launch (GUBER_FUSE_EP=1; built and checked through the
// kernel source on the CPU, off until it has been measured on the GPU).  A stream's passes are k_part, k_own, k_eval3, k_part, ...:
// k_part(b + 1) needs nothing k_eval3(b) produces — it hashes and groups the next batch's requests and reads no bucket — so the two
// can share a launch and a pass is two launches long instead of three (what bounds the fused pipeline is the latency of the
// launches' chains: DESIGN.md section 4).  What the two halves must not share, they do not: the packed words (Work::did) are
// double-buffered by batch parity, the owner count of a batch comes from its parity's slot (pm_bits_of), messages / {start, count}
// are dead once k_own(b) has run, k_eval3 reads records, k_part writes messages.  The engine only fuses what maintenance leaves
// alone: same tables in the same order, nothing to enqueue or to read first (guber_engine.hip launch_group).  A counter read-back
// riding on the k_part half reads counters that k_eval3's tiles are adding to: the host counts that batch as still to come.
// Workgroups [0, end_e[nb - 1]) are k_eval3's tiles, the rest k_part's.
constexpr int EP_MAX = MULTI_MAX < 4 ? MULTI_MAX : 4;     // (four tables' arguments fit the 4 KB kernel-argument segment)
struct EPSub {
    EvalArgs E; BatchView Bp; uint32_t* did_p; uint32_t pmslot_p;
    uint32_t snap_seq, snap_n, pad_; DevCounters* snap_c; BlockCounters* snap_b; uint32_t* snap_stamp;     // a counter read-back riding on the k_part half (Work::snap_*)
};
struct MultiEP {
    uint32_t nb; uint32_t end_e[EP_MAX]; uint32_t end_p[EP_MAX]; EPSub sub[EP_MAX];
    MultiEP() { memset((void*)this, 0, sizeof *this); for (int k = 0; k < EP_MAX; ++k) end_e[k] = end_p[k] = 0xffffffffu; }
};
static_assert(sizeof(MultiEP) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256, GUBER_EVAL2_WAVES) void k_evalpart_multi(MultiEP A) {
    static_assert(FT == 256, "k_eval3's workgroup is k_part's tile");
    const MultiEP* m = (const MultiEP*)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t tiles_e = m->end_e[m->nb - 1];
    const bool part = blockIdx.x >= tiles_e;
    const uint32_t wg = part ? blockIdx.x - tiles_e : blockIdx.x;
    uint32_t first;
    const uint32_t sb = multi_batch_of<EP_MAX>(part ? m->end_p : m->end_e, wg, first);
    const EPSub* a = (const EPSub*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiEP, sub)) + sb;
    if (!part) { eval3_body(a->E, wg - first); return; }
    Work W = a->E.W;                                       // the next batch's work arrays are this engine's, but for:
    W.did = a->did_p; W.pmslot = a->pmslot_p;
    W.snap_seq = a->snap_seq; W.snap_n = a->snap_n; W.snap_c = a->snap_c; W.snap_b = a->snap_b; W.snap_stamp = a->snap_stamp;
    part_body(a->E.T, a->Bp, W, wg - first);
}

}  // namespace guber
