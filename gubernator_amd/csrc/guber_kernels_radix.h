// guber_kernels_radix.h — the large-batch pipeline (n > 65 536 requests, or GUBER_FLAG_TEST_FORCE_RADIX): a global stable
// LSD radix sort of the requests by segment id (k_resolve, k_hist, k_scatter x P, k_heads) and k_eval.
// Overview at the top of guber_kernels.h.
#pragma once
#include "guber_table.h"

namespace guber {

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
        const uint32_t off = key_off_of(B, i);
        const uint32_t len = key_len_of(B, i, off);
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
                const uint32_t off = key_off_of(B, g);
                if (!key_equal(T, slot, B.key_bytes + off, key_len_of(B, g, off))) atomicOr(&W.seg_flags[d], SEG_RETRY);
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
                if (rank == last - first && r.algorithm <= ALGO_LEAKY) {   // (an invalid algorithm never reaches the cache: workers.go:317-321)
                    rec_set_stamp(after, W.touch + i);            // the key's place in the recency order: its last request
                    T.buckets[slot].rec = after;
                    c_size = (int)(rec_kind(after) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
                    if (out.err == 0) queue_global(T, slot, r, (uint64_t)rank + 1);
                }
            } else if (rank == 0) {
                // requests to this key differ: apply them one by one in request order
                Rec s = s0;
                uint32_t lastj = 0xffffffffu;                      // the last request of the segment that reached the cache
                for (uint32_t q = first; q <= last; ++q) {
                    const uint32_t j = W.order[q];
                    const Req rj = load_req(B, j);
                    if (rj.algorithm <= ALGO_LEAKY) lastj = j;
                    Resp out;
                    const uint32_t ev = apply(s, rj, B.now_ms, out);
                    store_resp(R, j, out);
                    store_events(W, j, ev, s);
                    if (out.err == 0) queue_global(T, slot, rj, 1);
                    c_over += (ev & EV_OVER) ? 1 : 0; c_hit += (ev & EV_HIT) ? 1 : 0; c_miss += (ev & EV_MISS) ? 1 : 0;
                }
                if (lastj != 0xffffffffu) {
                    rec_set_stamp(s, W.touch + lastj);
                    T.buckets[slot].rec = s;
                    c_size = (int)(rec_kind(s) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
                }
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

}  // namespace guber
