// guber_kernels_small.h — batches of at most one tile (n <= 256): ONE launch, one workgroup.  BASELINE configs[0] (the
// reference's BenchmarkServer shape, benchmark_test.go:63-84: one request per call) and the small flushes of a lightly
// loaded batcher live here; the two-launch pipeline's per-batch bookkeeping (claims, segment records, tile bitmaps)
// exists to coordinate tiles and is pointless with one.
//
// The workgroup groups its requests by key in LDS (as k_front), the head of every group finds / inserts the bucket and
// shares it through LDS, every request evaluates its own rank (as k_eval2), the last one of a group writes the bucket
// back.  Inputs are read and results written in place — the engine hands the kernel device-visible HOST memory, so
// there is no copy launch on either side — and completion is a sequence number stored to host memory after a
// system-scope release, which the caller polls instead of synchronising the stream.
//
// What the fast path does not do itself — groups whose requests differ, a 64-bit hash collision — it detects BEFORE
// touching any bucket and reports (SmallOut::fallback); the caller re-runs the batch through the general pipeline.
#pragma once
#include "guber_table.h"

namespace guber {

struct SmallOut {                 // lives in host memory (hipHostMallocCoherent)
    unsigned int done;            // = the call's sequence number when everything else is visible
    unsigned int fallback;        // 1: nothing was evaluated, use the general pipeline
    unsigned int over, hits, misses;
    int size_delta;
};

struct SmallReqs {
    int64_t hits[FT], limit[FT], duration[FT], burst[FT], created_at[FT];
    unsigned long long misc[FT];          // behavior | algorithm << 32 | is_owner << 40
};

// map (LDS, or nullptr): request tid of this workgroup's batch is entry map[tid] of the arrays B and R point at — a share of a
// routed stage, picked out of the stage's arrival order (k_small_routed)
__device__ __forceinline__ void small_body(const Table& T, const BatchView& B, const ResultView& R, SmallOut* out, const uint32_t seq, const uint64_t touch,
                                           const uint32_t* map = nullptr) {
    constexpr int GT_BITS = 9, GT = 1 << GT_BITS;
    __shared__ unsigned long long gkey[GT];
    __shared__ unsigned long long gbits[FT / 64][GT];
    __shared__ SmallReqs sreq;
    __shared__ Rec srec[FT];
    __shared__ uint32_t sslot[FT], soff[FT], slen[FT], serr[FT];
    __shared__ uint32_t bail, ins_n;
    __shared__ unsigned long long cnt[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool valid = tid < B.n;
    const uint32_t at = (map && valid) ? map[tid] : tid;              // where this request's columns and its answer live
    if (tid == 0) { bail = 0u; ins_n = 0u; }
    if (tid < 4) cnt[tid] = 0ull;
    for (uint32_t j = tid; j < GT; j += FT) {
        gkey[j] = 0ull;
#pragma unroll
        for (int w = 0; w < FT / 64; ++w) gbits[w][j] = 0ull;
    }
    uint32_t errcode = 0, off = 0, len = 0;
    const uint8_t* key = nullptr;
    unsigned long long gk = 0ull;
    uint64_t h = 0;
    Req r;
    if (valid) {
        r = load_req(B, at);
        off = key_off_of(B, at);
        len = key_len_of(B, at, off);
        key = B.key_bytes + off;
        if (len == 0) errcode = IE_EMPTY_KEY;
        else if (len > T.max_key) errcode = 7;
        if (!errcode) { h = xxhash64(key, len, 0) & T.hash_mask; gk = h ? h : 1ull; }
        sreq.hits[tid] = r.hits; sreq.limit[tid] = r.limit; sreq.duration[tid] = r.duration; sreq.burst[tid] = r.burst;
        sreq.created_at[tid] = r.created_at;
        sreq.misc[tid] = (unsigned long long)r.behavior | ((unsigned long long)r.algorithm << 32) | ((unsigned long long)r.is_owner << 40);
    }
    soff[tid] = off; slen[tid] = len;
    __syncthreads();
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
    __syncthreads();
    uint32_t rank = 0, total = 1, head_tid = tid;
    if (gk) {
        bool found_head = false;
        total = 0;
#pragma unroll
        for (uint32_t w = 0; w < FT / 64; ++w) {
            const unsigned long long bw = gbits[w][gh];
            const uint32_t c = __popcll(bw);
            total += c;
            if (w < wave) rank += c;
            else if (w == wave) rank += __popcll(bw & ((1ull << lane) - 1ull));
            if (!found_head && bw) { head_tid = w * 64 + (uint32_t)__ffsll((unsigned long long)bw) - 1; found_head = true; }
        }
    }
    const bool head = valid && gk != 0ull && rank == 0;
    // members: same key bytes and the same request as the head, or the general pipeline takes the batch
    if (valid && gk && rank != 0) {
        const uint32_t ho = soff[head_tid], hl = slen[head_tid];
        bool same = hl == len;
        if (same) {
            const uint32_t nw = (len + 7) >> 3;
            for (uint32_t w = 0; w < nw && same; ++w) {
                uint64_t x = ld_key_word(key + 8 * w) ^ ld_key_word(B.key_bytes + ho + 8 * w);
                if (w == nw - 1) x &= tail_mask(len - 8 * w);
                same = x == 0;
            }
        }
        same = same && sreq.hits[head_tid] == r.hits && sreq.limit[head_tid] == r.limit && sreq.duration[head_tid] == r.duration &&
               sreq.burst[head_tid] == r.burst && sreq.created_at[head_tid] == r.created_at &&
               sreq.misc[head_tid] == ((unsigned long long)r.behavior | ((unsigned long long)r.algorithm << 32) | ((unsigned long long)r.is_owner << 40)) &&
               !(r.behavior & BH_GREGORIAN && B.greg_expire && B.greg_duration &&
                 (B.greg_expire[tid] != B.greg_expire[head_tid] || B.greg_duration[tid] != B.greg_duration[head_tid]));
        if (!same) bail = 1u;
    }
    // heads: find or insert the bucket
    uint32_t slot = 0;
    int inserted = 0;
    Rec rec; rec_clear(rec);
    if (head) {
        const unsigned long long tag = gk;
        uint64_t pos = (h >> 7) & T.mask;
        const uint32_t home = (uint32_t)pos;
        bool cand = false, ready = false;
        // the home bucket is requested together with the home directory entry (one trip for keys at their home position)
        const ulonglong2 de0 = *(const ulonglong2*)&T.dir[pos];
        const Bucket* hb = &T.buckets[pos];
        KeyCell hc = hb->cell;
        rec = hb->rec;
        for (uint32_t step = 0; step < T.max_probe; ++step, pos = (pos + 1) & T.mask) {
            ulonglong2 de = de0;
            if (step) de = *(const ulonglong2*)&T.dir[pos];
            unsigned long long t = de.x, m = de.y;
            if (t == 0ull) {
                const unsigned long long old = atomicCAS(&T.dir[pos].tag, 0ull, tag);
                if (old == 0ull) {
                    slot = (uint32_t)pos; inserted = 1; cand = true;
                    if (!key_store(T, pos, key, len)) { errcode = 6; cand = false; }
                    break;
                }
                t = old;
                m = ld_agent(&T.dir[pos].meta);
            }
            if (t == tag) { slot = (uint32_t)pos; cand = true; ready = (m & META_READY) != 0; break; }
        }
        if (!cand && !errcode) errcode = 6;
        if (inserted) rec_clear(rec);
        else if (cand) {
            // entries are READY here (the previous launch published its inserts); a different key under the same 64-bit
            // hash goes to the general pipeline, whose careful round probes past it
            if (slot != home) { hc = T.buckets[slot].cell; rec = T.buckets[slot].rec; }
            bool eq = ready && (uint32_t)(hc.w[7] >> 48) == len;
            if (eq) {
                if (len <= INLINE_KEY) {
                    const uint32_t nw = (len + 7) >> 3;
#pragma unroll
                    for (uint32_t w = 0; w < 8; ++w) {
                        if (w < nw) {
                            uint64_t kv = ld_key_word(key + 8 * w), cv = hc.w[w];
                            if (w == 7) cv &= 0x0000ffffffffffffull;
                            if (w == nw - 1) { const uint64_t mk = tail_mask(len - 8 * w); kv &= mk; cv &= mk; }
                            eq = eq && kv == cv;
                        }
                    }
                } else eq = key_equal(T, slot, key, len);
            }
            if (!eq) bail = 1u;
        }
        if (inserted) { atomicOr(&T.dir[slot].meta, META_READY); atomicAdd(&ins_n, 1u); }
        srec[tid] = rec; sslot[tid] = slot; serr[tid] = errcode;
    }
    __syncthreads();
    if (tid == 0 && ins_n) atomicAdd(&T.ctr->tags_used, (unsigned long long)ins_n);
    if (bail) {                                                      // nothing has touched a bucket yet
        if (tid == 0) {
            out->fallback = 1u; out->over = out->hits = out->misses = 0u; out->size_delta = 0;
            __threadfence_system();
            __hip_atomic_store(&out->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    int c_over = 0, c_hit = 0, c_miss = 0, c_size = 0;
    if (valid) {
        if (gk) { errcode = serr[head_tid]; slot = sslot[head_tid]; }
        if (errcode) {
            store_err(R, at, (uint8_t)errcode);
        } else {
            const Rec s0 = srec[head_tid];
            Rec after; Resp o;
            uint32_t ev = 0;
            bool done = false;
            if (token_fast_ok(s0, r, B.now_ms)) { ev = token_fast(s0, r, rank, o, after); done = true; }
            else if (leaky_fast(s0, r, B.now_ms, rank, o, after, ev)) done = true;
            if (!done) ev = eval_uniform_rank_1x(s0, r, B.now_ms, rank, o, after);
            store_resp(R, at, o);
            c_over = (ev & EV_OVER) ? 1 : 0; c_hit = (ev & EV_HIT) ? 1 : 0; c_miss = (ev & EV_MISS) ? 1 : 0;
            if (rank == total - 1 && r.algorithm <= ALGO_LEAKY) {      // (an invalid algorithm never reaches the cache: workers.go:317-321)
                rec_set_stamp(after, touch + tid);                          // the key's place in the recency order: its last request
                T.buckets[slot].rec = after;
                c_size = (int)(rec_kind(after) != K_ABSENT) - (int)(rec_kind(s0) != K_ABSENT);
                if (o.err == 0) queue_global(T, slot, r, (uint64_t)rank + 1);
            }
        }
    }
    {
        const int w_over = wave_sum(c_over), w_hit = wave_sum(c_hit), w_miss = wave_sum(c_miss), w_size = wave_sum(c_size);
        if (lane == 0) {
            if (w_over) atomicAdd(&cnt[0], (unsigned long long)w_over);
            if (w_hit) atomicAdd(&cnt[1], (unsigned long long)w_hit);
            if (w_miss) atomicAdd(&cnt[2], (unsigned long long)w_miss);
            if (w_size) atomicAdd(&cnt[3], (unsigned long long)(long long)w_size);
        }
    }
    __syncthreads();                                                 // drains every store of the workgroup (vmcnt(0)) before the flag
    if (tid == 0) {
        BlockCounters* bc = &T.bctr[0];
        bc->over += cnt[0]; bc->hits += cnt[1]; bc->misses += cnt[2]; bc->size_delta += (long long)cnt[3];
        out->fallback = 0u; out->over = (unsigned int)cnt[0]; out->hits = (unsigned int)cnt[1]; out->misses = (unsigned int)cnt[2];
        out->size_delta = (int)(long long)cnt[3];
        __threadfence_system();
        __hip_atomic_store(&out->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(FT) void k_small(Table T, BatchView B, ResultView R, SmallOut* out, uint32_t seq, uint64_t touch) { small_body(T, B, R, out, seq, touch); }

// the small batches of several engines (the logical shards of a GPU: one table each) in ONE launch, one workgroup per batch —
// what a pool's dispatcher has when a handful of requests arrive spread over its shards (guber_stages_submit)
constexpr int SMALL_MULTI_MAX = 8;
struct SmallArgs { Table T; BatchView B; ResultView R; SmallOut* out; uint64_t touch; uint32_t seq; };
struct MultiSmall { uint32_t nb; SmallArgs sub[SMALL_MULTI_MAX]; };
static_assert(sizeof(MultiSmall) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(FT) void k_small_multi(MultiSmall A) {
    const SmallArgs* a = (const SmallArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiSmall, sub)) + blockIdx.x;
    small_body(a->T, a->B, a->R, a->out, a->seq, a->touch);
}

// The one-launch path for a routed stage of at most one tile (guber_stage_submit_routed: a handful of requests spread over the
// shards of a GPU — a lightly loaded pool): one workgroup per engine that has a share; it picks its requests out of the stage's
// arrival order by the dest column (engine << 24 | rank: the rank is the request's place in the workgroup), then runs the
// one-launch body on them in place.  Every share reports its own outcome (SmallOut): a share the fast path declines is re-run by
// the host through the general pipeline, the others stand.
struct SmallRoutedSub { Table T; SmallOut* out; uint64_t touch; uint32_t seq, n, engine; };
struct MultiSmallRouted { uint32_t nb, n_total; const uint32_t* dest; BatchView B; ResultView R; SmallRoutedSub sub[16]; };
static_assert(sizeof(MultiSmallRouted) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(FT) void k_small_routed(MultiSmallRouted A) {
    __shared__ uint32_t map[FT];
    __shared__ uint32_t bad;
    const SmallRoutedSub* a = (const SmallRoutedSub*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MultiSmallRouted, sub)) + blockIdx.x;
    // dest is written by the caller (guber_stage_dest is public ABI): the ranks of this engine's share must be a permutation of
    // 0..n-1 — anything else (a rank twice, a rank >= n, a count that disagrees with dest) would send small_body to an arbitrary
    // index of the stage's host arrays.  Such a share is not evaluated: it reports fallback and the host re-runs it by index list.
    map[threadIdx.x] = 0xffffffffu;
    if (threadIdx.x == 0) bad = 0u;
    __syncthreads();
    if (threadIdx.x < A.n_total) {
        const uint32_t dv = A.dest[threadIdx.x];
        if ((dv >> 24) == a->engine) {
            const uint32_t rank = dv & 0xffffffu;
            if (rank >= a->n || rank >= FT || atomicCAS(&map[rank], 0xffffffffu, threadIdx.x) != 0xffffffffu) bad = 1u;
        }
    }
    __syncthreads();
    if (threadIdx.x < a->n && map[threadIdx.x] == 0xffffffffu) bad = 1u;     // a rank nobody took
    __syncthreads();
    if (bad) {
        if (threadIdx.x == 0) {
            a->out->fallback = 1u; a->out->over = a->out->hits = a->out->misses = 0u; a->out->size_delta = 0;
            __threadfence_system();
            __hip_atomic_store(&a->out->done, a->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    BatchView B = A.B;
    B.n = a->n;
    small_body(a->T, B, A.R, a->out, a->seq, a->touch, map);
}

}  // namespace guber
