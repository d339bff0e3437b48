// guber_kernels_front.h — the device-resident front of a GPU's logical shards (guber_front_*, guber_front.h).
// Included by guber_kernels.h (and, like the batch pipelines, compiled for the host by the tests' fiber emulation).
#pragma once

namespace guber {

// ---- one generation of requests in ARRIVAL order -> the shards' shares -> the answers in ARRIVAL order --------------------------
// What the reference does per request — WorkerPool.GetRateLimit picks the worker from the XXH64 of the HashKey (workers.go:261-289,
// getWorker :180-184) and GetRateLimits answers in request order (gubernator.proto:51-54, gubernator.go:203-300) — done for a whole
// generation of requests that already lie in HBM:
//   k_fr_count    per request: XXH64 of the key, the placement's rule -> engine, its rank among the tile's requests of that engine
//                 (stable: arrival order); per tile of 1 024 requests the requests per engine
//   k_fr_scan     one workgroup: where every tile's part of every share starts; places the shares (base[engine]), hands the shares'
//                 sizes to the host and releases the flag it polls
//   k_fr_scatter  request i -> place d = base[engine] + tile_base + rank of the mirror: every engine's share is
//                 contiguous and in arrival order (requests of one key keep their order), the fused pipelines run on the shares as
//                 on any batch; fwd[i] = d.  Keys of ONE width (<= 32 bytes: every front end that formats its keys) travel with
//                 their requests — share j's keys are packed, key_off[d] = d x width, so k_part's speculative key fetch applies —
//                 other keys stay where they are and the share carries offset + length (BatchView.key_len)
//   k_fr_out      answer i = share answer fwd[i]: coalesced writes in arrival order
constexpr uint32_t FR_PER = 4;                     // requests per thread of the copy kernels
constexpr uint32_t FR_TILE = 256 * FR_PER;         // requests per tile
constexpr uint32_t FR_RANK_BITS = 10;              // er[i] = engine << 10 | rank among the tile's requests of that engine
static_assert((1u << FR_RANK_BITS) >= FR_TILE && (MULTI_MEM_MAX << FR_RANK_BITS) <= 65536, "engine and rank share sixteen bits");
constexpr uint32_t FR_SCAN_T = 1024;               // threads of k_fr_scan's workgroups
constexpr uint32_t FR_SCAN_PER = 4;                // tiles per thread, at most
constexpr uint32_t FR_MAX_N = FR_SCAN_T * FR_SCAN_PER * FR_TILE;   // 4 194 304 requests per generation
constexpr uint32_t FR_KEY_COPY_MAX = 32;           // keys of one width up to this many bytes are copied into the shares

struct FrontCtl {                                   // per slot, device memory, zeroed once
    uint32_t tot[MULTI_MEM_MAX];                    // the shares' sizes (a share starts where the shares before it end)
    uint32_t ragged_seq;                            // == seq of the generation: its keys are not of one width (or wider than FR_KEY_COPY_MAX)
    uint32_t pad_[15];
};
// device-visible host memory, one per slot: every word carries the generation's seq in its upper half, so the host needs no fence on the
// device's side to know a word is this generation's (a system-scope release makes the workgroup write back its XCD's whole L2 first:
// measured 66 - 100 us in k_fr_scan behind a generation's kernels); w[k] = seq << 32 | size of engine k's share, w[16] = seq << 32 |
// ragged << 8 | min(key width, 255)
struct FrontHost { unsigned long long w[MULTI_MEM_MAX + 1]; unsigned long long pad_[15]; };
static_assert(sizeof(FrontHost) == 256, "four lines of pinned memory per slot");

struct FrIn {
    uint32_t n, n_engines, max_key, seq;
    // the generation as the caller holds it (HBM, arrival order); burst / created_at / is_owner may be null
    const uint8_t* key_bytes; const uint32_t* key_off;
    // keys as rows instead (the device wire decoder's output): key i = key_bytes + i * key_stride, key_len[i] bytes; key_stride a multiple of 8
    uint32_t key_stride; const uint32_t* key_len;
    const int64_t *hits, *limit, *duration, *burst, *created_at; const uint32_t* behavior; const uint8_t *algorithm, *is_owner;
    // scratch of the slot
    uint16_t* er; uint32_t* tile_cnt; uint32_t* tile_base; FrontCtl* ctl; FrontHost* host;
    // the mirror: the shares, engine after engine
    uint32_t *d_key_off, *d_key_len, *d_fwd; int64_t *d_hits, *d_limit, *d_duration, *d_burst, *d_created_at; uint32_t* d_behavior; uint8_t *d_algorithm, *d_is_owner;
    uint8_t* d_keys;
    RouteRule R;
};
static_assert(sizeof(FrIn) <= 4096, "kernel arguments are limited to 4 KB");
__device__ __forceinline__ uint32_t fr_key_off(const FrIn& A, uint32_t i) { return A.key_stride ? i * A.key_stride : A.key_off[i]; }
__device__ __forceinline__ uint32_t fr_key_len(const FrIn& A, uint32_t i, uint32_t off) { return A.key_stride ? A.key_len[i] : A.key_off[i + 1] - off; }
// the width the generation's keys have if they all have one (the first key's), and where its keys end
__device__ __forceinline__ uint32_t fr_len0(const FrIn& A) { return A.key_stride ? A.key_len[0] : A.key_off[1] - A.key_off[0]; }

// No workgroup waits for another and none takes a ticket: the order between the three steps is the stream's (a device-scope ticket per
// workgroup was 25 ns each, one after the other: 50 us for a generation of 2 048 tiles, and the scans ran behind it — round 6's first form
// took 157 us for 524 288 requests with nothing else on the GPU).
// A tile is FR_TILE = 1 024 requests (round 6's second form; the first had 256): a generation has a quarter of the tiles and scan
// steps, and the runs the copy kernels move (a tile's requests of one engine, consecutive in the share) are four times as long — about
// 85 elements with twelve engines: sectors are used almost fully.  k_fr_scatter / k_fr_out take four requests per thread (request k of
// thread t: tile x 1024 + k x 256 + t — coalesced, every load before the first store).
__global__ __launch_bounds__(FR_TILE) void k_fr_count(FrIn A) {
    // one request per thread, 1 024 threads: the chain offsets -> key -> hot-key list -> slot table is four dependent trips, and what hides
    // them is waves in flight (four requests per thread, a quarter of the waves: 24 us instead of 20 for a million requests)
    __shared__ uint32_t wtot[FR_TILE / 64][MULTI_MEM_MAX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, i = blockIdx.x * FR_TILE + tid;
    if (tid < (FR_TILE / 64) * MULTI_MEM_MAX) (&wtot[0][0])[tid] = 0u;
    uint32_t e = 0xffu;
    if (i < A.n) {
        // keys of one width: the key's words are requested at the place the first two offsets suggest, together with the request's own
        // offsets, and used if those confirm the guess (as k_part does: one dependent trip less)
        // (keys as rows: a row's place is known, its length is the guess; 32 bytes of the row are readable when the rows are that long)
        const uint32_t len0 = fr_len0(A);
        const uint32_t o0 = A.key_stride ? 0u : A.key_off[0], oend = A.key_stride ? A.n * A.key_stride : A.key_off[A.n];
        const uint32_t off_g = A.key_stride ? i * A.key_stride : o0 + i * len0;
        uint64_t kw[4] = {0, 0, 0, 0};
        const bool spec = len0 != 0 && len0 < 32 && (A.key_stride ? A.key_stride >= 32u : (uint64_t)off_g + 32 <= (uint64_t)oend + 8);    // (a packed buffer is readable 8 bytes past the last key)
        if (spec) { const uint8_t* kp = A.key_bytes + off_g; kw[0] = ld_key_word(kp); kw[1] = ld_key_word(kp + 8); kw[2] = ld_key_word(kp + 16); kw[3] = ld_key_word(kp + 24); }
        const uint32_t off = fr_key_off(A, i), len = fr_key_len(A, i, off);
        e = 0;
        // (the hash is not handed on to k_part: a column for it costs the copy kernels 32 B per request, measured -2 % on the routed rate with
        //  k_part hashing less — the pipeline is closer to its transactions than to its instructions: profiles/r06_pass_hash_ab.txt)
        if (A.R.global_engine >= 0 && A.behavior && (A.behavior[i] & 2u)) e = (uint32_t)A.R.global_engine;      // Behavior_GLOBAL: the device's GLOBAL engine
        else if (len != 0 && len <= A.max_key && A.R.n_shards > 1)
            e = route_engine(A.R, (spec && off == off_g && len == len0) ? xxhash64_words4(kw, len, 0) : xxhash64(A.key_bytes + off, len, 0));
        if (e >= A.n_engines) e = 0;
        if (len != len0 || off != off_g || len0 == 0 || len0 > FR_KEY_COPY_MAX) {
            if (__hip_atomic_load(&A.ctl->ragged_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.seq) atomicExch(&A.ctl->ragged_seq, A.seq);
        }
    }
    __syncthreads();
    // the lanes of the wave that go to the same engine: four ballots (one per bit of the engine's number) instead of one per engine;
    // a lane's rank among them follows the arrival order (stable), the group's first lane leaves the group's size
    unsigned long long same = __ballot(true);
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) { const unsigned long long m = __ballot((e >> b) & 1u); same &= ((e >> b) & 1u) ? m : ~m; }
    if (e > 15u) same = 0ull;                                        // (lanes behind the generation's end)
    uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    if (same && rank == 0) wtot[wave][e] = (uint32_t)__popcll(same);
    __syncthreads();
    if (i < A.n) {
        for (uint32_t w = 0; w < wave; ++w) rank += wtot[w][e];
        A.er[i] = (uint16_t)(e << FR_RANK_BITS | rank);
    }
    if (tid < MULTI_MEM_MAX) {
        uint32_t c = 0;
        for (uint32_t w = 0; w < FR_TILE / 64; ++w) c += wtot[w][tid];
        A.tile_cnt[blockIdx.x * MULTI_MEM_MAX + tid] = c;
    }
}

// FOUR workgroups, one per four engines (one 16-byte word of a tile's counts): the exclusive scan of the tiles' counts (thread t takes
// `per` consecutive tiles) and the shares' sizes, which go to the host (it polls for words of this generation) and to ctl->tot — the
// shares' places are the prefix over the engines, which every workgroup of k_fr_scatter adds up for itself (sixteen numbers).
// A dozen live registers, no scratch: with all sixteen engines in registers and the tile loops unrolled the compiler spilled 250
// registers, and a kernel with 1 MB of scratch cost the queue 80 us (measured).
__global__ __launch_bounds__(FR_SCAN_T) void k_fr_scan(FrIn A, uint32_t nt, uint32_t per) {
    __shared__ uint32_t wsum[FR_SCAN_T / 64][4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, q = blockIdx.x;
    const uint32_t t0 = tid * per, t1 = t0 + per < nt ? t0 + per : nt;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
#pragma unroll 1
    for (uint32_t t = t0; t < t1; ++t) { const uint4 v = ((const uint4*)(A.tile_cnt + (size_t)t * MULTI_MEM_MAX))[q]; m0 += v.x; m1 += v.y; m2 += v.z; m3 += v.w; }
    const uint32_t i0 = (uint32_t)wave_incl_scan_i32((int)m0), i1 = (uint32_t)wave_incl_scan_i32((int)m1);
    const uint32_t i2 = (uint32_t)wave_incl_scan_i32((int)m2), i3 = (uint32_t)wave_incl_scan_i32((int)m3);
    if (lane == 63) { wsum[wave][0] = i0; wsum[wave][1] = i1; wsum[wave][2] = i2; wsum[wave][3] = i3; }
    __syncthreads();
    uint32_t e0 = i0 - m0, e1 = i1 - m1, e2 = i2 - m2, e3 = i3 - m3;
    for (uint32_t w = 0; w < wave; ++w) { e0 += wsum[w][0]; e1 += wsum[w][1]; e2 += wsum[w][2]; e3 += wsum[w][3]; }
#pragma unroll 1
    for (uint32_t t = t0; t < t1; ++t) {
        const uint4 v = ((const uint4*)(A.tile_cnt + (size_t)t * MULTI_MEM_MAX))[q];
        ((uint4*)(A.tile_base + (size_t)t * MULTI_MEM_MAX))[q] = make_uint4(e0, e1, e2, e3);
        e0 += v.x; e1 += v.y; e2 += v.z; e3 += v.w;
    }
    if (tid < 4) {
        uint32_t all = 0;
        for (uint32_t w = 0; w < FR_SCAN_T / 64; ++w) all += wsum[w][tid];
        A.ctl->tot[4 * q + tid] = all;
        __hip_atomic_store(&A.host->w[4 * q + tid], (unsigned long long)A.seq << 32 | all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (q == 0 && tid == 0) {
        const uint32_t ragged = __hip_atomic_load(&A.ctl->ragged_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.seq ? 1u : 0u;
        const uint32_t len0 = fr_len0(A);
        __hip_atomic_store(&A.host->w[MULTI_MEM_MAX], (unsigned long long)A.seq << 32 | ragged << 8 | (len0 < 255u ? len0 : 255u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void k_fr_scatter(FrIn A) {
    __shared__ uint32_t sbase[MULTI_MEM_MAX];                        // where each engine's share starts: the prefix over the shares' sizes
    if (threadIdx.x < MULTI_MEM_MAX) {
        uint32_t b = 0;
        for (uint32_t k = 0; k < threadIdx.x; ++k) b += A.ctl->tot[k];
        sbase[threadIdx.x] = b;
    }
    __syncthreads();
    const uint32_t tile = blockIdx.x, i0 = tile * FR_TILE + threadIdx.x;
    const bool packed = A.ctl->ragged_seq != A.seq;                   // keys of one width: they travel with their requests
    uint32_t er[FR_PER], o0[FR_PER], o1[FR_PER], beh[FR_PER]; int64_t hits[FR_PER], limit[FR_PER], duration[FR_PER]; uint8_t algo[FR_PER];
#pragma unroll
    for (int k = 0; k < FR_PER; ++k) {                               // every load of the thread's four requests before the first store
        const uint32_t i = i0 + k * 256u;
        if (i >= A.n) { er[k] = 0xffffffffu; continue; }
        er[k] = A.er[i]; o0[k] = fr_key_off(A, i); o1[k] = o0[k] + fr_key_len(A, i, o0[k]);
        hits[k] = A.hits[i]; limit[k] = A.limit[i]; duration[k] = A.duration[i];
        beh[k] = A.behavior ? A.behavior[i] : 0u; algo[k] = A.algorithm ? A.algorithm[i] : (uint8_t)0;
    }
#pragma unroll
    for (int k = 0; k < FR_PER; ++k) {
        const uint32_t i = i0 + k * 256u;
        if (er[k] == 0xffffffffu) continue;
        const uint32_t e = er[k] >> FR_RANK_BITS, rank = er[k] & ((1u << FR_RANK_BITS) - 1u);
        const uint32_t d = sbase[e] + A.tile_base[tile * MULTI_MEM_MAX + e] + rank;
        A.d_fwd[i] = d < A.n ? d : 0u;
        if (d >= A.n) continue;                                      // (cannot happen: the ranks are a permutation; nothing is written out of bounds)
        A.d_hits[d] = hits[k]; A.d_limit[d] = limit[k]; A.d_duration[d] = duration[k]; A.d_behavior[d] = beh[k]; A.d_algorithm[d] = algo[k];
        if (A.burst) A.d_burst[d] = A.burst[i];
        if (A.created_at) A.d_created_at[d] = A.created_at[i];
        if (A.is_owner) A.d_is_owner[d] = A.is_owner[i];
        if (packed) {
            const uint32_t len = o1[k] - o0[k];
            const uint8_t* src = A.key_bytes + o0[k]; uint8_t* dst = A.d_keys + (size_t)d * len;
            if (len >= 8) {                                          // whole words, the last one overlapping the one before: nothing is written behind the
                uint32_t b = 0;                                      // key, whose neighbour's first bytes are another thread's
                for (; b + 8 <= len; b += 8) { const uint64_t w = ld_key_word(src + b); __builtin_memcpy(dst + b, &w, 8); }
                if (b < len) { const uint64_t w = ld_key_word(src + len - 8); __builtin_memcpy(dst + len - 8, &w, 8); }
            } else {
                const uint64_t w = ld_key_word(src);                 // (a key buffer is readable 8 bytes past its last key)
                for (uint32_t q = 0; q < len; ++q) dst[q] = (uint8_t)(w >> (8 * q));
            }
            A.d_key_off[d] = d * len;
            if (d == A.n - 1) A.d_key_off[A.n] = A.n * len;
        } else {
            A.d_key_off[d] = o0[k]; A.d_key_len[d] = o1[k] - o0[k];
        }
    }
}

struct FrOut {
    uint32_t n; const uint32_t* fwd;
    const uint8_t *d_status, *d_err; const int64_t *d_limit, *d_remaining, *d_reset_time;      // the mirror's answers, in the shares' order
    uint8_t *status, *err; int64_t *limit, *remaining, *reset_time;                           // the caller's result arrays, arrival order
};
__global__ __launch_bounds__(256) void k_fr_out(FrOut A) {
    const uint32_t i0 = blockIdx.x * FR_TILE + threadIdx.x;
    uint32_t d[FR_PER]; uint8_t st[FR_PER], er[FR_PER]; int64_t l[FR_PER], r[FR_PER], t[FR_PER];
#pragma unroll
    for (int k = 0; k < FR_PER; ++k) d[k] = i0 + k * 256u < A.n ? A.fwd[i0 + k * 256u] : 0xffffffffu;
#pragma unroll
    for (int k = 0; k < FR_PER; ++k) {
        if (d[k] == 0xffffffffu) continue;
        st[k] = A.d_status[d[k]]; er[k] = A.d_err[d[k]]; l[k] = A.d_limit[d[k]]; r[k] = A.d_remaining[d[k]]; t[k] = A.d_reset_time[d[k]];
    }
#pragma unroll
    for (int k = 0; k < FR_PER; ++k) {
        if (d[k] == 0xffffffffu) continue;
        const uint32_t i = i0 + k * 256u;
        A.limit[i] = l[k]; A.remaining[i] = r[k]; A.reset_time[i] = t[k]; A.status[i] = st[k]; A.err[i] = er[k];
    }
}

}  // namespace guber
