// guber_kernels_route.h — the routing of a pool's front stage on the device (guber_stage_route, guber_engine.hip).
// Included by guber_kernels.h (and, like the batch pipelines, compiled for the host by tests/hostsim/devsim.cpp).
#pragma once

namespace guber {

// ---- which engine a request of a front stage belongs to, decided ON THE DEVICE (guber_stage_route) ---------------------------
// The reference's caller picks the worker of a request from the XXH64 of its HashKey (workers.go:153-155 ComputeHash63,
// :180-184 getWorker); a pool with several shards per GPU generalises that to hash slot -> shard through a table plus a list of
// individually placed hot keys (placement.cpp).  With many callers that is the host's largest per-request cost (the hash, the
// lookup, the counting sort by shard, the rank inside the shard's share), so the callers write their requests in arrival order
// and nothing else, and two launches produce what guber_stage_submit_routed needs:
//   k_route_in     keys, key offsets and behaviors of the stage -> HBM (coalesced)
//   k_route_count  per request: XXH64 of the key, the rule -> engine; per tile of 256 requests the
//                  requests per engine; the LAST workgroup to finish scans the tiles (every engine's share in arrival order),
//                  writes the shares' sizes to the host and releases the flag the host polls
//   k_route_dest   dest[i] = engine << 24 | rank in the engine's share (stable: arrival order), straight into the stage
struct RouteRule {                       // guber_placement's state, as the device applies it (guber_placement_impl.h slot_of / Exceptions::get)
    uint32_t n_shards, per, ex_cells, ex_n; int32_t global_engine;
    unsigned long long step, inv_step, inv_sub;
    const uint16_t* table; const unsigned long long* ex_hash; const uint16_t* ex_shard;
};
struct RouteArgs {
    uint32_t n, n_engines, max_key, seq;
    const uint8_t* key_bytes; const uint32_t* key_off; const uint32_t* behavior;      // the stage's columns (their HBM copies)
    uint8_t* eng; uint32_t* tile_cnt; uint32_t* tile_base; uint32_t* ticket;          // HBM scratch: per request, [tiles][16] twice, the finish counter
    uint32_t* dest;                                                                  // the stage's dest column (host)
    uint32_t* counts; unsigned int* done;                                            // host: the shares' sizes, then the flag (= seq)
    RouteRule R;
};
__device__ __forceinline__ uint32_t route_engine(const RouteRule& R, const unsigned long long h) {
    if (R.ex_n) {
        for (uint32_t i = (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 56) & (R.ex_cells - 1);; i = (i + 1) & (R.ex_cells - 1)) {
            const unsigned long long x = R.ex_hash[i];
            if (x == h) return R.ex_shard[i];
            if (x == 0ull) break;
        }
    }
    const unsigned long long h63 = h >> 1;
    unsigned long long w = __umul64hi(h63, R.inv_step);
    if ((w + 1) * R.step <= h63) ++w;
    if (w >= R.n_shards) w = R.n_shards - 1;
    unsigned long long sub = __umul64hi(h63 - w * R.step, R.inv_sub);
    if (sub >= R.per) sub = R.per - 1;
    return R.table[(uint32_t)(w * R.per + sub)];
}
// the columns the routing reads, brought to HBM first as full coalesced lines (XXH64 straight out of host memory would be a few small
// dependent PCIe reads per request: measured 4x the whole batch's time); the keys stay there for guber_stage_submit_routed
struct RouteIn { const uint4* src[3]; uint4* dst[3]; uint32_t n16[3]; uint32_t nb[3]; };
__global__ __launch_bounds__(256) void k_route_in(RouteIn A) {
    uint32_t b = blockIdx.x, k = 0;
    while (k < 2 && b >= A.nb[k]) { b -= A.nb[k]; ++k; }
    const uint32_t stride = A.nb[k] * 256u;
    for (uint32_t i = b * 256u + threadIdx.x; i < A.n16[k]; i += stride) A.dst[k][i] = A.src[k][i];
}
__global__ __launch_bounds__(256) void k_route_count(RouteArgs A) {
    __shared__ uint32_t cnt[MULTI_MEM_MAX];
    __shared__ uint32_t sc[256][MULTI_MEM_MAX + 1];
    __shared__ uint32_t last;
    const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid;
    if (tid < MULTI_MEM_MAX) cnt[tid] = 0u;
    __syncthreads();
    if (i < A.n) {
        const uint32_t off = A.key_off[i], len = A.key_off[i + 1] - off;
        uint32_t e = 0;
        if (A.R.global_engine >= 0 && (A.behavior[i] & 2u)) e = (uint32_t)A.R.global_engine;      // Behavior_GLOBAL: the device's GLOBAL engine
        else if (len != 0 && len <= A.max_key && A.R.n_shards > 1) e = route_engine(A.R, xxhash64(A.key_bytes + off, len, 0));
        if (e >= A.n_engines) e = 0;
        A.eng[i] = (uint8_t)e;
        atomicAdd(&cnt[e], 1u);
    }
    __syncthreads();
    if (tid < MULTI_MEM_MAX) A.tile_cnt[blockIdx.x * MULTI_MEM_MAX + tid] = cnt[tid];
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(A.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the last workgroup: exclusive scan over the tiles, per engine (<= 256 tiles: a stage holds at most 65 536 requests)
    const uint32_t nt = gridDim.x;
    uint32_t mine[MULTI_MEM_MAX];
#pragma unroll
    for (int e = 0; e < MULTI_MEM_MAX; ++e) { mine[e] = tid < nt ? __hip_atomic_load(&A.tile_cnt[tid * MULTI_MEM_MAX + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; sc[tid][e] = mine[e]; }
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t add[MULTI_MEM_MAX];
#pragma unroll
        for (int e = 0; e < MULTI_MEM_MAX; ++e) add[e] = tid >= d ? sc[tid - d][e] : 0u;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < MULTI_MEM_MAX; ++e) sc[tid][e] += add[e];
        __syncthreads();
    }
    if (tid < nt) {
#pragma unroll
        for (int e = 0; e < MULTI_MEM_MAX; ++e) A.tile_base[tid * MULTI_MEM_MAX + e] = sc[tid][e] - mine[e];
    }
    if (tid < MULTI_MEM_MAX) A.counts[tid] = sc[255][tid];
    if (tid == 0) *A.ticket = 0u;
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(A.done, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void k_route_dest(RouteArgs A) {
    __shared__ uint32_t wtot[4][MULTI_MEM_MAX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, i = blockIdx.x * 256u + tid;
    const uint32_t e = i < A.n ? A.eng[i] : 0xffu;
    uint32_t rank = 0;
    for (uint32_t k = 0; k < A.n_engines; ++k) {
        const unsigned long long m = __ballot(e == k);
        if (e == k) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wave][k] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (i >= A.n) return;
    for (uint32_t w = 0; w < wave; ++w) rank += wtot[w][e];
    A.dest[i] = e << 24 | (A.tile_base[blockIdx.x * MULTI_MEM_MAX + e] + rank);
}

}  // namespace guber
