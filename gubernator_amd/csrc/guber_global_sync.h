// guber_global_sync.h — GLOBAL behaviour across the GPUs of one node (BASELINE config 5), natively: the flush of the
// reference's globalManager (global.go:91-283) as device kernels + one exchange step over RCCL / xGMI (or device copies
// when several logical ranks share a GPU).  Included at the end of guber_engine.hip (same translation unit: it drives
// the engine's internals).  Python (gubernator_amd/global_sync_dev.py) keeps an independent implementation of the same
// steps as a test harness.
//
// Per sync, on every rank r (its engine = its replica of the GLOBAL keys):
//   A  k_gs_take      pending non-owner hits (global.go:144-187 sendHits) -> fixed-width rows + the owning rank of
//                     each (ReplicatedConsistentHash, ring in LDS) + a count per destination
//      k_gs_partition rows grouped by destination into the send buffer
//   -- counts to the host (the ONE place the host waits in the first half) --
//   B  exchange       every rank sends each owner its slice: ncclSend / ncclRecv pairs in one group (xGMI is point to
//                     point: 7 direct links, no ring), or hipMemcpyAsync between buffers in local mode
//      k_gs_unpack    received rows -> request columns; GLOBAL => DRAIN_OVER_LIMIT (gubernator.go:510-512)
//      launch_batch   the owner applies the hits (IsOwner), which queues the keys for broadcast (global.go:80-84)
//      k_gs_take      the owner's update rows (global.go:217-232); launch_batch with hits = 0 re-reads their state
//      k_gs_items     UpdatePeerGlobals items (gubernator.go:425-459) from that status, failed reads skipped (global.go:246-249)
//   -- item counts to the host --
//   C  all-gather     every rank's items to every other rank (grouped send / recv, or copies)
//      k_gs_item_in + k_items_probe + k_items_commit   AddCacheItem on every other rank (gubernator.go:452)
#pragma once
#include <dlfcn.h>

#include <rccl/rccl.h>

namespace guber {

constexpr uint32_t GS_HIT_EXTRA = 56;     // key_len u32 | behavior u32 | hits limit duration burst created_at | algorithm u8 | role u8 | 6 pad
constexpr uint32_t GS_ITEM_EXTRA = 72;    // key_len u32 | pad u32 | limit duration remaining | remaining_f | stamp burst expire_at | algorithm u8 | status u8 | 6 pad
constexpr uint32_t GS_MAX_WORLD = 64;

struct GsCounters {
    unsigned int n_rows, kept, n_items, bad, retry;
    unsigned int count[GS_MAX_WORLD];     // rows per destination
    unsigned int cursor[GS_MAX_WORLD];
};

__device__ __forceinline__ void gs_copy_words(uint8_t* dst, const uint8_t* src, uint32_t bytes) {
    for (uint32_t q = 0; q < bytes; q += 8) *(uint64_t*)(dst + q) = *(const uint64_t*)(src + q);
}

// pending records of the requested role -> rows (staging, arrival order) + owner of each row
__global__ __launch_bounds__(256) void k_gs_take(Table T, uint32_t role_mask, uint32_t* keep_list, GsCounters* C, uint8_t* rows, uint32_t stride,
                                                 uint32_t cap, uint32_t* owner, const uint64_t* ring_hash, const uint32_t* ring_owner, uint32_t npts,
                                                 int ring_kind, uint32_t world) {
    GUBER_DYN_LDS(smem);
    uint64_t* lh = (uint64_t*)smem;
    for (uint32_t j = threadIdx.x; j < npts; j += 256) lh[j] = ring_hash[j];
    __syncthreads();
    const uint32_t n = T.ctr->gdirty_n;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const uint32_t slot = T.gdirty[j];
        GPend p = T.gpend[slot];
        if (!((role_mask >> p.queued) & 1u)) {            // not asked for: stays pending
            keep_list[atomicAdd(&C->kept, 1u)] = slot;
            continue;
        }
        const uint32_t i = atomicAdd(&C->n_rows, 1u);
        if (i >= cap) { keep_list[atomicAdd(&C->kept, 1u)] = slot; continue; }   // no room this round: stays pending
        const KeyCell* c = &T.buckets[slot].cell;
        const uint32_t len = (uint32_t)(c->w[7] >> 48);
        const uint8_t* src = len <= INLINE_KEY ? (const uint8_t*)c->w : T.arena + c->w[0];
        uint8_t* row = rows + (size_t)i * (stride + GS_HIT_EXTRA);
        for (uint32_t q = 0; q < stride; q += 8) {
            uint64_t v = 0;
            if (q < len) { v = *(const uint64_t*)(src + q); if (q + 8 > len) v &= tail_mask(len - q); }
            *(uint64_t*)(row + q) = v;
        }
        uint8_t* x = row + stride;
        *(uint32_t*)(x + 0) = len; *(uint32_t*)(x + 4) = p.behavior;
        *(int64_t*)(x + 8) = p.hits; *(int64_t*)(x + 16) = p.limit; *(int64_t*)(x + 24) = p.duration;
        *(int64_t*)(x + 32) = p.burst; *(int64_t*)(x + 40) = p.created_at;
        *(uint64_t*)(x + 48) = (uint64_t)p.algorithm | ((uint64_t)p.queued << 8);
        uint32_t own = 0;
        if (world > 1 && npts) {                          // ReplicatedConsistentHash.Get (replicated_hash.go:104-119)
            const uint64_t hh = ring_kind == 1 ? fnv1a_64(row, len) : fnv1_64(row, len);
            uint32_t lo = 0, hi = npts;
            while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (lh[mid] >= hh) hi = mid; else lo = mid + 1; }
            if (lo == npts) lo = 0;
            own = ring_owner[lo];
        }
        owner[i] = own < GS_MAX_WORLD ? own : 0;
        GPend z; __builtin_memset(&z, 0, sizeof(z));
        T.gpend[slot] = z;
    }
}
// rows per destination: per-workgroup histogram in LDS, one global add per workgroup and destination
__global__ __launch_bounds__(256) void k_gs_count(GsCounters* C, const uint32_t* owner, uint32_t cap, uint32_t world) {
    __shared__ uint32_t lh[GS_MAX_WORLD];
    if (threadIdx.x < GS_MAX_WORLD) lh[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t n = C->n_rows < cap ? C->n_rows : cap;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&lh[owner[i]], 1u);
    __syncthreads();
    if (threadIdx.x < world && lh[threadIdx.x]) atomicAdd(&C->count[threadIdx.x], lh[threadIdx.x]);
}
// rows grouped by destination: every workgroup takes a contiguous chunk, reserves its share of each destination's range
// with one global add per destination, and places its rows with LDS counters
constexpr uint32_t GS_CHUNK = 2048;
__global__ __launch_bounds__(256) void k_gs_partition(GsCounters* C, const uint8_t* rows, const uint32_t* owner, uint32_t row_bytes, uint32_t cap,
                                                      uint32_t world, uint8_t* send) {
    __shared__ uint32_t lh[GS_MAX_WORLD], lbase[GS_MAX_WORLD], lcur[GS_MAX_WORLD];
    const uint32_t n = C->n_rows < cap ? C->n_rows : cap;
    for (uint32_t c0 = blockIdx.x * GS_CHUNK; c0 < n; c0 += gridDim.x * GS_CHUNK) {
        if (threadIdx.x < GS_MAX_WORLD) { lh[threadIdx.x] = 0u; lcur[threadIdx.x] = 0u; }
        __syncthreads();
        const uint32_t c1 = c0 + GS_CHUNK < n ? c0 + GS_CHUNK : n;
        for (uint32_t i = c0 + threadIdx.x; i < c1; i += 256) atomicAdd(&lh[owner[i]], 1u);
        __syncthreads();
        if (threadIdx.x < world) {
            uint32_t off = 0;
            for (uint32_t q = 0; q < threadIdx.x; ++q) off += C->count[q];
            lbase[threadIdx.x] = off + (lh[threadIdx.x] ? atomicAdd(&C->cursor[threadIdx.x], lh[threadIdx.x]) : 0u);
        }
        __syncthreads();
        for (uint32_t i = c0 + threadIdx.x; i < c1; i += 256) {
            const uint32_t o = owner[i];
            const uint32_t pos = lbase[o] + atomicAdd(&lcur[o], 1u);
            gs_copy_words(send + (size_t)pos * row_bytes, rows + (size_t)i * row_bytes, row_bytes);
        }
        __syncthreads();
    }
}
// received hit rows -> request columns (the key stays in the row: BatchView.key_stride = row_bytes)
struct GsCols { uint32_t* key_len; int64_t *hits, *limit, *duration, *burst, *created_at; uint32_t* behavior; uint8_t* algorithm; };
__global__ __launch_bounds__(256) void k_gs_unpack(const uint8_t* rows, uint32_t stride, uint32_t n, GsCols O, int drain, int zero_hits) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* x = rows + (size_t)i * (stride + GS_HIT_EXTRA) + stride;
    uint32_t beh = *(const uint32_t*)(x + 4);
    if (drain && (beh & BH_GLOBAL)) beh |= BH_DRAIN_OVER_LIMIT;
    O.key_len[i] = *(const uint32_t*)(x + 0); O.behavior[i] = beh;
    O.hits[i] = zero_hits ? 0 : *(const int64_t*)(x + 8);
    O.limit[i] = *(const int64_t*)(x + 16); O.duration[i] = *(const int64_t*)(x + 24);
    O.burst[i] = *(const int64_t*)(x + 32); O.created_at[i] = *(const int64_t*)(x + 40);
    O.algorithm[i] = x[48];
}
__global__ __launch_bounds__(256) void k_gs_count_retry(const uint8_t* err, uint32_t n, GsCounters* C) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && err[i] == IE_RETRY) atomicAdd(&C->retry, 1u);
}
// UpdatePeerGlobals items (gubernator.go:425-459) from the owner's hits = 0 status; rows whose read failed are skipped
// (global.go:246-249).  Leaky: Remaining = float64(status.Remaining), Burst = Limit, UpdatedAt = now; token: Remaining,
// Status, CreatedAt = now; ExpireAt = status.ResetTime.
__global__ __launch_bounds__(256) void k_gs_items(const uint8_t* rows, uint32_t stride, const GsCounters* Cin, uint32_t cap, ResultView R, int64_t now,
                                                  uint8_t* items, GsCounters* C) {
    const uint32_t n = Cin->n_rows < cap ? Cin->n_rows : cap;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (R.err[i] != 0) continue;
        const uint8_t* row = rows + (size_t)i * (stride + GS_HIT_EXTRA);
        const uint8_t* x = row + stride;
        const uint32_t k = atomicAdd(&C->n_items, 1u);
        uint8_t* out = items + (size_t)k * (stride + GS_ITEM_EXTRA);
        gs_copy_words(out, row, stride);
        uint8_t* y = out + stride;
        const uint8_t algo = x[48];
        const bool leaky = algo == ALGO_LEAKY;
        const int64_t limit = R.limit[i], remaining = R.remaining[i];
        *(uint32_t*)(y + 0) = *(const uint32_t*)(x + 0); *(uint32_t*)(y + 4) = 0;
        *(int64_t*)(y + 8) = limit; *(int64_t*)(y + 16) = *(const int64_t*)(x + 24);
        *(int64_t*)(y + 24) = leaky ? 0 : remaining;
        *(double*)(y + 32) = leaky ? (double)remaining : 0.0;
        *(int64_t*)(y + 40) = now; *(int64_t*)(y + 48) = leaky ? limit : 0; *(int64_t*)(y + 56) = R.reset_time[i];
        *(uint64_t*)(y + 64) = (uint64_t)algo | ((uint64_t)(leaky ? 0 : R.status[i]) << 8);
    }
}
// received item rows -> ItemIn (key referenced inside the row buffer)
__global__ __launch_bounds__(256) void k_gs_item_in(const uint8_t* items, uint32_t stride, uint32_t n, ItemIn* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t rb = stride + GS_ITEM_EXTRA;
    const uint8_t* y = items + (size_t)i * rb + stride;
    Rec s; rec_clear(s);
    const uint8_t algo = y[64], status = y[65];
    s.limit = *(const int64_t*)(y + 8); s.duration = *(const int64_t*)(y + 16); s.stamp = *(const int64_t*)(y + 40);
    s.burst = *(const int64_t*)(y + 48); s.expire_at = *(const int64_t*)(y + 56);
    if (algo == ALGO_TOKEN) { s.remaining = *(const int64_t*)(y + 24); s.burst = 0; s.meta = make_meta(K_TOKEN, status, ALGO_TOKEN); }
    else if (algo == ALGO_LEAKY) { s.remaining = f2bits(*(const double*)(y + 32)); s.meta = make_meta(K_LEAKY, 0, ALGO_LEAKY); }
    else s.meta = make_meta(K_NIL, 0, algo);
    ItemIn o; o.rec = s; o.key_off = i * rb; o.key_len = *(const uint32_t*)(y + 0);
    out[i] = o;
}
__global__ __launch_bounds__(256) void k_gs_count_bad(const uint8_t* res, uint32_t n, GsCounters* C) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && res[i] >= 0xFE) atomicAdd(&C->bad, 1u);
}

}  // namespace guber

// ---- RCCL through dlopen: the library loads (and every non-GLOBAL path works) without RCCL present; in a process that
// already has RCCL mapped (torch) the same copy is used ----
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("GUBER_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm || !*nm) continue;
            void* h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
            if (!h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (h) { api.h = h; break; }
        }
        if (!api.h) return;
#define GS_SYM(field, name) api.field = (decltype(api.field))dlsym(api.h, name)
        GS_SYM(GetUniqueId, "ncclGetUniqueId"); GS_SYM(CommInitRank, "ncclCommInitRank"); GS_SYM(CommInitAll, "ncclCommInitAll");
        GS_SYM(CommDestroy, "ncclCommDestroy"); GS_SYM(GroupStart, "ncclGroupStart"); GS_SYM(GroupEnd, "ncclGroupEnd");
        GS_SYM(Send, "ncclSend"); GS_SYM(Recv, "ncclRecv"); GS_SYM(AllGather, "ncclAllGather"); GS_SYM(GetErrorString, "ncclGetErrorString");
#undef GS_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send &&
                 api.Recv && api.AllGather;
    });
    return &api;
}
}  // namespace

struct GsRank {                 // one local rank of a communicator
    guber_engine* e = nullptr;
    uint32_t rank = 0;
    ncclComm_t comm = nullptr;
    DevBuf<uint8_t> stage, send, recv, items, items_recv, res8, cols8, owner_bytes;
    DevBuf<int64_t> cols64; DevBuf<uint32_t> cols32, owner; DevBuf<GsCounters> ctr; DevBuf<unsigned int> table;
    PinBuf<GsCounters> h_ctr; PinBuf<unsigned int> h_table;
    DevBuf<ItemIn> item_in; DevBuf<uint32_t> islots; DevBuf<uint8_t> iflags, ires, zero8;
    uint32_t cap = 0;
    guber_global_sync_stats_t last{};          // this rank's share of the last sync
    void release() {
        stage.release(); send.release(); recv.release(); items.release(); items_recv.release(); res8.release(); cols8.release(); owner_bytes.release();
        cols64.release(); cols32.release(); owner.release(); ctr.release(); table.release(); h_ctr.release(); h_table.release();
        item_in.release(); islots.release(); iflags.release(); ires.release(); zero8.release();
    }
};
struct guber_comm {
    uint32_t world = 0, stride = 0;
    bool use_rccl = false, local_all = false;       // local_all: every rank of the world lives in this process
    const guber_ring_t* ring = nullptr;
    std::vector<GsRank*> ranks;                     // the local ranks (all of them in local mode, one in per-process mode)
    std::mutex mu;
};

static uint32_t gs_hit_rb(const guber_comm* c) { return c->stride + GS_HIT_EXTRA; }
static uint32_t gs_item_rb(const guber_comm* c) { return c->stride + GS_ITEM_EXTRA; }

static int gs_rank_init(guber_comm* c, GsRank* r) {
    guber_engine* e = r->e;
    if (!e->T.gpend) return fail(GUBER_E_INVALID_ARG, "engine created without GUBER_FLAG_GLOBAL");
    if (hipSetDevice(e->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    r->cap = (uint32_t)std::min<uint64_t>(e->T.gdirty_cap, 1u << 21);   // rows one sync can carry per rank (what does not fit stays queued)
    r->cap = (uint32_t)std::min<uint64_t>(r->cap, std::max<uint64_t>((512ull << 20) / (c->stride + GS_ITEM_EXTRA), 1024));   // <= 512 MiB per row buffer
    int rc = r->ctr.ensure(1) | r->h_ctr.ensure(1) | r->table.ensure((size_t)c->world * (c->world + 1)) | r->h_table.ensure((size_t)c->world * (c->world + 1));
    rc |= r->zero8.ensure(e->max_batch);
    if (rc) return GUBER_E_NOMEM;
    HIPCHK(hipMemsetAsync(r->zero8.p, 0, e->max_batch, e->stream));
    return 0;
}

// grow-only buffers: the sending side (rows taken from this rank's queues, <= cap) and the evaluating side (rows received)
static int gs_ensure_rows(guber_comm* c, GsRank* r, size_t n) {
    n = std::max<size_t>(n, 1024);
    const int rc = r->stage.ensure(n * gs_hit_rb(c) + 64) | r->send.ensure(n * gs_hit_rb(c) + 64) | r->owner.ensure(n) | r->items.ensure(n * gs_item_rb(c) + 64);
    return rc ? GUBER_E_NOMEM : 0;
}
static int gs_ensure_eval(GsRank* r, size_t n) {
    n = std::max<size_t>(n, 1024);
    const int rc = r->cols64.ensure(n * 8) | r->cols32.ensure(n * 2) | r->cols8.ensure(n * 2) | r->res8.ensure(n * 2);
    return rc ? GUBER_E_NOMEM : 0;
}

// one request per row through the engine's batch pipeline, in chunks of max_batch; results into r->cols64[5n..8n) / r->res8
static int gs_eval_rows(guber_comm* c, GsRank* r, const uint8_t* rows, uint32_t n, int64_t now_ms, bool is_owner, bool zero_hits, bool drain) {
    guber_engine* e = r->e;
    if (n == 0) return 0;
    if (gs_ensure_eval(r, n)) return GUBER_E_NOMEM;
    int64_t* c64 = r->cols64.p;
    GsCols O{r->cols32.p, c64, c64 + n, c64 + 2 * (size_t)n, c64 + 3 * (size_t)n, c64 + 4 * (size_t)n, r->cols32.p + n, r->cols8.p};
    hipLaunchKernelGGL(k_gs_unpack, dim3((n + 255) / 256), dim3(256), 0, e->stream, rows, c->stride, n, O, drain ? 1 : 0, zero_hits ? 1 : 0);
    const uint32_t rb = gs_hit_rb(c);
    for (uint32_t lo = 0; lo < n; lo += e->max_batch) {
        const uint32_t m = std::min<uint32_t>(e->max_batch, n - lo);
        BatchView B{m, 0, rows + (size_t)lo * rb, nullptr, O.hits + lo, O.limit + lo, O.duration + lo, O.burst + lo, O.created_at + lo,
                    O.algorithm + lo, O.behavior + lo, is_owner ? nullptr : r->zero8.p, nullptr, nullptr, now_ms, rb, O.key_len + lo};
        ResultView R{r->res8.p + lo, c64 + 5 * (size_t)n + lo, c64 + 6 * (size_t)n + lo, c64 + 7 * (size_t)n + lo, r->res8.p + n + lo};
        const int rc = launch_batch(e, B, R);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_gs_count_retry, dim3((n + 255) / 256), dim3(256), 0, e->stream, r->res8.p + n, n, r->ctr.p);
    return 0;
}

// rare: a batch of the exchange hit a 64-bit hash / fingerprint collision (GUBER_ITEM_E_RETRY): those rows go through the
// host entry point, which runs the careful rounds
static int gs_retry_on_host(guber_comm* c, GsRank* r, const uint8_t* rows, uint32_t n, int64_t now_ms, bool is_owner, bool zero_hits, bool drain) {
    guber_engine* e = r->e;
    const uint32_t rb = gs_hit_rb(c);
    std::vector<uint8_t> h_rows((size_t)n * rb), h_err(n);
    HIPCHK(hipMemcpy(h_rows.data(), rows, h_rows.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h_err.data(), r->res8.p + n, n, hipMemcpyDeviceToHost));
    std::vector<uint32_t> idx;
    for (uint32_t i = 0; i < n; ++i) if (h_err[i] == guber::IE_RETRY) idx.push_back(i);
    const uint32_t m = (uint32_t)idx.size();
    if (!m) return 0;
    std::vector<uint8_t> keys; std::vector<uint32_t> off(m + 1), beh(m); std::vector<int64_t> hits(m), limit(m), dur(m), burst(m), created(m);
    std::vector<uint8_t> algo(m), owner(m, is_owner ? 1 : 0), st(m), er(m); std::vector<int64_t> ol(m), orem(m), ors(m);
    for (uint32_t k = 0; k < m; ++k) {
        const uint8_t* row = h_rows.data() + (size_t)idx[k] * rb; const uint8_t* x = row + c->stride;
        const uint32_t len = *(const uint32_t*)(x + 0);
        off[k] = (uint32_t)keys.size(); keys.insert(keys.end(), row, row + len);
        beh[k] = *(const uint32_t*)(x + 4); if (drain && (beh[k] & guber::BH_GLOBAL)) beh[k] |= guber::BH_DRAIN_OVER_LIMIT;
        hits[k] = zero_hits ? 0 : *(const int64_t*)(x + 8); limit[k] = *(const int64_t*)(x + 16); dur[k] = *(const int64_t*)(x + 24);
        burst[k] = *(const int64_t*)(x + 32); created[k] = *(const int64_t*)(x + 40); algo[k] = x[48];
    }
    off[m] = (uint32_t)keys.size(); keys.resize(keys.size() + 16, 0);
    guber_batch_t b{}; guber_result_t res{};
    b.n = m; b.key_bytes = keys.data(); b.key_off = off.data(); b.hits = hits.data(); b.limit = limit.data(); b.duration = dur.data();
    b.burst = burst.data(); b.created_at = created.data(); b.algorithm = algo.data(); b.behavior = beh.data(); b.is_owner = owner.data(); b.now_ms = now_ms;
    res.status = st.data(); res.limit = ol.data(); res.remaining = orem.data(); res.reset_time = ors.data(); res.err = er.data();
    const int rc = eval_batch_host_locked(e, &b, &res, nullptr);      // (the engine's mutex stays held: see guber_global_sync)
    if (rc) return rc;
    // patch the device-side results (the item builder reads them)
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t i = idx[k];
        HIPCHK(hipMemcpy(r->res8.p + i, &st[k], 1, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(r->res8.p + n + i, &er[k], 1, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(r->cols64.p + 5 * (size_t)n + i, &ol[k], 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(r->cols64.p + 6 * (size_t)n + i, &orem[k], 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(r->cols64.p + 7 * (size_t)n + i, &ors[k], 8, hipMemcpyHostToDevice));
    }
    return 0;
}

static int gs_take(guber_comm* c, GsRank* r, uint32_t role_mask, bool route) {
    guber_engine* e = r->e;
    int rc = 0;
    uint32_t npts = 0; int kind = 0;
    if (route && c->world > 1) {
        rc = ensure_ring_on_device(e, c->ring);
        if (rc) return rc;
        npts = e->ring_npts; kind = guber_ring_kind(c->ring);
    }
    HIPCHK(hipMemsetAsync(r->ctr.p, 0, sizeof(GsCounters), e->stream));
    hipLaunchKernelGGL(k_gs_take, dim3(512), dim3(256), (size_t)npts * 8, e->stream, e->T, role_mask, e->gdirty2.p, r->ctr.p, r->stage.p, c->stride,
                       r->cap, r->owner.p, e->d_ring_h.p, e->d_ring_o.p, npts, kind, route ? c->world : 1u);
    // the kept list becomes the dirty list
    HIPCHK(hipMemcpyAsync(&e->ctr.p->gdirty_n, &r->ctr.p->kept, sizeof(unsigned int), hipMemcpyDeviceToDevice, e->stream));
    std::swap(e->gdirty.p, e->gdirty2.p);
    e->T.gdirty = e->gdirty.p;
    return 0;
}

extern "C" void guber_comm_destroy(guber_comm_t* c) {
    if (!c) return;
    for (GsRank* r : c->ranks) {
        if (r->e) { (void)hipSetDevice(r->e->device); (void)hipStreamSynchronize(r->e->stream); }
        if (r->comm) rccl_api()->CommDestroy(r->comm);        // (RCCL is only ever loaded when a communicator uses it)
        r->release();
        delete r;
    }
    delete c;
}

static int gs_comm_common(guber_comm* c, const guber_ring_t* ring) {
    if (c->world == 0 || c->world > GS_MAX_WORLD) return fail(GUBER_E_INVALID_ARG, "world size must be 1..64");
    c->ring = ring;
    uint32_t mk = 0;
    for (GsRank* r : c->ranks) mk = std::max(mk, r->e->max_key);
    c->stride = (mk + 7u) & ~7u;
    for (GsRank* r : c->ranks) {
        if (r->e->max_key != mk) return fail(GUBER_E_INVALID_ARG, "engines of one communicator need the same max_key_bytes");
        std::lock_guard<std::mutex> lk(r->e->mu);
        int rc = gs_rank_init(c, r);
        if (rc) return rc;
        rc = gs_ensure_rows(c, r, r->cap);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int guber_comm_create_local(guber_engine_t* const* engines, uint32_t n, const guber_ring_t* ring, int use_rccl, guber_comm_t** out) {
    if (!engines || !n || !out || (n > 1 && !ring)) return fail(GUBER_E_INVALID_ARG, "null argument");
    *out = nullptr;
    for (uint32_t i = 0; i < n; ++i) {
        if (!engines[i]) return fail(GUBER_E_INVALID_ARG, "null engine");
        for (uint32_t j = 0; j < i; ++j) if (engines[j] == engines[i]) return fail(GUBER_E_INVALID_ARG, "the same engine listed twice");
    }
    guber_comm* c = new guber_comm();
    c->world = n; c->local_all = true; c->use_rccl = use_rccl != 0 && n > 1;
    for (uint32_t i = 0; i < n; ++i) { GsRank* r = new GsRank(); r->e = engines[i]; r->rank = i; c->ranks.push_back(r); }
    int rc = gs_comm_common(c, ring);
    if (!rc && c->use_rccl) {
        RcclApi* A = rccl_api();
        if (!A->ok) rc = fail(GUBER_E_HIP, "RCCL is not available (librccl.so not found)");
        else {
            std::vector<int> devs(n); std::vector<ncclComm_t> comms(n);
            for (uint32_t i = 0; i < n; ++i) devs[i] = engines[i]->device;
            const ncclResult_t nr = A->CommInitAll(comms.data(), (int)n, devs.data());
            if (nr != ncclSuccess) rc = fail(GUBER_E_HIP, A->GetErrorString ? A->GetErrorString(nr) : "ncclCommInitAll failed");
            else for (uint32_t i = 0; i < n; ++i) c->ranks[i]->comm = comms[i];
        }
    }
    if (rc) { guber_comm_destroy(c); return rc; }
    *out = c;
    return GUBER_OK;
}

extern "C" int guber_comm_unique_id(uint8_t* id128) {
    RcclApi* A = rccl_api();
    if (!id128) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (!A->ok) return fail(GUBER_E_HIP, "RCCL is not available (librccl.so not found)");
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (A->GetUniqueId(&id) != ncclSuccess) return fail(GUBER_E_HIP, "ncclGetUniqueId failed");
    memcpy(id128, &id, 128);
    return GUBER_OK;
}

extern "C" int guber_comm_create_rank(guber_engine_t* e, uint32_t rank, uint32_t world, const uint8_t* id128, const guber_ring_t* ring,
                                      guber_comm_t** out) {
    if (!e || !out || rank >= world || (world > 1 && (!ring || !id128))) return fail(GUBER_E_INVALID_ARG, "bad argument");
    *out = nullptr;
    guber_comm* c = new guber_comm();
    c->world = world; c->local_all = world == 1; c->use_rccl = world > 1;
    GsRank* r = new GsRank(); r->e = e; r->rank = rank; c->ranks.push_back(r);
    int rc = gs_comm_common(c, ring);
    if (!rc && world > 1) {
        RcclApi* A = rccl_api();
        if (!A->ok) rc = fail(GUBER_E_HIP, "RCCL is not available (librccl.so not found)");
        else {
            ncclUniqueId id; memcpy(&id, id128, 128);
            if (hipSetDevice(e->device) != hipSuccess) rc = fail(GUBER_E_HIP, "hipSetDevice");
            else {
                const ncclResult_t nr = A->CommInitRank(&r->comm, (int)world, id, (int)rank);
                if (nr != ncclSuccess) rc = fail(GUBER_E_HIP, A->GetErrorString ? A->GetErrorString(nr) : "ncclCommInitRank failed");
            }
        }
    }
    if (rc) { guber_comm_destroy(c); return rc; }
    *out = c;
    return GUBER_OK;
}

// table[src * world + dst] = rows src sends dst, for every src (local: read from each rank; RCCL: all-gather of the count rows)
static int gs_exchange_counts(guber_comm* c, std::vector<unsigned int>& table, bool items) {
    const uint32_t W = c->world;
    table.assign((size_t)W * W, 0);
    if (c->local_all) {
        for (GsRank* r : c->ranks) {
            HIPCHK(hipSetDevice(r->e->device));
            HIPCHK(hipMemcpyAsync(r->h_ctr.p, r->ctr.p, sizeof(GsCounters), hipMemcpyDeviceToHost, r->e->stream));
        }
        for (GsRank* r : c->ranks) { HIPCHK(hipSetDevice(r->e->device)); HIPCHK(hipStreamSynchronize(r->e->stream)); }
        for (GsRank* r : c->ranks)
            for (uint32_t d = 0; d < W; ++d) table[(size_t)r->rank * W + d] = items ? r->h_ctr.p->n_items : r->h_ctr.p->count[d];
        return 0;
    }
    GsRank* r = c->ranks[0];
    RcclApi* A = rccl_api();
    HIPCHK(hipSetDevice(r->e->device));
    // my row of counts -> everybody (ncclAllGather of W ints), then to the host
    if (items) { /* one number: n_items replicated over the row */
        HIPCHK(hipMemcpyAsync(r->h_ctr.p, r->ctr.p, sizeof(GsCounters), hipMemcpyDeviceToHost, r->e->stream));
        HIPCHK(hipStreamSynchronize(r->e->stream));
        for (uint32_t d = 0; d < W; ++d) r->h_table.p[d] = r->h_ctr.p->n_items;
        HIPCHK(hipMemcpyAsync(r->table.p + (size_t)W * W, r->h_table.p, W * sizeof(unsigned int), hipMemcpyHostToDevice, r->e->stream));
    } else {
        HIPCHK(hipMemcpyAsync(r->table.p + (size_t)W * W, r->ctr.p->count, W * sizeof(unsigned int), hipMemcpyDeviceToDevice, r->e->stream));
    }
    if (A->AllGather(r->table.p + (size_t)W * W, r->table.p, W, ncclUint32, r->comm, r->e->stream) != ncclSuccess) return fail(GUBER_E_HIP, "ncclAllGather failed");
    HIPCHK(hipMemcpyAsync(r->h_table.p, r->table.p, (size_t)W * W * sizeof(unsigned int), hipMemcpyDeviceToHost, r->e->stream));
    HIPCHK(hipMemcpyAsync(r->h_ctr.p, r->ctr.p, sizeof(GsCounters), hipMemcpyDeviceToHost, r->e->stream));
    HIPCHK(hipStreamSynchronize(r->e->stream));
    for (size_t k = 0; k < (size_t)W * W; ++k) table[k] = r->h_table.p[k];
    return 0;
}

// every local rank receives, from every source, the slice meant for it (hits: table[src][me] rows at the source's offset
// for me; items: all of the source's items), into `recv` in source order.  Returns rows received per local rank.
static int gs_exchange(guber_comm* c, const std::vector<unsigned int>& table, bool items, std::vector<uint32_t>& got) {
    const uint32_t W = c->world;
    const uint32_t rb = items ? gs_item_rb(c) : gs_hit_rb(c);
    got.assign(c->ranks.size(), 0);
    // sizes first (receive buffers grow)
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k];
        uint64_t tot = 0;
        for (uint32_t s = 0; s < W; ++s) if (!(items && s == r->rank)) tot += table[(size_t)s * W + r->rank];
        if (tot > 0xffffffffull) return fail(GUBER_E_NOMEM, "GLOBAL exchange too large");
        got[k] = (uint32_t)tot;
        HIPCHK(hipSetDevice(r->e->device));
        DevBuf<uint8_t>& dst = items ? r->items_recv : r->recv;
        if (dst.ensure((size_t)std::max<uint32_t>(got[k], 1) * rb + 64)) return GUBER_E_NOMEM;
    }
    if (c->use_rccl) {
        RcclApi* A = rccl_api();
        if (A->GroupStart() != ncclSuccess) return fail(GUBER_E_HIP, "ncclGroupStart failed");
        for (GsRank* r : c->ranks) {
            const uint8_t* src = items ? r->items.p : r->send.p;
            uint8_t* dstb = items ? r->items_recv.p : r->recv.p;
            size_t soff = 0, roff = 0;
            for (uint32_t p = 0; p < W; ++p) {
                const size_t scnt = items ? table[(size_t)r->rank * W + p] : table[(size_t)r->rank * W + p];
                const size_t rcnt = table[(size_t)p * W + r->rank];
                if (p == r->rank) {
                    if (!items && scnt) HIPCHK(hipMemcpyAsync(dstb + roff * rb, src + soff * rb, scnt * rb, hipMemcpyDeviceToDevice, r->e->stream));
                    if (!items) { soff += scnt; roff += rcnt; }
                    continue;
                }
                if (scnt && A->Send(src + (items ? 0 : soff * rb), scnt * rb, ncclUint8, (int)p, r->comm, r->e->stream) != ncclSuccess) return fail(GUBER_E_HIP, "ncclSend failed");
                if (rcnt && A->Recv(dstb + roff * rb, rcnt * rb, ncclUint8, (int)p, r->comm, r->e->stream) != ncclSuccess) return fail(GUBER_E_HIP, "ncclRecv failed");
                if (!items) soff += scnt;
                roff += rcnt;
            }
        }
        if (A->GroupEnd() != ncclSuccess) return fail(GUBER_E_HIP, "ncclGroupEnd failed");
        return 0;
    }
    // local mode: plain device copies on the RECEIVER's stream (the host has already waited for every sender's pack)
    for (GsRank* r : c->ranks) {
        HIPCHK(hipSetDevice(r->e->device));
        uint8_t* dstb = items ? r->items_recv.p : r->recv.p;
        size_t roff = 0;
        for (uint32_t s = 0; s < W; ++s) {
            if (items && s == r->rank) continue;
            GsRank* sr = c->ranks[s];
            const size_t cnt = table[(size_t)s * W + r->rank];
            size_t soff = 0;
            if (!items) for (uint32_t q = 0; q < r->rank; ++q) soff += table[(size_t)s * W + q];
            const uint8_t* src = items ? sr->items.p : sr->send.p;
            if (cnt) HIPCHK(hipMemcpyAsync(dstb + roff * rb, src + soff * rb, cnt * rb, hipMemcpyDeviceToDevice, r->e->stream));
            roff += cnt;
        }
    }
    return 0;
}

extern "C" int guber_global_sync(guber_comm_t* c, int64_t now_ms, guber_global_sync_stats_t* stats) {
    if (!c) return fail(GUBER_E_INVALID_ARG, "null communicator");
    std::lock_guard<std::mutex> lk(c->mu);
    const auto t0 = std::chrono::steady_clock::now();
    guber_global_sync_stats_t st{};
    // every local rank's engine, locked in ADDRESS order — the rule of every path that holds several engines at once (fused
    // launches, guber_stages_submit, guber_move_items_by_hash) — and held until the tick is over
    std::vector<guber_engine*> held;
    for (GsRank* r : c->ranks) held.push_back(r->e);
    std::sort(held.begin(), held.end());
    for (guber_engine* e : held) { e->mu.lock(); ep_flush_held(e); }
    struct Unlock { std::vector<guber_engine*>& h; ~Unlock() { for (size_t i = h.size(); i-- > 0;) h[i]->mu.unlock(); } } unlock{held};
    const uint32_t W = c->world;
    int rc = 0;
    // ---- A: pending hits -> rows grouped by owner ----
    for (GsRank* r : c->ranks) {
        guber_engine* e = r->e;
        HIPCHK(hipSetDevice(e->device));
        if (now_ms > e->clock_ms) e->clock_ms = now_ms;
        rc = gs_take(c, r, 1u << 1, true); if (rc) return rc;
        hipLaunchKernelGGL(k_gs_count, dim3(256), dim3(256), 0, e->stream, r->ctr.p, r->owner.p, r->cap, W);
        hipLaunchKernelGGL(k_gs_partition, dim3(512), dim3(256), 0, e->stream, r->ctr.p, r->stage.p, r->owner.p, gs_hit_rb(c), r->cap, W, r->send.p);
        HIPCHK(hipGetLastError());
    }
    std::vector<unsigned int> table; std::vector<uint32_t> got;
    rc = gs_exchange_counts(c, table, false); if (rc) return rc;
    for (GsRank* r : c->ranks) {
        r->last = guber_global_sync_stats_t{};
        for (uint32_t d = 0; d < W; ++d) r->last.hits_rows_sent += table[(size_t)r->rank * W + d];
        st.hits_rows_sent += r->last.hits_rows_sent;
    }
    // ---- B: rows to their owners; owners apply, then read back what they have to broadcast ----
    rc = gs_exchange(c, table, false, got); if (rc) return rc;
    std::vector<uint32_t> n_upd(c->ranks.size(), 0);
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k]; guber_engine* e = r->e;
        HIPCHK(hipSetDevice(e->device));
        st.hits_rows_applied += got[k]; r->last.hits_rows_applied = got[k];
        st.bytes_moved += (uint64_t)got[k] * gs_hit_rb(c); r->last.bytes_moved += (uint64_t)got[k] * gs_hit_rb(c);
        rc = gs_eval_rows(c, r, r->recv.p, got[k], now_ms, true, false, true); if (rc) return rc;
    }
    // collisions inside an apply batch (rare): host path for those rows, before the updates are read
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k];
        if (!got[k]) continue;
        HIPCHK(hipSetDevice(r->e->device));
        HIPCHK(hipMemcpyAsync(r->h_ctr.p, r->ctr.p, sizeof(GsCounters), hipMemcpyDeviceToHost, r->e->stream));
    }
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k];
        if (!got[k]) continue;
        HIPCHK(hipSetDevice(r->e->device));
        HIPCHK(hipStreamSynchronize(r->e->stream));
        if (r->h_ctr.p->retry) { st.fallbacks++; rc = gs_retry_on_host(c, r, r->recv.p, got[k], now_ms, true, false, true); if (rc) return rc; }
    }
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k]; guber_engine* e = r->e;
        HIPCHK(hipSetDevice(e->device));
        rc = gs_take(c, r, 1u << 2, false); if (rc) return rc;       // rows into r->stage, count on the device
    }
    // the update rows are evaluated with hits = 0; their number is only known on the device: read it
    rc = gs_exchange_counts(c, table, true);                         // (n_items still 0 here; the call also brings n_rows to the host)
    if (rc) return rc;
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k]; guber_engine* e = r->e;
        HIPCHK(hipSetDevice(e->device));
        n_upd[k] = std::min<uint32_t>(r->h_ctr.p->n_rows, r->cap);
        st.update_rows += n_upd[k]; r->last.update_rows = n_upd[k];
        HIPCHK(hipMemsetAsync(&r->ctr.p->retry, 0, sizeof(unsigned int), e->stream));
        rc = gs_eval_rows(c, r, r->stage.p, n_upd[k], now_ms, false, true, false); if (rc) return rc;
    }
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k];
        if (!n_upd[k]) continue;
        HIPCHK(hipSetDevice(r->e->device));
        HIPCHK(hipMemcpyAsync(r->h_ctr.p, r->ctr.p, sizeof(GsCounters), hipMemcpyDeviceToHost, r->e->stream));
        HIPCHK(hipStreamSynchronize(r->e->stream));
        if (r->h_ctr.p->retry) { st.fallbacks++; rc = gs_retry_on_host(c, r, r->stage.p, n_upd[k], now_ms, false, true, false); if (rc) return rc; }
    }
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k]; guber_engine* e = r->e;
        HIPCHK(hipSetDevice(e->device));
        if (n_upd[k]) {
            const size_t n = n_upd[k];
            int64_t* c64 = r->cols64.p;
            ResultView R{r->res8.p, c64 + 5 * n, c64 + 6 * n, c64 + 7 * n, r->res8.p + n};
            hipLaunchKernelGGL(k_gs_items, dim3(512), dim3(256), 0, e->stream, r->stage.p, c->stride, r->ctr.p, r->cap, R, now_ms, r->items.p, r->ctr.p);
        }
    }
    rc = gs_exchange_counts(c, table, true); if (rc) return rc;
    // ---- C: every rank's items to every other rank; install ----
    rc = gs_exchange(c, table, true, got); if (rc) return rc;
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k]; guber_engine* e = r->e;
        HIPCHK(hipSetDevice(e->device));
        const uint32_t n = got[k];
        st.items_installed += n; r->last.items_installed = n;
        st.bytes_moved += (uint64_t)n * gs_item_rb(c); r->last.bytes_moved += (uint64_t)n * gs_item_rb(c);
        if (!n) continue;
        rc = maintain(e, n, now_ms); if (rc) return rc;
        note_enqueued(e, n);
        if (r->item_in.ensure(n) || r->islots.ensure(n) || r->iflags.ensure(n) || r->ires.ensure(n)) return GUBER_E_NOMEM;
        HIPCHK(hipMemsetAsync(&r->ctr.p->bad, 0, sizeof(unsigned int), e->stream));
        hipLaunchKernelGGL(k_gs_item_in, dim3((n + 255) / 256), dim3(256), 0, e->stream, r->items_recv.p, c->stride, n, r->item_in.p);
        hipLaunchKernelGGL(k_items_probe, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, r->item_in.p, r->items_recv.p, n, r->islots.p, r->iflags.p);
        hipLaunchKernelGGL(k_items_commit, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, r->item_in.p, r->items_recv.p, n, r->islots.p, r->iflags.p, r->ires.p,
                           take_stamps(e, n));
        hipLaunchKernelGGL(k_gs_count_bad, dim3((n + 255) / 256), dim3(256), 0, e->stream, r->ires.p, n, r->ctr.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(r->h_ctr.p, r->ctr.p, sizeof(GsCounters), hipMemcpyDeviceToHost, e->stream));
    }
    for (size_t k = 0; k < c->ranks.size(); ++k) {
        GsRank* r = c->ranks[k]; guber_engine* e = r->e;
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (got[k] && r->h_ctr.p->bad) {
            // in-call hash collision between two received keys (rare): hand those items to the host entry point
            st.fallbacks++;
            const uint32_t n = got[k], rb = gs_item_rb(c);
            std::vector<uint8_t> h_items((size_t)n * rb), h_res(n);
            HIPCHK(hipMemcpy(h_items.data(), r->items_recv.p, h_items.size(), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(h_res.data(), r->ires.p, n, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; ++i) {
                if (h_res[i] < 0xFE) continue;
                const uint8_t* row = h_items.data() + (size_t)i * rb; const uint8_t* y = row + c->stride;
                guber_item_t it{};
                it.algorithm = y[64]; it.status = y[65]; it.key = row; it.key_len = *(const uint32_t*)(y + 0);
                it.limit = *(const int64_t*)(y + 8); it.duration = *(const int64_t*)(y + 16); it.remaining = *(const int64_t*)(y + 24);
                it.remaining_f = *(const double*)(y + 32); it.stamp = *(const int64_t*)(y + 40); it.burst = *(const int64_t*)(y + 48);
                it.expire_at = *(const int64_t*)(y + 56);
                const int rc2 = add_items_locked(e, &it, 1, nullptr);
                if (rc2) return rc2;
            }
        }
    }
    st.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return GUBER_OK;
}

extern "C" int guber_comm_last_stats(guber_comm_t* c, uint32_t local_index, guber_global_sync_stats_t* out) {
    if (!c || !out || local_index >= c->ranks.size()) return fail(GUBER_E_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    *out = c->ranks[local_index]->last;
    return GUBER_OK;
}
