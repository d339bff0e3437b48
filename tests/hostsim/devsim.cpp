// devsim.cpp — TEST-ONLY: runs the engine's batch-pipeline KERNEL SOURCE (gubernator_amd/csrc/guber_kernels.h and
// guber_kernels_part.h, compiled for the host against tests/hostsim/fakehip) one workgroup at a time, every device thread a
// cooperative fiber, on a table in host memory — so that k_part / k_own / k_eval3 and k_front / k_eval2 can be differential-tested
// against the oracle on a machine without a GPU (tests/test_kernels_devsim.py).  It checks the kernels' LOGIC (grouping, ranks,
// flags, probing, the serial walk); it cannot see memory-ordering bugs or races between workgroups.  Nothing in the product links it.
#include <hip/hip_runtime.h>

#include <functional>
#include <string>

#define GUBER_KERNELS_PIPELINES_ONLY
#include "../../gubernator_amd/csrc/guber_kernels.h"
#include "../../gubernator_amd/csrc/guber_kernels_lru.h"
#include "../../include/guber_gpu.h"

#include "fakehip/fiber_runtime.h"   // the fiber runtime behind fakehip (definitions)

using namespace guber;

// ---- a table and the per-batch work arrays in host memory, laid out as guber_engine_create does -----------------------------
struct DevSim {
    uint64_t slots = 0; uint32_t max_batch = 0, cap = 0;
    uint32_t pmode[8] = {7, 0, 0, 0, 7, 7, 0, 0};                  // Work::pmode: the owner count follows the traffic (guber_kernels_part.h); [4..5]: per batch parity (fuse_ep)
    bool fuse_ep = false; uint64_t part_seq = 0;                   // GUBER_FUSE_EP: every owner-partitioned batch reads its parity's slot, did3 is double-buffered
    Table T{}; Work W{};
    std::vector<DirEntry> dir; std::vector<Bucket> buckets; std::vector<uint8_t> arena; DevCounters ctr{}; std::vector<BlockCounters> bctr;
    std::vector<uint32_t> u32; std::vector<uint8_t> rflags; std::vector<unsigned long long> tilemask, claims, segtiles; std::vector<SegRec> srec;
    std::vector<int64_t> sinv; std::vector<uint16_t> tilerow; std::vector<uint32_t> did2, did3, gse; std::vector<GMsg> gmsg; std::vector<GRec> grec;
    uint32_t epoch16 = 0, fast_batches = 0, fast_prev_n = 0, claims_cells = 0;
    uint64_t seq_next = 1;
    // the bounded cache (guber_kernels_lru.h), driven as guber_engine.hip lru_admit / lru_rebuild drive it
    uint64_t cache_size = 0; LruCtl ctl{}; std::vector<unsigned long long> tstamp; std::vector<uint32_t> tslot; bool tail_ok = true;
    uint64_t admits = 0, applied = 0, rebuilds = 0, cuts = 0, passes = 0; uint32_t split_at = 0;
};
static long long ds_size(DevSim* d) { long long sz = d->ctr.size; for (auto& bc : d->bctr) sz += bc.size_delta; return sz; }
static void ds_rebuild(DevSim* d) {
    std::vector<unsigned long long> st(d->slots); std::vector<uint32_t> sl(d->slots);
    unsigned long long cnt = 0;
    fakehip::launch(dim3((unsigned)((d->slots + 255) / 256)), dim3(256), nullptr, [&] { k_lru_gather(d->T, d->slots, st.data(), sl.data(), d->slots, &cnt); });
    std::vector<uint32_t> ord(cnt);
    for (uint32_t i = 0; i < cnt; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return st[a] < st[b]; });
    d->tstamp.assign(cnt + 16, 0); d->tslot.assign(cnt + 16, 0);
    for (uint32_t i = 0; i < cnt; ++i) { d->tstamp[i] = st[ord[i]]; d->tslot[i] = sl[ord[i]]; }
    d->ctl.cursor = 0; d->ctl.tail_n = cnt;
    d->tail_ok = true; d->rebuilds++;
}
// the launch sequence of lru_admit (guber_engine.hip); returns the status
static uint32_t ds_admit(DevSim* d, const LruKeys& K, uint32_t n, int64_t now) {
    uint32_t cells = 1024; while (cells < 2 * (uint64_t)n) cells <<= 1;
    d->admits++;
    if (d->tstamp.empty()) { d->tstamp.assign(16, 0); d->tslot.assign(16, 0); }
    uint64_t w_len = std::max<uint64_t>(2 * (uint64_t)n, 64);            // (small on purpose: the tests go through LRU_MORE)
    for (int round = 0; round < 64; ++round) {
        if (!d->tail_ok) ds_rebuild(d);
        if (w_len > d->tstamp.size()) w_len = d->tstamp.size();
        const uint32_t W = (uint32_t)w_len, wblocks = (W + 255) / 256;
        const size_t nn = (size_t)n + 1;
        std::vector<unsigned long long> gid(cells, ~0ull), rstamp(nn), zstamp(W + 1);
        std::vector<uint32_t> gfirst(cells, 0xffffffffu), gfirst_ok(cells, 0xffffffffu), gfirst_reset(cells, 0xffffffffu), rfirst(nn), rslot(nn), zslot(W + 1), zwidx(W + 1), qfirst(nn), qrank(nn), qslot(nn), blockcnt(wblocks + 1),
            new_before(nn), touched_before(W + 1);
        std::vector<uint8_t> isnew_at(nn, 0), wflag(W + 1, 0), ztouched(W + 1, 0);
        uint32_t n_risk = 0;
        LruGroups G{gid.data(), gfirst.data(), gfirst_ok.data(), gfirst_reset.data(), cells - 1}; LruRes R{rfirst.data(), rslot.data(), rstamp.data()};
        LruWin Z{zstamp.data(), zslot.data(), zwidx.data()}; LruRisk Q{qfirst.data(), qrank.data(), qslot.data()};
        LruCtl* C = &d->ctl;
        fakehip::launch(dim3(1), dim3(256), nullptr, [&] { k_lru_begin(d->T, C, (uint32_t)d->bctr.size()); });
        if (n) {
            fakehip::launch(dim3((n + 255) / 256), dim3(256), nullptr, [&] { k_lru_probe(d->T, K, n, G); });
            fakehip::launch(dim3(cells / 256), dim3(256), nullptr, [&] { k_lru_keys(d->T, G, C, isnew_at.data(), R); });
        }
        if (W) {
            fakehip::launch(dim3(wblocks), dim3(256), nullptr, [&] { k_lru_win_flag(d->T, d->tstamp.data(), d->tslot.data(), C, W, wflag.data(), blockcnt.data()); });
            fakehip::launch(dim3(1), dim3(1024), nullptr, [&] { k_lru_scan_u32(blockcnt.data(), wblocks, &C->win_valid); });
            fakehip::launch(dim3(wblocks), dim3(256), nullptr, [&] { k_lru_win_emit(d->tstamp.data(), d->tslot.data(), C, W, wflag.data(), blockcnt.data(), Z); });
        }
        fakehip::launch(dim3(1), dim3(1), nullptr, [&] { k_lru_check(C, W, n, d->cache_size); });
        if (W) {
            if (n) {
                fakehip::launch(dim3((n + 255) / 256), dim3(256), nullptr, [&] { k_lru_risk(C, R, Z, ztouched.data(), Q, &n_risk); });
                fakehip::launch(dim3(1), dim3(1024), nullptr, [&] { k_lru_scan_u8(isnew_at.data(), n, new_before.data()); });
            }
            fakehip::launch(dim3(1), dim3(1024), nullptr, [&] { k_lru_scan_u8(ztouched.data(), W, touched_before.data()); });
            if (n) fakehip::launch(dim3((n + 255) / 256), dim3(256), nullptr, [&] { k_lru_decide(d->T, C, d->cache_size, Q, &n_risk, new_before.data(), now); });
            fakehip::launch(dim3(wblocks), dim3(256), nullptr, [&] { k_lru_evict(d->T, C, Z, ztouched.data(), touched_before.data(), now); });
            fakehip::launch(dim3(1), dim3(1), nullptr, [&] { k_lru_end(d->T, C, Z); });
        }
        d->passes++;
        const uint32_t st = C->status;
        if (st == LRU_NONE || st == LRU_APPLIED) { if (st == LRU_APPLIED) d->applied++; return st; }
        if (st == LRU_CUT) { d->cuts++; return st; }
        if (st == LRU_SPLIT) { d->cuts++; d->split_at = C->split_at; return st; }
        if (st == LRU_MORE) { w_len *= 4; continue; }
        if (st == LRU_REBUILD) { d->tail_ok = false; w_len = std::max<uint64_t>(w_len, 2 * (uint64_t)n + C->zone); continue; }
        return 0;
    }
    return 0;
}

extern "C" {
void* ds_create_bounded(uint64_t slots, uint32_t max_batch, int weak_hash, uint64_t cache_size);
void* ds_create(uint64_t slots, uint32_t max_batch, int weak_hash) { return ds_create_bounded(slots, max_batch, weak_hash, 0); }
void* ds_create_bounded(uint64_t slots, uint32_t max_batch, int weak_hash, uint64_t cache_size) {
    DevSim* d = new DevSim();
    d->cache_size = cache_size;
    uint64_t s = 1024; while (s < slots) s <<= 1;
    d->slots = s; d->max_batch = max_batch; d->cap = (max_batch + 255u) & ~255u;
    d->dir.assign(s, DirEntry{0, 0}); d->buckets.resize(s); memset(d->buckets.data(), 0, s * sizeof(Bucket));
    d->arena.assign(std::max<uint64_t>(s * 16, 1 << 20) + 64, 0); d->bctr.assign((d->cap + 255) / 256, BlockCounters{0, 0, 0, 0});
    d->u32.assign((size_t)d->cap * 2, 0); d->rflags.assign(d->cap, 0);
    d->tilemask.assign((size_t)2 * d->cap * FT_WORDS, 0); d->srec.resize(d->cap); memset(d->srec.data(), 0, d->cap * sizeof(SegRec));
    d->sinv.assign(d->cap, 0); d->tilerow.assign((size_t)d->cap * FT_MAX_TILES, 0); d->did2.assign((size_t)2 * d->cap, 0); d->did3.assign((size_t)2 * d->cap, 0);
    d->claims_cells = 1024; while (d->claims_cells < 4 * d->cap) d->claims_cells <<= 1;
    d->claims.assign(d->claims_cells, 0);
    d->gmsg.resize(d->cap); d->grec.resize((size_t)d->cap + d->cap / 2); d->gse.assign((size_t)FT_MAX_TILES * PT_PARTS, 0); d->segtiles.assign((size_t)d->cap * 4, 0);
    memset(&d->T, 0, sizeof(Table)); memset(&d->W, 0, sizeof(Work));
    d->T.dir = d->dir.data(); d->T.buckets = d->buckets.data(); d->T.arena = d->arena.data(); d->T.mask = s - 1;
    d->T.arena_cap = d->arena.size() - 64; d->T.ctr = &d->ctr; d->T.bctr = d->bctr.data();
    d->T.max_probe = (uint32_t)std::min<uint64_t>(s, 1u << 12); d->T.max_key = 1024;
    d->T.hash_mask = weak_hash ? 0x1f80ull : ~0ull;
    d->W.slot = d->u32.data(); d->W.rflags = d->rflags.data();
    d->W.seg_tilemask = d->tilemask.data(); d->W.srec = d->srec.data(); d->W.sinv = d->sinv.data(); d->W.tilerow = d->tilerow.data();
    d->W.claims = d->claims.data();
    d->W.gmsg = d->gmsg.data();
    d->W.grs = (GRecS*)d->grec.data(); d->W.grec = d->grec.data() + d->cap / 2; d->W.gse = d->gse.data(); d->W.segtiles = d->segtiles.data();
    uint32_t lg = 0; while ((1ull << lg) < s) lg++;
    d->W.pshift = lg - 8;
    d->pmode[0] = d->pmode[4] = d->pmode[5] = 7; d->pmode[1] = d->pmode[2] = d->pmode[3] = 0; d->W.pmode = d->pmode;
    return d;
}
void ds_destroy(void* h) { delete (DevSim*)h; }
// owners per batch: bits = 7 | 8 pinned, 0 = follow the traffic (the default); ds_owner_bits: what the next batch will use
void ds_pin_owner_bits(void* h, uint32_t bits) { DevSim* d = (DevSim*)h; d->pmode[3] = bits ? 1u : 0u; if (bits) d->pmode[0] = d->pmode[4] = d->pmode[5] = bits; d->pmode[1] = d->pmode[2] = 0; }
uint32_t ds_owner_bits(void* h) { return ((DevSim*)h)->pmode[0]; }
// batches left with 256 owners (pmode[1]): read, or shorten for a test (n != 0)
uint32_t ds_owner_hold(void* h, uint32_t n) { DevSim* d = (DevSim*)h; if (n) d->pmode[1] = n; return d->pmode[1]; }
void ds_chaos(uint32_t on) { fakehip::S.chaos = on; }
// the order a launch's workgroups run in (0 ascending, 1 descending, 2 a seeded shuffle): the two halves of k_evalpart_multi must
// not care
void ds_block_order(uint32_t order) { fakehip::S.block_order = order; }
// GUBER_FUSE_EP: every owner-partitioned batch of this table reads its owner count from its parity's slot and has packed words
// of its own parity (what guber_engine.hip plan_part does for such an engine)
void ds_fuse_ep(void* h, int on) { DevSim* d = (DevSim*)h; d->fuse_ep = on != 0; d->pmode[4] = d->pmode[5] = d->pmode[0]; }
// the owner bits the batch after the next will use (its parity's slot), for the tests of the mode's one-batch lag
uint32_t ds_owner_slot(void* h, uint32_t parity) { return ((DevSim*)h)->pmode[4 + (parity & 1u)]; }

static int ds_eval_piece(DevSim* d, const BatchView& B, const ResultView& R, int pipeline, int careful);
// pipeline 0: k_front + k_eval2 (careful = the retry round), 1: k_part + k_own + k_eval3.  With a bounded cache (ds_create_bounded)
// a batch that may overflow it goes through the eviction pre-pass and, when it is larger than the cache, in pieces — as
// launch_batch (guber_engine.hip) does.
int ds_eval(void* h, const guber_batch_t* b, guber_result_t* r, int pipeline, int careful) {
    DevSim* d = (DevSim*)h;
    const uint32_t n = b->n;
    if (n == 0) return 0;
    if (n > d->max_batch || n > 65536) return -1;
    BatchView B{n, d->cap, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms};
    B.key_stride = 0; B.key_len = nullptr;
    ResultView R{r->status, r->limit, r->remaining, r->reset_time, r->err};
    if (!d->cache_size || (uint64_t)ds_size(d) + n <= d->cache_size) return ds_eval_piece(d, B, R, pipeline, careful);
    for (uint32_t pos = 0; pos < n;) {
        uint32_t len = n - pos;
        auto slice = [&](uint32_t p, uint32_t l) {
            BatchView S = B; S.n = l; S.key_off = B.key_off + p; S.hits = B.hits + p; S.limit = B.limit + p; S.duration = B.duration + p;
            if (B.burst) S.burst = B.burst + p;
            if (B.created_at) S.created_at = B.created_at + p;
            if (B.algorithm) S.algorithm = B.algorithm + p;
            if (B.behavior) S.behavior = B.behavior + p;
            if (B.is_owner) S.is_owner = B.is_owner + p;
            if (B.greg_expire) S.greg_expire = B.greg_expire + p;
            if (B.greg_duration) S.greg_duration = B.greg_duration + p;
            return S;
        };
        auto keys = [&](const BatchView& S) { LruKeys K{}; K.bytes = S.key_bytes; K.off_p = (const uint8_t*)S.key_off; K.off_stride = 4; K.algorithm = S.algorithm;
                                             K.behavior = S.behavior; K.duration = S.duration; K.greg_duration = (S.greg_expire && S.greg_duration) ? S.greg_duration : nullptr; return K; };
        uint32_t st = ds_admit(d, keys(slice(pos, len)), len, b->now_ms);
        if (st == LRU_CUT) { len = (uint32_t)std::min<uint64_t>(len, d->cache_size); st = ds_admit(d, keys(slice(pos, len)), len, b->now_ms); }
        if (st == LRU_SPLIT) { len = d->split_at ? std::min(len, d->split_at) : 1u; st = ds_admit(d, keys(slice(pos, len)), len, b->now_ms); }   // (launch_batch, guber_engine.hip)
        if (st != LRU_NONE && st != LRU_APPLIED) return -3;
        const int rc = ds_eval_piece(d, slice(pos, len), ResultView{R.status + pos, R.limit + pos, R.remaining + pos, R.reset_time + pos, R.err + pos}, pipeline, careful);
        if (rc) return rc;
        pos += len;
    }
    return 0;
}
// admits, applied, rebuilds, cuts, passes, evicted unexpired (total)
void ds_lru_stats(void* h, unsigned long long* out) {
    DevSim* d = (DevSim*)h;
    out[0] = d->admits; out[1] = d->applied; out[2] = d->rebuilds; out[3] = d->cuts; out[4] = d->passes; out[5] = d->ctr.evictions;
}
static int ds_eval_piece(DevSim* d, const BatchView& B, const ResultView& R, int pipeline, int careful) {
    const uint32_t n = B.n;
    Work W = d->W;
    W.touch = d->seq_next; d->seq_next += n;
    const uint32_t tiles = (n + FT - 1) / FT;
    if (pipeline == 0) {
        W.careful = careful ? 1u : 0u;
        if (++d->epoch16 > 0xffffu) d->epoch16 = 1;
        W.epoch16 = d->epoch16;
        uint32_t cells = 1024; while (cells < 4 * n && cells < d->claims_cells) cells <<= 1;
        W.cmask = cells - 1;
        W.parity = d->fast_batches & 1u;
        W.did = d->did2.data() + (size_t)W.parity * d->cap;
        W.did_prev = d->did2.data() + (size_t)(W.parity ^ 1u) * d->cap;
        W.clear_n = d->fast_prev_n;
        fakehip::launch(dim3(tiles), dim3(FT), nullptr, [&] { k_front(d->T, B, W); });
        EvalArgs A{d->T, B, R, W};
        fakehip::launch(dim3(tiles), dim3(256), &A, [&] { k_eval2(A); });
        d->fast_batches++; d->fast_prev_n = n;
    } else {
        W.did = d->did3.data() + (d->fuse_ep ? (size_t)(d->part_seq & 1) * d->cap : 0);
        W.pmslot = d->fuse_ep ? 1u + (uint32_t)(d->part_seq & 1) : 0u;
        d->part_seq++;
        fakehip::launch(dim3(tiles), dim3(FT), nullptr, [&] { k_part(d->T, B, W); });
        fakehip::launch(dim3(PT_PARTS), dim3(256), nullptr, [&] { k_own(d->T, B, W, tiles); });
        EvalArgs A{d->T, B, R, W};
        fakehip::launch(dim3(tiles), dim3(256), &A, [&] { k_eval3(A); });
        for (auto v : d->segtiles) if (v) return -2;      // the walk's tile maps must be all zero between batches
    }
    return 0;
}
// `rounds` batches for each of nh (<= EP_MAX) tables — batch r of table j is batches[r * nh + j] — through the owner-partitioned
// pipeline as a GUBER_FUSE_EP stream enqueues them (guber_engine.hip launch_group): k_part_multi, k_own_multi, then per further round ONE
// k_evalpart_multi (the previous round's k_eval3 + this round's k_part) and k_own_multi, and the last round's k_eval3_multi.
int ds_eval_stream_ep(void* const* hs, uint32_t nh, const guber_batch_t* batches, guber_result_t* results, uint32_t rounds) {
    if (nh == 0 || nh > (uint32_t)EP_MAX || rounds == 0) return -1;
    struct Planned { BatchView B; ResultView R; Work W; uint32_t tiles; };
    std::vector<Planned> prev, cur(nh);
    for (uint32_t r = 0; r < rounds; ++r) {
        for (uint32_t j = 0; j < nh; ++j) {
            DevSim* d = (DevSim*)hs[j];
            const guber_batch_t* b = &batches[(size_t)r * nh + j]; guber_result_t* rs = &results[(size_t)r * nh + j];
            if (!d->fuse_ep || d->cache_size || b->n == 0 || b->n > d->max_batch || b->n > 65536) return -1;
            Planned& P = cur[j];
            P.B = BatchView{b->n, d->cap, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                            b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms};
            P.B.key_stride = 0; P.B.key_len = nullptr;
            P.R = ResultView{rs->status, rs->limit, rs->remaining, rs->reset_time, rs->err};
            P.W = d->W;
            P.W.touch = d->seq_next; d->seq_next += b->n;
            P.W.did = d->did3.data() + (size_t)(d->part_seq & 1) * d->cap;
            P.W.pmslot = 1u + (uint32_t)(d->part_seq & 1);
            d->part_seq++;
            P.tiles = (b->n + FT - 1) / FT;
        }
        MultiFront MF{}; MF.nb = nh;
        uint32_t tp = 0;
        for (uint32_t j = 0; j < nh; ++j) { tp += cur[j].tiles; MF.end_tile[j] = tp; MF.sub[j] = FrontArgs{((DevSim*)hs[j])->T, cur[j].B, cur[j].W}; }
        if (prev.empty())
            fakehip::launch(dim3(tp), dim3(FT), &MF, [&] { k_part_multi(MF); });
        else {
            MultiEP A{}; A.nb = nh;
            uint32_t te = 0;
            for (uint32_t j = 0; j < nh; ++j) {
                te += prev[j].tiles; A.end_e[j] = te; A.end_p[j] = MF.end_tile[j];
                A.sub[j] = EPSub{EvalArgs{((DevSim*)hs[j])->T, prev[j].B, prev[j].R, prev[j].W}, cur[j].B, cur[j].W.did, cur[j].W.pmslot, 0u};
            }
            fakehip::launch(dim3(te + tp), dim3(256), &A, [&] { k_evalpart_multi(A); });
        }
        fakehip::launch(dim3(nh * PT_PARTS), dim3(256), &MF, [&] { k_own_multi(MF); });
        prev = cur;
    }
    MultiEval ME{}; ME.nb = nh;
    uint32_t te = 0;
    for (uint32_t j = 0; j < nh; ++j) { te += prev[j].tiles; ME.end_tile[j] = te; ME.sub[j] = EvalArgs{((DevSim*)hs[j])->T, prev[j].B, prev[j].R, prev[j].W}; }
    fakehip::launch(dim3(te), dim3(256), &ME, [&] { k_eval3_multi(ME); });
    for (uint32_t j = 0; j < nh; ++j) for (auto v : ((DevSim*)hs[j])->segtiles) if (v) return -2;
    return 0;
}
// the last owner-partitioned batch: (key, tile) groups, of which answered by a 32-byte record (out[2]: unused)
void ds_part_forms(void* h, uint32_t n, unsigned long long* out) {
    DevSim* d = (DevSim*)h;
    out[0] = out[1] = out[2] = 0;
    const uint32_t tiles = (n + FT - 1) / FT;
    for (uint32_t t = 0; t < tiles; ++t) {
        uint32_t groups = 0;
        for (uint32_t p = 0; p < PT_PARTS; ++p) groups += d->gse[(size_t)t * PT_PARTS + p] >> 16;
        out[0] += groups;
        for (uint32_t j = 0; j < groups; ++j) {
            out[1] += d->W.grs[(size_t)t * FT + j].pk & 1ull;
        }
    }
}
// over, hits, misses, size, retries, tags_used
void ds_counters(void* h, long long* out) {
    DevSim* d = (DevSim*)h;
    long long o = (long long)d->ctr.over, hi = (long long)d->ctr.hits, mi = (long long)d->ctr.misses, sz = d->ctr.size;
    for (auto& bc : d->bctr) { o += (long long)bc.over; hi += (long long)bc.hits; mi += (long long)bc.misses; sz += bc.size_delta; }
    out[0] = o; out[1] = hi; out[2] = mi; out[3] = sz; out[4] = (long long)d->ctr.retries; out[5] = (long long)d->ctr.tags_used;
}
// guber_stage_route's two launches (k_route_count + k_route_dest) on host memory: dest[n], counts[16]
int ds_route(const uint8_t* key_bytes, const uint32_t* key_off, const uint32_t* behavior, uint32_t n, uint32_t n_engines, uint32_t max_key,
             uint32_t n_shards, uint32_t per, unsigned long long step, unsigned long long inv_step, unsigned long long inv_sub, const uint16_t* table,
             uint32_t ex_cells, uint32_t ex_n, const unsigned long long* ex_hash, const uint16_t* ex_shard, int global_engine,
             uint32_t* dest, uint32_t* counts) {
    if (n == 0 || n > 65536) return -1;
    const uint32_t tiles = (n + 255u) / 256u;
    std::vector<uint32_t> tile_cnt((size_t)256 * MULTI_MEM_MAX, 0xdeadbeefu), tile_base((size_t)256 * MULTI_MEM_MAX, 0xdeadbeefu);
    std::vector<uint8_t> eng(n + 64, 0xee);
    uint32_t ticket = 0; unsigned int done = 0; uint32_t cnt_out[MULTI_MEM_MAX];
    RouteArgs A{};
    A.n = n; A.n_engines = n_engines; A.max_key = max_key; A.seq = 7;
    A.key_bytes = key_bytes; A.key_off = key_off; A.behavior = behavior;
    A.eng = eng.data(); A.tile_cnt = tile_cnt.data(); A.tile_base = tile_base.data(); A.ticket = &ticket;
    A.dest = dest; A.counts = cnt_out; A.done = &done;
    A.R = RouteRule{n_shards, per, ex_cells, ex_n, global_engine, step, inv_step, inv_sub, table, ex_hash, ex_shard};
    fakehip::launch(dim3(tiles), dim3(256), &A, [&] { k_route_count(A); });
    if (done != 7 || ticket != 0) return -2;
    fakehip::launch(dim3(tiles), dim3(256), &A, [&] { k_route_dest(A); });
    for (int e = 0; e < MULTI_MEM_MAX; ++e) counts[e] = cnt_out[e];
    return 0;
}
}
