// guber_front.h — guber_front_*: a generation of requests in ARRIVAL order, resident in HBM -> routed to the GPU's logical shards on
// the device -> evaluated by the fused pipelines -> answered in ARRIVAL order.  Part of guber_engine.hip's translation unit (it uses
// the dispatcher's internals: launch_group, PendSet, the engines' locks).
//
// Reference: WorkerPool.GetRateLimit picks a request's worker from the XXH64 of its HashKey (workers.go:261-289, getWorker :180-184)
// and V1Instance.GetRateLimits answers in request order (gubernator.go:203-300, gubernator.proto:51-54).  guber_eval_batches_routed_dev
// takes batches that somebody has already split by shard and leaves the answers in the shards' order; this is the whole of it.
//
// One generation g (slot = g mod depth):
//   routing stream g mod 2   k_fr_count(g)  k_fr_scan(g)  k_fr_scatter(g)  [event in(g)]
//   engine streams                                             wait in(g) | the shares as batches of the fused pipelines | [event done(g, s)]
//   answers' stream                                                                                              wait done(g, *) | k_fr_out(g) [event out(g)]
// The routing runs ahead of the evaluation (two generations), so the shares' sizes — which the host needs to size the launches and to
// keep the bounded caches' admission exact — are in pinned memory by the time the host asks: it never waits for the GPU in steady state.
#pragma once
#include "guber_kernels_wire.h"    // guber::WireEnc / k_wire_enc: the payload stage's form of the answers' last hop (front_out)

// generations of at most this many requests whose tables share ONE stream go as one pair of launches for all tables (launch_group_mem);
// larger ones keep the owner-partitioned pipeline in groups of four tables, whose advantage grows with the share.
// Measured on the payload stage, eight tables, alternating on one box (profiles/r06_wire_pool.txt): generations of 28 000 requests +16 %,
// of 40 000 +5 ... +10 %, of 57 000 -8 %.
constexpr uint32_t FRONT_ONE_PAIR_MAX = 49152;
struct guber_front {
    std::mutex mu;
    int device = 0;
    std::vector<guber_engine*> eng;
    std::vector<hipStream_t> streams;                 // the engines' distinct streams
    std::vector<int> stream_of;                       // engine -> index into streams
    // ONE stream of the front's own: the routing (k_fr_count, k_fr_scan, k_fr_scatter).  The answers' last hop (k_fr_out) rides on the
    // engines' streams, generation by generation in turn: measured on one box (profiles/r06_front_streams.txt), with it behind the routing
    // on the same stream the routing stream was the pipeline's bottleneck (it also stalled on every generation's evaluation) and the
    // engines' streams idled two fifths of the time: 4.36 -> 5.7 G decisions/s at 8 batches per generation, 5.93 -> 6.2 at 16.  More
    // streams of its own (answers, a second routing stream: GUBER_FRONT_STREAMS=2|3 in the laboratory build) lose 5 - 10 %: the HIP
    // runtime maps streams onto four hardware queues, a fifth stream shares one (and GPU_MAX_HW_QUEUES=8 halves the rate).
    const guber::WireEnc* enc_hook = nullptr;         // set around a front_eval call by the payload stage: k_wire_enc in k_fr_out's place (front_out)
    hipStream_t rs = nullptr, rs2 = nullptr, os = nullptr, last_os = nullptr; int n_own_streams = 1; bool out_on_eval = false; uint32_t out_delay = 0, one_pair_max = FRONT_ONE_PAIR_MAX; bool rs_borrowed = false;
    uint32_t cap = 0, depth = 0, max_key = 0;
    uint32_t seq = 0;
    DevBuf<uint16_t> rt_table, rt_exs; DevBuf<uint64_t> rt_exh; RouteRule rule{}; bool have_rule = false;
    struct Slot {
        DevBuf<uint8_t> mem; CohBuf<FrontHost> host;
        FrIn in{};
        uint8_t *o_status = nullptr, *o_err = nullptr; int64_t *o_limit = nullptr, *o_remaining = nullptr, *o_reset = nullptr;
        hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_a = nullptr; bool out_recorded = false;
        std::vector<std::unique_ptr<EvalHook>> hooks;   // one per engine stream
        CohBuf<uint8_t> h_margs; DevBuf<uint8_t> d_margs; // the argument blocks of a generation that goes as ONE pair of launches (launch_group_mem)
        uint32_t seq = 0, n = 0;
        int64_t gen = -1;                               // the generation the slot holds (-1: free)
        bool dispatched = false;
    };
    std::vector<Slot> slots;
    uint64_t generations = 0, forced_flushes = 0, host_waits = 0; double host_wait_ms = 0;
    bool pre_routed = false;                          // front_route_ahead has routed the next call's first generation already
    // with the first engine's per-kernel timing on (guber_profile_enable): every generation's way through the GPU, first routing kernel's
    // start -> the answers' last hop's end (guber_front_latencies)
    struct GenSpan { hipEvent_t a, b; };
    std::vector<GenSpan> gen_spans;
};

static size_t front_col(size_t bytes) { return (bytes + 63) & ~(size_t)63; }
// a generation as the front's internals see it: the caller's guber_batch_t, or columns whose keys are rows (the device wire decoder's output)
struct FrontGen { guber_batch_t b; uint32_t key_stride = 0; const uint32_t* key_len = nullptr; };

extern "C" void guber_front_destroy(guber_front_t* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    for (hipStream_t st : {f->rs, f->rs2, f->os}) if (st) (void)hipStreamSynchronize(st);
    for (auto st : f->streams) (void)hipStreamSynchronize(st);
    for (auto& s : f->slots) {
        s.mem.release(); s.host.release(); s.h_margs.release(); s.d_margs.release();
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
        for (auto& h : s.hooks) if (h && h->ev) (void)hipEventDestroy(h->ev);
    }
    f->rt_table.release(); f->rt_exs.release(); f->rt_exh.release();
    if (f->rs2 && f->rs2 != f->rs) (void)hipStreamDestroy(f->rs2);
    if (f->os && f->os != f->rs) (void)hipStreamDestroy(f->os);
    if (f->rs && !f->rs_borrowed) (void)hipStreamDestroy(f->rs);
    delete f;
}

static int front_set_rule(guber_front* f, const guber_route_rule_t* rule) {
    if (rule->n_shards == 0 || rule->per == 0 || !rule->table || rule->n_shards > 4096 || (rule->ex_cells & (rule->ex_cells - 1)) ||
        (rule->ex_n && (!rule->ex_hash || !rule->ex_shard || rule->ex_n >= rule->ex_cells)))
        return fail(GUBER_E_INVALID_ARG, "malformed route rule");
    const size_t slots = (size_t)rule->n_shards * rule->per, cells = rule->ex_cells ? rule->ex_cells : 1;
    if (f->rt_table.ensure(slots) || f->rt_exh.ensure(cells) || f->rt_exs.ensure(cells)) return GUBER_E_NOMEM;
    hipError_t he = hipStreamSynchronize(f->rs);                     // (launches still reading the previous rule)
    if (he == hipSuccess && f->rs2) he = hipStreamSynchronize(f->rs2);
    if (he == hipSuccess) he = hipMemcpy(f->rt_table.p, rule->table, slots * 2, hipMemcpyHostToDevice);
    if (he == hipSuccess && rule->ex_n) he = hipMemcpy(f->rt_exh.p, rule->ex_hash, cells * 8, hipMemcpyHostToDevice);
    if (he == hipSuccess && rule->ex_n) he = hipMemcpy(f->rt_exs.p, rule->ex_shard, cells * 2, hipMemcpyHostToDevice);
    if (he != hipSuccess) { f->have_rule = false; return fail(GUBER_E_HIP, "guber_front: rule upload", he); }
    f->rule = RouteRule{rule->n_shards, rule->per, rule->ex_cells, rule->ex_n, rule->global_engine, rule->step, rule->inv_step, rule->inv_sub,
                        f->rt_table.p, (const unsigned long long*)f->rt_exh.p, f->rt_exs.p};
    f->have_rule = true;
    return 0;
}

extern "C" int guber_front_create(guber_engine_t* const* engines, uint32_t n_engines, const guber_route_rule_t* rule, uint32_t max_n,
                                  uint32_t depth, guber_front_t** out) {
    if (!engines || !out || !n_engines || n_engines > (uint32_t)MULTI_MEM_MAX) return fail(GUBER_E_INVALID_ARG, "1 .. 16 engines per front");
    if (n_engines > 1 && !rule) return fail(GUBER_E_INVALID_ARG, "a front over several engines needs the placement's rule");
    if (max_n == 0 || max_n > FR_MAX_N) return fail(GUBER_E_BATCH_TOO_LARGE, "a generation holds at most 4 194 304 requests");
    if (depth == 0) depth = 4;
    if (depth < 3 || depth > 16) return fail(GUBER_E_INVALID_ARG, "3 .. 16 generations in flight");
    std::unique_ptr<guber_front, void (*)(guber_front*)> f(new guber_front(), [](guber_front* p) { guber_front_destroy(p); });
    for (uint32_t j = 0; j < n_engines; ++j) {
        guber_engine* e = engines[j];
        if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
        if (e->device != engines[0]->device) return fail(GUBER_E_INVALID_ARG, "the engines of a front live on one device");
        for (uint32_t q = 0; q < j; ++q) if (engines[q] == e) return fail(GUBER_E_INVALID_ARG, "an engine twice in one front");
        f->eng.push_back(e);
        int si = -1;
        for (size_t q = 0; q < f->streams.size(); ++q) if (f->streams[q] == e->stream) si = (int)q;
        if (si < 0) { si = (int)f->streams.size(); f->streams.push_back(e->stream); }
        f->stream_of.push_back(si);
        f->max_key = j == 0 ? e->max_key : std::min(f->max_key, e->max_key);
    }
    f->device = engines[0]->device; f->cap = max_n; f->depth = depth;
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamCreateWithFlags(&f->rs, hipStreamNonBlocking));
    f->n_own_streams = 1;
    if (const char* v = guber_lab_env("GUBER_FRONT_STREAMS")) f->n_own_streams = std::max(1, std::min(3, atoi(v)));
    f->out_on_eval = true;
    if (const char* v = guber_lab_env("GUBER_FRONT_OUT_ON_EVAL")) f->out_on_eval = atoi(v) != 0;
    if (const char* v = guber_lab_env("GUBER_FRONT_OUT_DELAY")) f->out_delay = (uint32_t)std::max(0, std::min(2, atoi(v)));
    if (const char* v = guber_lab_env("GUBER_FRONT_ONE_PAIR_MAX")) f->one_pair_max = (uint32_t)strtoul(v, nullptr, 10);
    f->rs2 = f->os = f->rs;
    if (f->n_own_streams >= 2) HIPCHK(hipStreamCreateWithFlags(&f->os, hipStreamNonBlocking));
    if (f->n_own_streams >= 3) HIPCHK(hipStreamCreateWithFlags(&f->rs2, hipStreamNonBlocking));
    if (rule) { const int rc = front_set_rule(f.get(), rule); if (rc) return rc; }
    else f->rule = RouteRule{1, 1, 0, 0, -1, 0, 0, 0, nullptr, nullptr, nullptr};      // one engine: everything is its share
    const size_t cap = max_n, tiles = (cap + FR_TILE - 1) / FR_TILE;
    f->slots.resize(depth);
    for (auto& s : f->slots) {
        // the mirror: request columns, where each request went, the answers in the shares' order, the keys (<= FR_KEY_COPY_MAX bytes each
        // when they travel), then the routing's scratch
        const size_t bytes = 3 * front_col(cap * 4 + 4) + 5 * front_col(cap * 8) + front_col(cap * 4) + 2 * front_col(cap) +   // requests
                             3 * front_col(cap * 8) + 2 * front_col(cap) +                                                  // answers
                             front_col(cap * FR_KEY_COPY_MAX + 64) +                                                         // keys
                             front_col(cap * 2) + 2 * front_col(tiles * MULTI_MEM_MAX * 4) + front_col(sizeof(FrontCtl));
        if (s.mem.ensure(bytes) || s.host.ensure(1)) return GUBER_E_NOMEM;
        if (f->streams.size() == 1 && n_engines > 1 && (s.h_margs.ensure(sizeof(MultiArgsMem)) || s.d_margs.ensure(sizeof(MultiArgsMem)))) return GUBER_E_NOMEM;
        uint8_t* p = s.mem.p;
        FrIn& A = s.in;
        A.d_key_off = (uint32_t*)p; p += front_col(cap * 4 + 4); A.d_key_len = (uint32_t*)p; p += front_col(cap * 4 + 4); A.d_fwd = (uint32_t*)p; p += front_col(cap * 4 + 4);
        A.d_hits = (int64_t*)p; p += front_col(cap * 8); A.d_limit = (int64_t*)p; p += front_col(cap * 8); A.d_duration = (int64_t*)p; p += front_col(cap * 8);
        A.d_burst = (int64_t*)p; p += front_col(cap * 8); A.d_created_at = (int64_t*)p; p += front_col(cap * 8);
        A.d_behavior = (uint32_t*)p; p += front_col(cap * 4); A.d_algorithm = p; p += front_col(cap); A.d_is_owner = p; p += front_col(cap);
        s.o_limit = (int64_t*)p; p += front_col(cap * 8); s.o_remaining = (int64_t*)p; p += front_col(cap * 8); s.o_reset = (int64_t*)p; p += front_col(cap * 8);
        s.o_status = p; p += front_col(cap); s.o_err = p; p += front_col(cap);
        A.d_keys = p; p += front_col(cap * FR_KEY_COPY_MAX + 64);
        A.er = (uint16_t*)p; p += front_col(cap * 2);
        A.tile_cnt = (uint32_t*)p; p += front_col(tiles * MULTI_MEM_MAX * 4); A.tile_base = (uint32_t*)p; p += front_col(tiles * MULTI_MEM_MAX * 4);
        A.ctl = (FrontCtl*)p; p += front_col(sizeof(FrontCtl));
        A.host = s.host.p;
        memset((void*)s.host.p, 0, sizeof(FrontHost));
        HIPCHK(hipMemsetAsync(A.ctl, 0, sizeof(FrontCtl), f->rs));
        HIPCHK(hipMemsetAsync(A.d_keys, 0, front_col(cap * FR_KEY_COPY_MAX + 64), f->rs));   // (the kernels read keys as 8-byte words: the bytes behind the last key are defined)
        HIPCHK(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
        for (size_t q = 0; q < f->streams.size(); ++q) {
            s.hooks.emplace_back(new EvalHook());
            HIPCHK(hipEventCreateWithFlags(&s.hooks.back()->ev, hipEventDisableTiming));
            s.hooks.back()->st = f->streams[q];
        }
    }
    HIPCHK(hipStreamSynchronize(f->rs));
    *out = f.release();
    return GUBER_OK;
}

// The routing on a stream of the caller's choice — the engines' own, when they share one — instead of the front's (before the front's first
// generation).  For a caller that needs the hardware queue the routing stream would take: the runtime has four that run at full speed
// (profiles/r06_wire_pool_hw_queues.txt: a fifth quarters the payload stage's rate), and a payload stage wants two for its decodes.
static int front_route_on(guber_front* f, hipStream_t st) {
    std::lock_guard<std::mutex> lk(f->mu);
    if (f->generations || f->pre_routed || f->n_own_streams != 1) return fail(GUBER_E_INVALID_ARG, "guber_front: the routing stream is chosen before the first generation");
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamSynchronize(f->rs));
    HIPCHK(hipStreamDestroy(f->rs));
    f->rs = f->rs2 = f->os = st; f->rs_borrowed = true;
    return 0;
}

extern "C" int guber_front_set_rule(guber_front_t* f, const guber_route_rule_t* rule) {
    if (!f || !rule) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(f->mu);
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    return front_set_rule(f, rule);
}

// k_fr_count, k_fr_scan and k_fr_scatter of generation g on its routing stream
static int front_route(guber_front* f, guber_front::Slot& s, const FrontGen* fg, int64_t gen) {
    const guber_batch_t* b = &fg->b;
    hipStream_t rs = (gen & 1) ? f->rs2 : f->rs;
    // the slot's previous generation has left it: its answers' way home read what this routing writes
    if (s.out_recorded) { HIPCHK(hipStreamWaitEvent(rs, s.ev_out, 0)); s.out_recorded = false; }
    s.gen = gen; s.n = b->n; s.dispatched = false;
    s.seq = ++f->seq ? f->seq : ++f->seq;
    for (auto& h : s.hooks) { h->outstanding.store(1); h->recorded.store(false); }      // (1: the dispatcher's own hold until the generation's groups are out)
    if (b->n == 0) return 0;
    FrIn& A = s.in;
    A.n = b->n; A.n_engines = (uint32_t)f->eng.size(); A.max_key = f->max_key; A.seq = s.seq;
    A.key_bytes = b->key_bytes; A.key_off = b->key_off; A.key_stride = fg->key_stride; A.key_len = fg->key_len; A.hits = b->hits; A.limit = b->limit; A.duration = b->duration;
    A.burst = b->burst; A.created_at = b->created_at; A.behavior = b->behavior; A.algorithm = b->algorithm; A.is_owner = b->is_owner;
    A.R = f->rule;
    const uint32_t tiles = (b->n + FR_TILE - 1u) / FR_TILE;
    guber_engine* e0 = f->eng[0];
    std::unique_lock<std::mutex> pl(e0->mu, std::defer_lock);       // (the per-kernel timing's spans and events belong to the first engine)
    if (e0->profiling) pl.lock();
    s.ev_a = nullptr;
    if (e0->profiling) { s.ev_a = e0->get_event(); (void)hipEventRecord(s.ev_a, rs); }
    e0->span_begin(KT_FR_COUNT, b->n, rs);
    hipLaunchKernelGGL(k_fr_count, dim3(tiles), dim3(FR_TILE), 0, rs, A);
    e0->span_end();
    e0->span_begin(KT_FR_SCAN, b->n, rs);
    hipLaunchKernelGGL(k_fr_scan, dim3(MULTI_MEM_MAX / 4), dim3(FR_SCAN_T), 0, rs, A, tiles, (tiles + FR_SCAN_T - 1) / FR_SCAN_T);
    e0->span_end();
    e0->span_begin(KT_FR_SCATTER, b->n, rs);
    hipLaunchKernelGGL(k_fr_scatter, dim3(tiles), dim3(256), 0, rs, A);
    e0->span_end();
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    HIPCHK(hipEventRecord(s.ev_in, rs));
    return 0;
}

// the answers of the slot's generation home, in arrival order (every evaluation of the generation has been launched)
static int front_out(guber_front* f, guber_front::Slot& s, guber_result_t* r) {
    if (s.n == 0) return 0;
    // (out_on_eval: the answers' last hop rides on one of the engines' streams, generation by generation in turn — those are idle two fifths
    //  of the time, the routing stream is the pipeline's bottleneck)
    hipStream_t os = f->out_on_eval ? f->streams[(size_t)s.gen % f->streams.size()] : f->os;
    f->last_os = os;
    for (auto& h : s.hooks) if (h->st != os) HIPCHK(hipStreamWaitEvent(os, h->ev, 0));
    FrOut O{};
    O.n = s.n; O.fwd = s.in.d_fwd;
    O.d_status = s.o_status; O.d_err = s.o_err; O.d_limit = s.o_limit; O.d_remaining = s.o_remaining; O.d_reset_time = s.o_reset;
    O.status = r->status; O.err = r->err; O.limit = r->limit; O.remaining = r->remaining; O.reset_time = r->reset_time;
    guber_engine* e0 = f->eng[0];
    std::unique_lock<std::mutex> pl(e0->mu, std::defer_lock);
    if (e0->profiling) pl.lock();
    e0->span_begin(KT_FR_OUT, s.n, os);
    if (f->enc_hook) {
        // the payload stage (guber_wire_pool.h): the answers' last hop is the encoder — every RPC's GetRateLimitsResp bytes from the shares, through fwd
        guber::WireEnc E = *f->enc_hook;
        E.fwd = O.fwd; E.d_status = O.d_status; E.d_err = O.d_err; E.d_limit = O.d_limit; E.d_remaining = O.d_remaining; E.d_reset = O.d_reset_time;
        hipLaunchKernelGGL(guber::k_wire_enc, dim3(E.nrpc), dim3(guber::WE_T), 0, os, E);
    } else
    hipLaunchKernelGGL(k_fr_out, dim3((s.n + FR_TILE - 1u) / FR_TILE), dim3(256), 0, os, O);
    e0->span_end();
    if (s.ev_a) { hipEvent_t evb = e0->get_event(); (void)hipEventRecord(evb, os); f->gen_spans.push_back({s.ev_a, evb}); s.ev_a = nullptr; }
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    if (os != f->rs) { HIPCHK(hipEventRecord(s.ev_out, os)); s.out_recorded = true; }
    return 0;
}

static bool front_evals_launched(const guber_front::Slot& s) {
    for (auto& h : s.hooks) if (!h->recorded.load(std::memory_order_acquire)) return false;
    return true;
}

// gens[k] -> results[k], k = 0 .. count-1: every pointer inside is a DEVICE pointer, requests in arrival order, answers in arrival
// order.  Asynchronous: returns when everything is enqueued (guber_front_synchronize waits).  The caller's arrays must stay valid and
// their contents ready (produced before the call, on any stream the caller has synchronised with) until then.
static int front_eval(guber_front* f, const FrontGen* gens, guber_result_t* results, uint32_t count, uint32_t* done, hipEvent_t after = nullptr);
extern "C" int guber_front_eval_dev(guber_front_t* f, const guber_batch_t* gens, guber_result_t* results, uint32_t count, uint32_t* done) {
    if (done) *done = 0;
    if (!f || (count && (!gens || !results))) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::vector<FrontGen> g(count);
    for (uint32_t k = 0; k < count; ++k) {
        const int rc = check_batch_args(&gens[k], &results[k]);
        if (rc) return rc;
        g[k].b = gens[k];
    }
    return front_eval(f, g.data(), results, count, done);
}
// `after`: recorded behind the last generation's answers (on the stream their last hop ran on: the answers of a call leave in order)
static int front_eval(guber_front* f, const FrontGen* gens, guber_result_t* results, uint32_t count, uint32_t* done, hipEvent_t after) {
    if (done) *done = 0;
    for (uint32_t k = 0; k < count; ++k) {
        if (gens[k].b.n > f->cap) return fail(GUBER_E_BATCH_TOO_LARGE, "generation larger than the front was created for");
        if (gens[k].b.greg_expire || gens[k].b.greg_duration) return fail(GUBER_E_INVALID_ARG, "a front takes its calendar intervals from the device");
        if (gens[k].key_stride & 7u) return fail(GUBER_E_INVALID_ARG, "key rows are a multiple of 8 bytes apart");
        guber_result_t* r = &results[k];
        r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
    }
    std::lock_guard<std::mutex> lk(f->mu);
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    const uint32_t ne = (uint32_t)f->eng.size(), D = f->depth, ahead = D - 2;
    PendSet pendset;
    bool any_ep = false;
    for (auto* e : f->eng) any_ep = any_ep || e->fuse_ep;
    PendSet* const ps = any_ep ? &pendset : nullptr;
    struct Dispatching { bool on; Dispatching(bool o) : on(o) { if (on) ++tl_ep_dispatcher; } ~Dispatching() { if (on) --tl_ep_dispatcher; } } dispatching(any_ep);
    uint32_t next_route = 0, next_out = 0, enq = 0;
    int rc = 0;
    uint64_t fp[4] = {0, 0, 0, 0};                                  // GUBER_DISPATCH_PROFILE: ns spent enqueueing the routing, waiting for the shares' sizes, dispatching, on the answers
    struct FpSpan { uint64_t* a; uint64_t t0; explicit FpSpan(uint64_t* x) : a(x), t0(dp_now()) {} ~FpSpan() { if (g_dprof) *a += dp_now() - t0; } };
    // (the slots go round across calls: a caller that hands over one generation per call — the payload stage — gets the routing of its next
    //  generation beside the evaluation of the last one, as a caller that hands over sixteen at once does)
    const uint64_t g0 = f->generations;
    auto slot_of = [&](uint32_t k) -> guber_front::Slot& { return f->slots[(g0 + k) % D]; };
    // the answers of every generation whose evaluations have all been launched go home, oldest first
    auto drain_outs = [&](uint32_t upto, bool force) -> int {
        while (next_out < upto) {
            guber_front::Slot& s = slot_of(next_out);
            if (!s.dispatched) break;
            if (s.n && !front_evals_launched(s)) {
                if (!force) break;
                f->forced_flushes++;
                const int r2 = pendset.flush_all();
                if (r2) return r2;
                if (!front_evals_launched(s)) return fail(GUBER_E_HIP, "guber_front: an evaluation was not launched");
            }
            const int r3 = front_out(f, s, &results[next_out]);
            if (r3) return r3;
            ++next_out;
            if (done) *done = next_out;
        }
        return 0;
    };
    std::vector<std::vector<GroupItem>> fifo(ne);
    std::vector<EvalHook*> hook_of(ne, nullptr);
    for (uint32_t k = 0; k < count && !rc; ++k) {
        // the routing runs ahead; a slot is routed into again only after its previous generation's answers have left it
        while (next_route < count && next_route <= k + ahead && !rc) {
            if (next_route >= D) { rc = drain_outs(next_route - D + 1, true); if (rc) break; }
            if (next_route == 0 && f->pre_routed) f->pre_routed = false;          // (front_route_ahead did it: same slot, same generation number)
            else { FpSpan sp(&fp[0]); rc = front_route(f, slot_of(next_route), &gens[next_route], (int64_t)f->generations + next_route); }
            ++next_route;
        }
        if (rc) break;
        guber_front::Slot& s = slot_of(k);
        const guber_batch_t* b = &gens[k].b;
        if (s.n) {
            // the shares' sizes (in pinned memory since k_fr_scan: normally long there)
            auto reported = [&]() {
                for (int q = 0; q <= MULTI_MEM_MAX; ++q)
                    if ((uint32_t)(__atomic_load_n((volatile unsigned long long*)&s.host.p->w[q], __ATOMIC_ACQUIRE) >> 32) != s.seq) return false;
                return true;
            };
            if (!reported()) {
                FpSpan sp(&fp[1]);
                const auto t0 = std::chrono::steady_clock::now();
                f->host_waits++;
                uint32_t spins = 0;
                while (!reported()) {
                    if (++spins > 2000) {
                        std::this_thread::yield();
                        if ((spins & 0x3ffu) == 0 && hipStreamQuery((s.gen & 1) ? f->rs2 : f->rs) != hipErrorNotReady && !reported()) {
                            rc = fail(GUBER_E_HIP, "guber_front: the routing of a generation did not report"); break;
                        }
                    }
                }
                f->host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rc) break;
            }
            uint32_t counts[MULTI_MEM_MAX];
            for (int q = 0; q < MULTI_MEM_MAX; ++q) counts[q] = (uint32_t)s.host.p->w[q];
            const bool packed = !((uint32_t)s.host.p->w[MULTI_MEM_MAX] >> 8 & 1u);
            for (auto st : f->streams) { if (hipStreamWaitEvent(st, s.ev_in, 0) != hipSuccess) { rc = fail(GUBER_E_HIP, "hipStreamWaitEvent"); break; } }
            if (rc) break;
            uint32_t base = 0, total = 0;
            for (uint32_t j = 0; j < ne; ++j) {
                fifo[j].clear();
                hook_of[j] = s.hooks[f->stream_of[j]].get();
                const uint32_t nj = counts[j];
                total += nj;
                guber_engine* e = f->eng[j];
                // a share larger than the engine's pipelines take goes in pieces of EQUAL size (whole tiles): the rounds of a generation then
                // carry the same load (65 536 + a rest would make the second round a third of the first)
                const uint32_t cap_e = std::max<uint32_t>(1u, std::min<uint32_t>(e->fast_cap ? e->fast_cap : e->max_batch, e->max_batch));
                const uint32_t np = (nj + cap_e - 1) / cap_e;
                uint32_t piece = np > 1 ? ((nj + np - 1) / np + 255u) & ~255u : cap_e;
                if (piece > cap_e) piece = cap_e;
                for (uint32_t pos = 0; pos < nj; pos += piece) {
                    const uint32_t len = std::min(piece, nj - pos), d0 = base + pos;
                    const FrIn& A = s.in;
                    BatchView B{len, 0, packed ? A.d_keys : b->key_bytes, A.d_key_off + d0, A.d_hits + d0, A.d_limit + d0, A.d_duration + d0,
                                b->burst ? A.d_burst + d0 : nullptr, b->created_at ? A.d_created_at + d0 : nullptr,
                                A.d_algorithm + d0, A.d_behavior + d0, b->is_owner ? A.d_is_owner + d0 : nullptr, nullptr, nullptr, b->now_ms,
                                0, packed ? nullptr : A.d_key_len + d0};
                    fifo[j].push_back(GroupItem{B, ResultView{s.o_status + d0, s.o_limit + d0, s.o_remaining + d0, s.o_reset + d0, s.o_err + d0}});
                }
                base += nj;
            }
            if (total != s.n) { rc = fail(GUBER_E_HIP, "guber_front: the shares do not add up to the generation"); break; }
            // a generation of small shares on ONE stream (the payload stage's): all its tables in one pair of launches
            bool one_pair = f->streams.size() == 1 && ne > 1 && s.h_margs.p && s.n <= f->one_pair_max;
            guber_engine* og[MULTI_MEM_MAX]; GroupItem oi[MULTI_MEM_MAX]; int on = 0;
            for (uint32_t j = 0; j < ne && one_pair; ++j) {
                if (fifo[j].empty()) continue;
                one_pair = fifo[j].size() == 1 && can_fuse(f->eng[j], fifo[j][0].B.n);
                og[on] = f->eng[j]; oi[on] = fifo[j][0]; ++on;
            }
            if (one_pair && on > 1) {
                FpSpan sp(&fp[2]);
                rc = pendset.flush_all();                                   // (what an earlier generation of this call holds back on these tables goes first)
                int r1 = rc ? rc : launch_group_mem(og, oi, on, &enq, (MultiArgsMem*)s.h_margs.p, (MultiArgsMem*)s.d_margs.p);
                if (r1 == 1) r1 = dispatch_rounds(f->eng.data(), ne, fifo, ps, &enq, hook_of.data());   // (a table turned out tight: the ordinary way)
                rc = r1;
            } else { FpSpan sp(&fp[2]); rc = dispatch_rounds(f->eng.data(), ne, fifo, ps, &enq, hook_of.data()); }
        }
        s.dispatched = true;
        for (auto& h : s.hooks) h->launched();                      // the dispatcher's hold: the event is recorded once nothing of the generation is held back
        if (rc) break;
        { FpSpan sp(&fp[3]); rc = drain_outs(k + 1 > f->out_delay ? k + 1 - f->out_delay : 0, false); }
    }
    // (what was routed or enqueued is completed, also after an error: its evaluations go now, its answers go home)
    {
        for (uint32_t k = 0; k < next_route; ++k) {
            guber_front::Slot& s = slot_of(k);
            if (k >= next_out && !s.dispatched) { s.dispatched = true; s.n = 0; for (auto& h : s.hooks) h->launched(); }   // (routed, never evaluated: an error above)
        }
        const int rcf = pendset.flush_all();
        if (!rc) rc = rcf;
        const int rco = drain_outs(next_route, true);
        if (!rc) rc = rco;
    }
    f->generations += next_out;
    if (after && !rc && hipEventRecord(after, f->last_os ? f->last_os : f->rs) != hipSuccess) rc = fail(GUBER_E_HIP, "hipEventRecord");
    if (g_dprof && next_out) {
        fprintf(stderr, "[front] %u generations; per generation: routing enqueue %.2f us, waiting for the shares' sizes %.2f, dispatch %.2f, answers %.2f\n", next_out,
                fp[0] / 1e3 / next_out, fp[1] / 1e3 / next_out, fp[2] / 1e3 / next_out, fp[3] / 1e3 / next_out);
        if (tl_dp[6]) fprintf(stderr, "[front dispatch] %llu batches in %llu groups; per batch: wait-for-progress %.2f us, locks %.2f, preludes+plans %.2f, argument blocks %.2f, launches %.2f\n",
                (unsigned long long)tl_dp[6], (unsigned long long)tl_dp[5], tl_dp[0] / 1e3 / tl_dp[6], tl_dp[1] / 1e3 / tl_dp[6], tl_dp[2] / 1e3 / tl_dp[6],
                tl_dp[3] / 1e3 / tl_dp[6], tl_dp[4] / 1e3 / tl_dp[6]);
        for (auto& v : tl_dp) v = 0;
    }
    return rc;
}

// The routing of the NEXT call's first generation, enqueued now (the payload stage: its flusher never waits for the GPU, so it routes as
// soon as a stage is decoded and comes back for the evaluation when the shares' sizes are in host memory).  The next front_eval call must
// hand over the same generation first.
static int front_route_ahead(guber_front* f, const FrontGen* g) {
    if (g->b.n > f->cap) return fail(GUBER_E_BATCH_TOO_LARGE, "generation larger than the front was created for");
    if (g->key_stride & 7u) return fail(GUBER_E_INVALID_ARG, "key rows are a multiple of 8 bytes apart");
    std::lock_guard<std::mutex> lk(f->mu);
    if (f->pre_routed) return fail(GUBER_E_INVALID_ARG, "guber_front: a generation is routed ahead already");
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    const int rc = front_route(f, f->slots[f->generations % f->depth], g, (int64_t)f->generations);
    if (!rc) f->pre_routed = true;
    return rc;
}
// have the shares' sizes of the generation routed ahead reached host memory?
static bool front_routed_ahead_ready(guber_front* f) {
    std::lock_guard<std::mutex> lk(f->mu);
    if (!f->pre_routed) return true;
    guber_front::Slot& s = f->slots[f->generations % f->depth];
    if (s.n == 0) return true;
    for (int q = 0; q <= MULTI_MEM_MAX; ++q)
        if ((uint32_t)(__atomic_load_n((volatile unsigned long long*)&s.host.p->w[q], __ATOMIC_ACQUIRE) >> 32) != s.seq) return false;
    return true;
}

extern "C" int guber_front_synchronize(guber_front_t* f) {
    if (!f) return fail(GUBER_E_INVALID_ARG, "null front");
    std::lock_guard<std::mutex> lk(f->mu);
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    for (auto st : f->streams) HIPCHK(hipStreamSynchronize(st));
    for (hipStream_t st : {f->rs, f->rs2, f->os}) HIPCHK(hipStreamSynchronize(st));
    return GUBER_OK;
}

// microseconds per generation since the last call (needs guber_profile_enable on the front's FIRST engine while the generations ran);
// waits for the routing stream
extern "C" int guber_front_latencies(guber_front_t* f, float* us, uint32_t cap, uint32_t* n_out) {
    if (!f || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(f->mu);
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamSynchronize(f->os));
    uint32_t n = 0;
    guber_engine* e0 = f->eng[0];
    std::lock_guard<std::mutex> lk2(e0->mu);
    for (auto& g : f->gen_spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, g.a, g.b) == hipSuccess && us && n < cap) us[n] = ms * 1e3f;
        ++n;
        e0->event_pool.push_back(g.a); e0->event_pool.push_back(g.b);
    }
    f->gen_spans.clear();
    *n_out = n;
    return GUBER_OK;
}

extern "C" void* guber_front_stream(guber_front_t* f) { return f ? (void*)f->os : nullptr; }

extern "C" int guber_front_stats(guber_front_t* f, guber_front_stats_t* out) {
    if (!f || !out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(f->mu);
    out->generations = f->generations; out->forced_flushes = f->forced_flushes; out->host_waits = f->host_waits;
    out->host_wait_us = (uint64_t)(f->host_wait_ms * 1e3);
    return GUBER_OK;
}
