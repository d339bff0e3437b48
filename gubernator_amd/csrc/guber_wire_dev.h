// guber_wire_dev.h — host side of the device wire decoder (include/guber_wire.h guber_wire_dev_*; kernels: guber_kernels_wire.h).
// Included by guber_engine.hip (it evaluates the decoded batch through the engine's own launch path).
#pragma once
#include "../../include/guber_wire.h"
#include "guber_kernels_wire.h"

struct guber_wire_dev {
    guber_engine* e = nullptr;
    uint32_t max_items = 0, max_bytes = 0, max_rpcs = 0, cap_per_rpc = 0, stride = 0;
    // staging, host and device alike: the decode's arguments off[R] | len[R] | wfirst[R + 1] | owner bytes[R] (meta_bytes, a multiple of 64), then the
    // payload bytes — ONE copy command brings both to HBM
    PinBuf<uint8_t> h_buf; size_t meta_bytes = 0;
    uint32_t* h_meta() const { return (uint32_t*)h_buf.p; }
    uint8_t* h_pay() const { return h_buf.p + meta_bytes; }
    CohBuf<uint32_t> h_rep;                                                              // the verdicts, written by k_wire_kill: first[R + 1] | count[R] | status[R]
    DevBuf<uint8_t> d_buf, d_rows, d_u8; DevBuf<uint32_t> d_u32, d_rec; DevBuf<int32_t> d_status, d_algo; DevBuf<int64_t> d_i64;
    DevBuf<int64_t> d_out64; DevBuf<uint8_t> d_out8;
    DevBuf<uint2> d_went; uint32_t max_windows = 0;                                        // k_wire_win_a -> k_wire_win_b (8 KB per window of 8 KB)
    PinBuf<uint8_t> h_cols;                                                              // read-back of the decoded columns (tests, response encoding)
    uint32_t nrpc = 0, n_items = 0; int64_t now_ms = 0;
    guber::WireIn in{}; guber::WireScratch sc{}; guber::WireOut out{};
    // the asynchronous calls (guber_wire_dev_*_async / _poll: the payload stage of guber_wire_pool.h): a stream of the caller's for the
    // decode, one event for "the decode's verdicts are in host memory", one for "the answers are where the caller wanted them"
    hipStream_t own = nullptr; hipEvent_t ev_dec = nullptr, ev_eval = nullptr; bool dec_pending = false, eval_pending = false, routed = false;
    hipStream_t stream() const { return own ? own : e->stream; }
};

extern "C" int guber_wire_dev_create(guber_engine_t* e, uint32_t max_items, uint32_t max_payload_bytes, uint32_t max_rpcs, guber_wire_dev_t** outp) {
    if (!e || !outp || !max_items || !max_payload_bytes || !max_rpcs) return fail(GUBER_E_INVALID_ARG, "bad argument");
    *outp = nullptr;
    // (a decoder for a front may hold more than its engine's pipelines take in one batch: guber_wire_dev_eval refuses such a batch,
    //  guber_wire_dev_eval_front routes it to the front's engines in pieces)
    if (max_items > FR_MAX_N) return fail(GUBER_E_BATCH_TOO_LARGE, "a decoder holds at most 4 194 304 items");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    guber_wire_dev* d = new guber_wire_dev();
    d->e = e; d->max_items = max_items; d->max_rpcs = max_rpcs;
    d->max_bytes = max_payload_bytes + 16 * max_rpcs + 64;                                // every payload starts 16-byte aligned
    d->cap_per_rpc = std::min<uint32_t>(max_items, 4096);
    d->stride = ((e->max_key + 7u) & ~7u) + 8u;
    const size_t M = max_items, R = max_rpcs;
    d->max_windows = d->max_bytes / guber::WP_WIN + max_rpcs;                            // (a payload of len bytes: len / 8 KB + 1 windows)
    d->meta_bytes = (((3 * R + 1) * 4 + R) + 63) & ~(size_t)63;                          // the arguments' block in front of the payload bytes
    int rc = d->h_buf.ensure(d->meta_bytes + d->max_bytes) | d->h_rep.ensure(3 * R + 8) | d->d_buf.ensure(d->meta_bytes + d->max_bytes) |
             d->d_u32.ensure(R + (R + 1) + 1 + M + M + M) | d->d_rec.ensure(2 * R * d->cap_per_rpc) | d->d_status.ensure(R) | d->d_algo.ensure(M) |
             d->d_i64.ensure(5 * M) | d->d_rows.ensure(M * d->stride + 64) | d->d_u8.ensure(3 * M) | d->d_out64.ensure(3 * M) | d->d_out8.ensure(2 * M) |
             d->d_went.ensure((size_t)d->max_windows * guber::WP_ENT);
    if (rc) { guber_wire_dev_destroy(d); return GUBER_E_NOMEM; }
    uint32_t* a = (uint32_t*)d->d_buf.p;                                                 // (the block the host's arguments are copied over, with the payload, in one piece)
    d->in.rpc_off = a; a += R; d->in.rpc_len = a; a += R; d->sc.wfirst = a; a += R + 1; d->in.rpc_owner = (const uint8_t*)a;
    d->in.buf = d->d_buf.p + d->meta_bytes;
    uint32_t* u = d->d_u32.p;
    d->sc.count = u; u += R; d->sc.first = u; u += R + 1;
    d->sc.done = u; u += 1; d->sc.went = d->d_went.p;
    if (hipMemset(d->sc.done, 0, 4) != hipSuccess) { guber_wire_dev_destroy(d); return fail(GUBER_E_HIP, "hipMemset"); }
    d->out.key_len = u; u += M; d->out.behavior = u; u += M; d->out.item_rpc = u; u += M;
    d->in.cap_per_rpc = d->cap_per_rpc; d->in.cap_items = max_items;
    d->out.rep_first = d->h_rep.p; d->out.rep_count = d->h_rep.p + R + 1; d->out.rep_status = (int32_t*)(d->h_rep.p + 2 * R + 1);
    d->sc.rec_off = d->d_rec.p; d->sc.rec_len = d->d_rec.p + (size_t)R * d->cap_per_rpc; d->sc.status = d->d_status.p;
    d->out.key_rows = d->d_rows.p; d->out.key_stride = d->stride;
    int64_t* q = d->d_i64.p;
    d->out.hits = q; d->out.limit = q + M; d->out.duration = q + 2 * M; d->out.burst = q + 3 * M; d->out.created_at = q + 4 * M;
    d->out.algo_raw = d->d_algo.p;
    d->out.algorithm = d->d_u8.p; d->out.is_owner = d->d_u8.p + M; d->out.pre_err = d->d_u8.p + 2 * M;
    *outp = d;
    return GUBER_OK;
}

extern "C" void guber_wire_dev_destroy(guber_wire_dev_t* d) {
    if (!d) return;
    if (d->e) { (void)hipSetDevice(d->e->device); (void)hipStreamSynchronize(d->stream()); }
    if (d->ev_dec) (void)hipEventDestroy(d->ev_dec);
    if (d->ev_eval) (void)hipEventDestroy(d->ev_eval);
    d->h_buf.release(); d->h_rep.release(); d->d_buf.release(); d->d_rows.release(); d->d_u8.release();
    d->d_u32.release(); d->d_went.release(); d->d_rec.release(); d->d_status.release(); d->d_algo.release(); d->d_i64.release(); d->d_out64.release(); d->d_out8.release();
    d->h_cols.release();
    delete d;
}

// The payloads lie in the decoder's pinned buffer (h_buf) at h_off[] (16-byte aligned): copy [lo, hi) to the device and decode.
// Engine mutex held, device set.
static int wire_dev_decode_enqueue(guber_wire_dev* d, uint32_t nrpc, size_t lo, size_t hi, uint32_t windows, bool multi, const uint8_t* is_owner,
                                   uint32_t max_per_rpc, int64_t now_ms) {
    const size_t R = d->max_rpcs;
    uint32_t* h_off = d->h_meta();
    uint8_t* h_owner = (uint8_t*)(h_off + 3 * R + 1);
    hipStream_t st = d->stream();
    if (is_owner) memcpy(h_owner, is_owner, nrpc); else memset(h_owner, 1, nrpc);
    // ONE copy — the decode's arguments (off | len | wfirst | owner) and, behind them, the payload bytes up to the last payload's end — then the
    // kernels; the verdicts come back through k_wire_kill's stores into host memory: no copy, no memset behind or between (every command costs
    // the stream ~5 us; what lies between the arguments and the first payload of a caller that staged at an offset travels along unread)
    (void)lo;
    HIPCHK(hipMemcpyAsync(d->d_buf.p, d->h_buf.p, d->meta_bytes + hi, hipMemcpyHostToDevice, st));
    d->in.nrpc = nrpc; d->in.max_per_rpc = max_per_rpc; d->out.now_ms = now_ms;
    // the chain of every payload: in parallel (a workgroup per 8 KB window, pointer doubling: k_wire_win_a — only when a payload has
    // more than one window — says where the chain enters each window, k_wire_win_b finds the records), then the serial walk for the
    // payloads that hold anything but plain records, and the numbering of the batch (by that launch's last workgroup; a launch of its
    // own for batches of many payloads)
    // (payloads of less than 1 KB have no windows: a wave walks three dozen records sooner than a workgroup sets up; GUBER_WIRE_SERIAL=1: the
    // serial walk for all, for A/B runs)
    static const bool wire_serial = [] { const char* v = guber_lab_env("GUBER_WIRE_SERIAL"); return v && atoi(v) != 0; }();
    if (!wire_serial) {
        if (multi) hipLaunchKernelGGL(guber::k_wire_win_a, dim3(windows), dim3(guber::WP_T), 0, st, d->in, d->sc);
        if (windows) hipLaunchKernelGGL(guber::k_wire_win_b, dim3(windows), dim3(guber::WP_T), 0, st, d->in, d->sc);
    }
    const bool fused_numbering = nrpc <= guber::WIRE_NUMBER_FUSED;
    hipLaunchKernelGGL(guber::k_wire_scan, dim3(nrpc), dim3(64), 0, st, d->in, d->sc, wire_serial ? 0u : 1u, fused_numbering ? 1u : 0u);
    if (!fused_numbering) hipLaunchKernelGGL(guber::k_wire_prefix, dim3(1), dim3(1024), 0, st, d->in, d->sc);
    const unsigned blocks = (d->max_items + 255) / 256;
    hipLaunchKernelGGL(guber::k_wire_fill, dim3(blocks), dim3(256), 0, st, d->in, d->sc, d->out);
    hipLaunchKernelGGL(guber::k_wire_kill, dim3(blocks), dim3(256), 0, st, d->in, d->sc, d->out);
    HIPCHK(hipGetLastError());
    return GUBER_OK;
}
// ... and its verdicts, once the stream has got there
static int wire_dev_decode_finish(guber_wire_dev* d, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    const uint32_t nrpc = d->nrpc;
    const volatile uint32_t* h_first = d->out.rep_first; const volatile uint32_t* h_count = d->out.rep_count; const volatile int32_t* h_status = d->out.rep_status;
    d->n_items = h_first[nrpc];
    *n_items = d->n_items;
    for (uint32_t r = 0; r < nrpc; ++r) {
        const int32_t st = h_status[r];
        if (status) status[r] = st;
        if (first) first[r] = h_first[r];
        if (count) count[r] = st == GUBER_OK ? h_count[r] : (st == GUBER_E_WIRE_TOO_LARGE ? h_count[r] : 0);
    }
    return GUBER_OK;
}
static int wire_dev_decode_staged_locked(guber_wire_dev* d, uint32_t nrpc, size_t lo, size_t hi, uint32_t windows, bool multi, const uint8_t* is_owner,
                                         uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    const int rc = wire_dev_decode_enqueue(d, nrpc, lo, hi, windows, multi, is_owner, max_per_rpc, now_ms);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(d->stream()));
    return wire_dev_decode_finish(d, status, first, count, n_items);
}

// Decode nrpc serialized GetRateLimitsReq / GetPeerRateLimitsReq payloads into ONE device batch.  The payload bytes are copied into
// the decoder's pinned buffer (this copy and the H2D transfer are what the C call's rate is made of: DESIGN.md 5b; a receive path
// that reads its sockets straight into guber_wire_dev_buffer() and calls guber_wire_dev_decode_staged skips the copy).
extern "C" int guber_wire_dev_decode(guber_wire_dev_t* d, const uint8_t* const* msgs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                     uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    if (!d || (nrpc && (!msgs || !lens)) || !n_items) return fail(GUBER_E_INVALID_ARG, "null argument");
    *n_items = 0;
    if (nrpc > d->max_rpcs) return fail(GUBER_E_BATCH_TOO_LARGE, "more RPCs than the decoder was created for");
    guber_engine* e = d->e;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    d->nrpc = nrpc; d->n_items = 0; d->now_ms = now_ms;
    if (!nrpc) return GUBER_OK;
    uint32_t* h_off = d->h_meta(); uint32_t* h_len = h_off + d->max_rpcs; uint32_t* h_wfirst = h_off + 2 * (size_t)d->max_rpcs;
    size_t pos = 0;
    uint32_t windows = 0;
    bool multi = false;                                                                    // a payload of more than one window: k_wire_win_a has something to say
    for (uint32_t r = 0; r < nrpc; ++r) {
        const uint32_t nw = guber::wire_windows_of(lens[r]);
        h_wfirst[r] = windows; windows += nw; multi = multi || nw > 1;
        pos = (pos + 15) & ~(size_t)15;
        if (pos + lens[r] + 16 > d->max_bytes) return fail(GUBER_E_WIRE_FULL, "payload bytes exceed the decoder's buffer");
        if (lens[r] && !msgs[r]) return fail(GUBER_E_INVALID_ARG, "null payload");
        memcpy(d->h_pay() + pos, msgs[r], lens[r]);
        h_off[r] = (uint32_t)pos; h_len[r] = lens[r];
        pos += lens[r];
    }
    h_wfirst[nrpc] = windows;                                                              // (<= max_windows: the bytes fit)
    memset(d->h_pay() + pos, 0, 16);
    return wire_dev_decode_staged_locked(d, nrpc, 0, pos + 16, windows, multi, is_owner, max_per_rpc, now_ms, status, first, count, n_items);
}

// The decoder's pinned staging buffer, and the decode of payloads that already lie in it: offs[r] 16-byte aligned, ascending, the
// payloads not overlapping, 16 bytes of room behind the last one (the kernels read whole 16-byte pieces: what lies there is never
// interpreted).  Everything else as guber_wire_dev_decode.
extern "C" int guber_wire_dev_buffer(guber_wire_dev_t* d, uint8_t** buf, size_t* cap) {
    if (!d || !buf || !cap) return fail(GUBER_E_INVALID_ARG, "null argument");
    *buf = d->h_pay(); *cap = d->max_bytes;
    return GUBER_OK;
}
static int wire_dev_decode_staged_any(guber_wire_dev_t* d, const uint32_t* offs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                      uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items, bool async) {
    if (!d || (nrpc && (!offs || !lens)) || (!async && !n_items)) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n_items) *n_items = 0;
    if (nrpc > d->max_rpcs) return fail(GUBER_E_BATCH_TOO_LARGE, "more RPCs than the decoder was created for");
    if (d->dec_pending || d->eval_pending) return fail(GUBER_E_INVALID_ARG, "the decoder's previous asynchronous call has not been collected");
    guber_engine* e = d->e;
    // (a decoder with a stream of its own touches nothing of its engine's: it does not take the engine's lock — the payload stage's intake
    //  thread enqueues decodes while its front thread holds the engines' locks to launch their groups)
    std::unique_lock<std::mutex> lk(e->mu, std::defer_lock);
    if (!d->own) lk.lock();
    if (d->own ? hipSetDevice(e->device) != hipSuccess : e->set_device() != 0) return fail(GUBER_E_HIP, "hipSetDevice");
    d->nrpc = nrpc; d->n_items = 0; d->now_ms = now_ms;
    if (!nrpc) return GUBER_OK;
    uint32_t* h_off = d->h_meta(); uint32_t* h_len = h_off + d->max_rpcs; uint32_t* h_wfirst = h_off + 2 * (size_t)d->max_rpcs;
    uint32_t windows = 0;
    bool multi = false;
    size_t end = 0;
    for (uint32_t r = 0; r < nrpc; ++r) {
        if ((offs[r] & 15u) || (size_t)offs[r] < end || (size_t)offs[r] + lens[r] + 16 > d->max_bytes)
            return fail(GUBER_E_INVALID_ARG, "staged payloads: offsets 16-byte aligned, ascending, not overlapping, 16 bytes of room behind the last");
        const uint32_t nw = guber::wire_windows_of(lens[r]);
        h_wfirst[r] = windows; windows += nw; multi = multi || nw > 1;
        h_off[r] = offs[r]; h_len[r] = lens[r];
        end = (size_t)offs[r] + lens[r];
    }
    h_wfirst[nrpc] = windows;
    if (windows > d->max_windows) return fail(GUBER_E_WIRE_FULL, "staged payloads: more 8 KB windows than the decoder was created for");
    if (!async) return wire_dev_decode_staged_locked(d, nrpc, offs[0], end + 16, windows, multi, is_owner, max_per_rpc, now_ms, status, first, count, n_items);
    const int rc = wire_dev_decode_enqueue(d, nrpc, offs[0], end + 16, windows, multi, is_owner, max_per_rpc, now_ms);
    if (rc) return rc;
    if (!d->ev_dec) HIPCHK(hipEventCreateWithFlags(&d->ev_dec, hipEventDisableTiming));
    HIPCHK(hipEventRecord(d->ev_dec, d->stream()));
    d->dec_pending = true;
    return GUBER_OK;
}
extern "C" int guber_wire_dev_decode_staged(guber_wire_dev_t* d, const uint32_t* offs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                            uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    return wire_dev_decode_staged_any(d, offs, lens, nrpc, is_owner, max_per_rpc, now_ms, status, first, count, n_items, false);
}

// ---- the same in two halves, for a caller that keeps several decoders busy (the payload stage: guber_wire_pool.h) ----
// guber_wire_dev_set_stream       the decode's copies and kernels go to `stream` (hipStream_t) instead of the engine's stream; the decoder's
//                                 previous work is waited for.  NULL = the engine's stream again.
// guber_wire_dev_decode_staged_async   as guber_wire_dev_decode_staged up to the last enqueue: returns at once (is_owner must stay valid
//                                 until the decode has been collected; pinned memory keeps the copy asynchronous)
// guber_wire_dev_decode_collect   wait = 0: GUBER_PENDING (1) while the GPU is still at it; otherwise (or wait = 1: waits) the verdicts
extern "C" int guber_wire_dev_set_stream(guber_wire_dev_t* d, void* stream) {
    if (!d) return fail(GUBER_E_INVALID_ARG, "null argument");
    guber_engine* e = d->e;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamSynchronize(d->stream()));
    d->own = (hipStream_t)stream;
    return GUBER_OK;
}
extern "C" int guber_wire_dev_decode_staged_async(guber_wire_dev_t* d, const uint32_t* offs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                                  uint32_t max_per_rpc, int64_t now_ms) {
    return wire_dev_decode_staged_any(d, offs, lens, nrpc, is_owner, max_per_rpc, now_ms, nullptr, nullptr, nullptr, nullptr, true);
}
extern "C" int guber_wire_dev_decode_collect(guber_wire_dev_t* d, int wait, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    if (!d || !n_items) return fail(GUBER_E_INVALID_ARG, "null argument");
    *n_items = 0;
    if (!d->dec_pending) { if (d->nrpc == 0) return GUBER_OK; return fail(GUBER_E_INVALID_ARG, "no asynchronous decode to collect"); }
    if (wait) HIPCHK(hipEventSynchronize(d->ev_dec));
    else {
        const hipError_t q = hipEventQuery(d->ev_dec);
        if (q == hipErrorNotReady) return GUBER_PENDING;
        if (q != hipSuccess) { d->dec_pending = false; return fail(GUBER_E_HIP, "hipEventQuery", q); }
    }
    d->dec_pending = false;
    return wire_dev_decode_finish(d, status, first, count, n_items);
}

// The decoded batch through the engine's pipelines; results to host arrays of n_items entries.
extern "C" int guber_wire_dev_eval(guber_wire_dev_t* d, guber_result_t* r) {
    if (!d || !r) return fail(GUBER_E_INVALID_ARG, "null argument");
    guber_engine* e = d->e;
    const uint32_t n = d->n_items;
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
    if (!n) return GUBER_OK;
    if (!r->status || !r->limit || !r->remaining || !r->reset_time || !r->err) return fail(GUBER_E_INVALID_ARG, "result is missing an array");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    BatchView B{n, 0, d->out.key_rows, nullptr, d->out.hits, d->out.limit, d->out.duration, d->out.burst, d->out.created_at,
                d->out.algorithm, d->out.behavior, d->out.is_owner, nullptr, nullptr, d->now_ms};
    B.key_stride = d->stride; B.key_len = d->out.key_len;
    const size_t M = d->max_items;
    ResultView R{d->d_out8.p, d->d_out64.p, d->d_out64.p + M, d->d_out64.p + 2 * M, d->d_out8.p + M};
    const DevCounters before = e->last_ctr;
    int rc = launch_batch(e, B, R);
    if (rc) return rc;
    hipStream_t st = e->stream;
    HIPCHK(hipMemcpyAsync(r->status, d->d_out8.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->err, d->d_out8.p + M, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->limit, d->d_out64.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->remaining, d->d_out64.p + M, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->reset_time, d->d_out64.p + 2 * M, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    rc = enqueue_counter_readback(e);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));
    fold_counters(e);
    r->over_limit_count = e->last_ctr.over - before.over; r->cache_hits = e->last_ctr.hits - before.hits;
    r->cache_misses = e->last_ctr.misses - before.misses; r->cache_size = e->last_ctr.size;
    return GUBER_OK;
}

// The decoded batch through a FRONT (guber_front.h) instead of the decoder's own engine: the items — in the order of their RPCs, i.e. in
// arrival order — are routed to the front's engines on the device, evaluated there and answered in arrival order; results to host arrays
// of n_items entries.  Decode (k_wire_*), routing (k_fr_*), evaluation and the answers' order all happen in HBM: what the pool's callers
// do per request on the host today (hash, placement, copy: DESIGN.md 5b) has a device counterpart for every step.  The decoder's engine
// only lends its stream to the decode; it need not be one of the front's.
static int wire_dev_front_gen(guber_wire_dev* d, FrontGen& g) {
    memset(&g.b, 0, sizeof g.b);
    g.b.n = d->n_items; g.b.key_bytes = d->out.key_rows; g.b.hits = d->out.hits; g.b.limit = d->out.limit; g.b.duration = d->out.duration; g.b.burst = d->out.burst;
    g.b.created_at = d->out.created_at; g.b.algorithm = d->out.algorithm; g.b.behavior = d->out.behavior; g.b.is_owner = d->out.is_owner; g.b.now_ms = d->now_ms;
    g.key_stride = d->stride; g.key_len = d->out.key_len;
    return 0;
}
extern "C" int guber_wire_dev_eval_front(guber_wire_dev_t* d, guber_front_t* f, guber_result_t* r) {
    if (!d || !f || !r) return fail(GUBER_E_INVALID_ARG, "null argument");
    const uint32_t n = d->n_items;
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
    if (!n) return GUBER_OK;
    if (!r->status || !r->limit || !r->remaining || !r->reset_time || !r->err) return fail(GUBER_E_INVALID_ARG, "result is missing an array");
    if (n > f->cap) return fail(GUBER_E_BATCH_TOO_LARGE, "more items than the front's generations hold");
    const size_t M = d->max_items;
    FrontGen g;
    wire_dev_front_gen(d, g);
    guber_result_t dr{};
    dr.status = d->d_out8.p; dr.err = d->d_out8.p + M; dr.limit = d->d_out64.p; dr.remaining = d->d_out64.p + M; dr.reset_time = d->d_out64.p + 2 * M;
    {   // (the decode was synchronised on its stream: guber_wire_dev_decode* read its verdicts back)
        const int rc = front_eval(f, &g, &dr, 1, nullptr);
        if (rc) return rc;
    }
    {
        const int rc = guber_front_synchronize(f);
        if (rc) return rc;
    }
    guber_engine* e = d->e;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    hipStream_t st = d->stream();
    HIPCHK(hipMemcpyAsync(r->status, d->d_out8.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->err, d->d_out8.p + M, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->limit, d->d_out64.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->remaining, d->d_out64.p + M, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->reset_time, d->d_out64.p + 2 * M, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return GUBER_OK;
}
// guber_wire_dev_eval_front_async   the same, enqueued only: `r`'s arrays are DEVICE-VISIBLE (HBM, or host memory the device writes in
//                                   place over PCIe: hipHostMalloc / guber_alloc_pinned) — the answers' last hop (k_fr_out) writes them where
//                                   they are wanted, no copy follows
// guber_wire_dev_eval_collect       wait = 0: GUBER_PENDING while the answers are on their way; GUBER_OK: they are there
// guber_wire_dev_route_front_async  only the front's routing of the decoded batch (k_fr_count / k_fr_scan / k_fr_scatter), enqueued; the
//                                   guber_wire_dev_eval_front_async that follows — for the same decoder, before anything else goes through the
//                                   front — finds the shares' sizes in host memory instead of waiting for them
// guber_wire_dev_route_ready        GUBER_PENDING until they are
extern "C" int guber_wire_dev_route_front_async(guber_wire_dev_t* d, guber_front_t* f) {
    if (!d || !f) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (d->dec_pending || d->eval_pending || d->routed) return fail(GUBER_E_INVALID_ARG, "the decoder's previous asynchronous call has not been collected");
    if (!d->n_items) return GUBER_OK;
    if (d->n_items > f->cap) return fail(GUBER_E_BATCH_TOO_LARGE, "more items than the front's generations hold");
    FrontGen g;
    wire_dev_front_gen(d, g);
    const int rc = front_route_ahead(f, &g);
    if (!rc) d->routed = true;
    return rc;
}
extern "C" int guber_wire_dev_route_ready(guber_wire_dev_t* d, guber_front_t* f) {
    if (!d || !f) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (!d->routed) return GUBER_OK;
    return front_routed_ahead_ready(f) ? GUBER_OK : GUBER_PENDING;
}
extern "C" int guber_wire_dev_eval_front_async(guber_wire_dev_t* d, guber_front_t* f, guber_result_t* r) {
    if (!d || !f || !r) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (d->dec_pending || d->eval_pending) return fail(GUBER_E_INVALID_ARG, "the decoder's previous asynchronous call has not been collected");
    const uint32_t n = d->n_items;
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
    if (!n) return GUBER_OK;
    if (!r->status || !r->limit || !r->remaining || !r->reset_time || !r->err) return fail(GUBER_E_INVALID_ARG, "result is missing an array");
    if (n > f->cap) return fail(GUBER_E_BATCH_TOO_LARGE, "more items than the front's generations hold");
    if (!d->ev_eval) {
        if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
        HIPCHK(hipEventCreateWithFlags(&d->ev_eval, hipEventDisableTiming));
    }
    FrontGen g;
    wire_dev_front_gen(d, g);
    d->routed = false;
    const int rc = front_eval(f, &g, r, 1, nullptr, d->ev_eval);
    if (rc) return rc;
    d->eval_pending = true;
    return GUBER_OK;
}
// wire_dev_eval_front_enc_async     the payload stage's form of guber_wire_dev_eval_front_async: k_wire_enc IS the answers' last hop (front_out
//                                   launches it in k_fr_out's place) — it reads the shares' answers through the routing's fwd[] and leaves every
//                                   RPC's GetRateLimitsResp bytes in `enc` (DEVICE-VISIBLE HOST memory of wire_enc_bytes(max_items, max_rpcs)
//                                   bytes; RPC r's at wire_enc_off(first[r], r), enc_len[r] of them).  An RPC with an item error is not encoded:
//                                   enc_len[r] = WIRE_ENC_RAW and its raw answers are in `raw`'s (device-visible host) arrays, in arrival order,
//                                   for the host transcoder.
static int wire_dev_eval_front_enc_async(guber_wire_dev* d, guber_front* f, uint8_t* enc, uint32_t* enc_len, guber_result_t* raw) {
    if (!d || !f || !enc || !enc_len || !raw) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (d->dec_pending || d->eval_pending) return fail(GUBER_E_INVALID_ARG, "the decoder's previous asynchronous call has not been collected");
    const uint32_t n = d->n_items;
    raw->over_limit_count = raw->cache_hits = raw->cache_misses = raw->unexpired_evictions = 0; raw->cache_size = 0;
    if (!n) return GUBER_OK;
    if (!raw->status || !raw->limit || !raw->remaining || !raw->reset_time || !raw->err) return fail(GUBER_E_INVALID_ARG, "result is missing an array");
    if (n > f->cap) return fail(GUBER_E_BATCH_TOO_LARGE, "more items than the front's generations hold");
    if (hipSetDevice(f->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    if (!d->ev_eval) HIPCHK(hipEventCreateWithFlags(&d->ev_eval, hipEventDisableTiming));
    const size_t M = d->max_items;
    FrontGen g;
    wire_dev_front_gen(d, g);
    guber_result_t dr{};
    dr.status = d->d_out8.p; dr.err = d->d_out8.p + M; dr.limit = d->d_out64.p; dr.remaining = d->d_out64.p + M; dr.reset_time = d->d_out64.p + 2 * M;
    d->routed = false;
    // (the encoder takes the place of the answers' last hop, front_out: it reads the shares' answers through the routing's fwd[] — one launch
    //  and one pass over the answers less than k_fr_out into HBM + an encoder behind it, which measured -10 % with 64 - 128 callers)
    guber::WireEnc E{};
    E.nrpc = d->nrpc; E.first = d->sc.first; E.count = d->sc.count; E.status = d->sc.status;
    E.enc = enc; E.enc_len = enc_len;
    E.h_status = raw->status; E.h_err = raw->err; E.h_limit = raw->limit; E.h_remaining = raw->remaining; E.h_reset = raw->reset_time;
    f->enc_hook = &E;
    const int rc = front_eval(f, &g, &dr, 1, nullptr, d->ev_eval);
    f->enc_hook = nullptr;
    if (rc) return rc;
    d->eval_pending = true;
    return GUBER_OK;
}
extern "C" int guber_wire_dev_eval_collect(guber_wire_dev_t* d, int wait) {
    if (!d) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (!d->eval_pending) return GUBER_OK;
    if (wait) HIPCHK(hipEventSynchronize(d->ev_eval));
    else {
        const hipError_t q = hipEventQuery(d->ev_eval);
        if (q == hipErrorNotReady) return GUBER_PENDING;
        if (q != hipSuccess) { d->eval_pending = false; return fail(GUBER_E_HIP, "hipEventQuery", q); }
    }
    d->eval_pending = false;
    return GUBER_OK;
}

// The decoded columns on the host (for response encoding and for tests): pointers into the decoder's own pinned memory, valid until
// the next decode.
extern "C" int guber_wire_dev_columns(guber_wire_dev_t* d, guber_wire_columns_t* c) {
    if (!d || !c) return fail(GUBER_E_INVALID_ARG, "null argument");
    guber_engine* e = d->e;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    memset(c, 0, sizeof *c);
    const size_t n = d->n_items, M = d->max_items;
    c->n = (uint32_t)n; c->key_stride = d->stride;
    if (!n) return GUBER_OK;
    const size_t bytes = n * d->stride + n * (5 * 8 + 4 + 4 + 4 + 3) + 256;
    if (d->h_cols.ensure(bytes)) return GUBER_E_NOMEM;
    uint8_t* p = d->h_cols.p;
    hipStream_t st = e->stream;
    auto take = [&](const void* src, size_t sz) { void* dst = p; (void)hipMemcpyAsync(dst, src, sz, hipMemcpyDeviceToHost, st); p += (sz + 15) & ~(size_t)15; return dst; };
    c->hits = (const int64_t*)take(d->out.hits, n * 8); c->limit = (const int64_t*)take(d->out.limit, n * 8); c->duration = (const int64_t*)take(d->out.duration, n * 8);
    c->burst = (const int64_t*)take(d->out.burst, n * 8); c->created_at = (const int64_t*)take(d->out.created_at, n * 8);
    c->key_len = (const uint32_t*)take(d->out.key_len, n * 4); c->behavior = (const uint32_t*)take(d->out.behavior, n * 4);
    c->algo_raw = (const int32_t*)take(d->out.algo_raw, n * 4);
    c->algorithm = (const uint8_t*)take(d->out.algorithm, n); c->is_owner = (const uint8_t*)take(d->out.is_owner, n); c->pre_err = (const uint8_t*)take(d->out.pre_err, n);
    c->key_rows = (const uint8_t*)take(d->out.key_rows, n * d->stride);
    (void)M;
    HIPCHK(hipStreamSynchronize(st));
    return GUBER_OK;
}
