// guber_wire_dev.h — host side of the device wire decoder (include/guber_wire.h guber_wire_dev_*; kernels: guber_kernels_wire.h).
// Included by guber_engine.hip (it evaluates the decoded batch through the engine's own launch path).
#pragma once
#include "../../include/guber_wire.h"
#include "guber_kernels_wire.h"

struct guber_wire_dev {
    guber_engine* e = nullptr;
    uint32_t max_items = 0, max_bytes = 0, max_rpcs = 0, cap_per_rpc = 0, stride = 0;
    PinBuf<uint8_t> h_buf; PinBuf<uint32_t> h_u32; PinBuf<int32_t> h_status;           // staging: payload bytes; off | len | first | count; status
    DevBuf<uint8_t> d_buf, d_owner, d_rows, d_u8; DevBuf<uint32_t> d_u32, d_rec; DevBuf<int32_t> d_status, d_algo; DevBuf<int64_t> d_i64;
    DevBuf<int64_t> d_out64; DevBuf<uint8_t> d_out8;
    DevBuf<uint2> d_went; uint32_t max_windows = 0;                                        // k_wire_win_a -> k_wire_win_b (8 KB per window of 8 KB)
    PinBuf<uint8_t> h_cols;                                                              // read-back of the decoded columns (tests, response encoding)
    uint32_t nrpc = 0, n_items = 0; int64_t now_ms = 0;
    guber::WireIn in{}; guber::WireScratch sc{}; guber::WireOut out{};
};

extern "C" int guber_wire_dev_create(guber_engine_t* e, uint32_t max_items, uint32_t max_payload_bytes, uint32_t max_rpcs, guber_wire_dev_t** outp) {
    if (!e || !outp || !max_items || !max_payload_bytes || !max_rpcs) return fail(GUBER_E_INVALID_ARG, "bad argument");
    *outp = nullptr;
    if (max_items > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "max_items above the engine's max_batch");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    guber_wire_dev* d = new guber_wire_dev();
    d->e = e; d->max_items = max_items; d->max_rpcs = max_rpcs;
    d->max_bytes = max_payload_bytes + 16 * max_rpcs + 64;                                // every payload starts 16-byte aligned
    d->cap_per_rpc = std::min<uint32_t>(max_items, 4096);
    d->stride = ((e->max_key + 7u) & ~7u) + 8u;
    const size_t M = max_items, R = max_rpcs;
    d->max_windows = d->max_bytes / guber::WP_WIN + max_rpcs;                            // (a payload of len bytes: len / 8 KB + 1 windows)
    int rc = d->h_buf.ensure(d->max_bytes) | d->h_u32.ensure(5 * R + 8) | d->h_status.ensure(R) | d->d_buf.ensure(d->max_bytes) | d->d_owner.ensure(R) |
             d->d_u32.ensure(2 * R + R + (R + 1) + (R + 1) + 1 + M + M + M) | d->d_rec.ensure(2 * R * d->cap_per_rpc) | d->d_status.ensure(R) | d->d_algo.ensure(M) |
             d->d_i64.ensure(5 * M) | d->d_rows.ensure(M * d->stride + 64) | d->d_u8.ensure(3 * M) | d->d_out64.ensure(3 * M) | d->d_out8.ensure(2 * M) |
             d->d_went.ensure((size_t)d->max_windows * guber::WP_ENT);
    if (rc) { guber_wire_dev_destroy(d); return GUBER_E_NOMEM; }
    uint32_t* u = d->d_u32.p;
    d->in.buf = d->d_buf.p; d->in.rpc_off = u; u += R; d->in.rpc_len = u; u += R;
    d->sc.count = u; u += R; d->sc.first = u; u += R + 1;
    d->sc.wfirst = u; u += R + 1; d->sc.done = u; u += 1; d->sc.went = d->d_went.p;
    if (hipMemset(d->sc.done, 0, 4) != hipSuccess) { guber_wire_dev_destroy(d); return fail(GUBER_E_HIP, "hipMemset"); }
    d->out.key_len = u; u += M; d->out.behavior = u; u += M; d->out.item_rpc = u; u += M;
    d->in.rpc_owner = d->d_owner.p; d->in.cap_per_rpc = d->cap_per_rpc; d->in.cap_items = max_items;
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
    if (d->e) { (void)hipSetDevice(d->e->device); (void)hipStreamSynchronize(d->e->stream); }
    d->h_buf.release(); d->h_u32.release(); d->h_status.release(); d->d_buf.release(); d->d_owner.release(); d->d_rows.release(); d->d_u8.release();
    d->d_u32.release(); d->d_went.release(); d->d_rec.release(); d->d_status.release(); d->d_algo.release(); d->d_i64.release(); d->d_out64.release(); d->d_out8.release();
    d->h_cols.release();
    delete d;
}

// The payloads lie in the decoder's pinned buffer (h_buf) at h_off[] (16-byte aligned): copy [lo, hi) to the device and decode.
// Engine mutex held, device set.
static int wire_dev_decode_staged_locked(guber_wire_dev* d, uint32_t nrpc, size_t lo, size_t hi, uint32_t windows, bool multi, const uint8_t* is_owner,
                                         uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    guber_engine* e = d->e;
    uint32_t* h_off = d->h_u32.p; uint32_t* h_len = h_off + d->max_rpcs; uint32_t* h_wfirst = h_off + 4 * (size_t)d->max_rpcs + 4;
    hipStream_t st = e->stream;
    HIPCHK(hipMemcpyAsync(d->d_buf.p + lo, d->h_buf.p + lo, hi - lo, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)d->in.rpc_off, h_off, (size_t)nrpc * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)d->in.rpc_len, h_len, (size_t)nrpc * 4, hipMemcpyHostToDevice, st));
    if (is_owner) HIPCHK(hipMemcpyAsync(d->d_owner.p, is_owner, nrpc, hipMemcpyHostToDevice, st));
    else HIPCHK(hipMemsetAsync(d->d_owner.p, 1, nrpc, st));
    d->in.nrpc = nrpc; d->in.max_per_rpc = max_per_rpc; d->out.now_ms = now_ms;
    // (the serial walk's last workgroup numbers the batch and resets this counter; a decode that failed or faulted part of the way would
    //  leave it raised and every later batch unnumbered, silently: four bytes a decode — ADVICE r05)
    HIPCHK(hipMemsetAsync(d->sc.done, 0, 4, st));
    // the chain of every payload: in parallel (a workgroup per 8 KB window, pointer doubling: k_wire_win_a — only when a payload has
    // more than one window — says where the chain enters each window, k_wire_win_b finds the records), then the serial walk for the
    // payloads that hold anything but plain records, and the numbering of the batch (by that launch's last workgroup; a launch of its
    // own for batches of many payloads)
    // (payloads of less than 1 KB have no windows: a wave walks three dozen records sooner than a workgroup sets up; GUBER_WIRE_SERIAL=1: the
    // serial walk for all, for A/B runs)
    static const bool wire_serial = [] { const char* v = guber_lab_env("GUBER_WIRE_SERIAL"); return v && atoi(v) != 0; }();
    if (!wire_serial) {
        HIPCHK(hipMemcpyAsync((void*)d->sc.wfirst, h_wfirst, (size_t)(nrpc + 1) * 4, hipMemcpyHostToDevice, st));
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
    uint32_t* h_first = h_len + d->max_rpcs; uint32_t* h_count = h_first + d->max_rpcs + 1;
    HIPCHK(hipMemcpyAsync(h_first, d->sc.first, (size_t)(nrpc + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_count, d->sc.count, (size_t)nrpc * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(d->h_status.p, d->sc.status, (size_t)nrpc * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    d->n_items = h_first[nrpc];
    *n_items = d->n_items;
    for (uint32_t r = 0; r < nrpc; ++r) {
        if (status) status[r] = d->h_status.p[r];
        if (first) first[r] = h_first[r];
        if (count) count[r] = d->h_status.p[r] == GUBER_OK ? h_count[r] : (d->h_status.p[r] == GUBER_E_WIRE_TOO_LARGE ? h_count[r] : 0);
    }
    return GUBER_OK;
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
    uint32_t* h_off = d->h_u32.p; uint32_t* h_len = h_off + d->max_rpcs; uint32_t* h_wfirst = h_off + 4 * (size_t)d->max_rpcs + 4;
    size_t pos = 0;
    uint32_t windows = 0;
    bool multi = false;                                                                    // a payload of more than one window: k_wire_win_a has something to say
    for (uint32_t r = 0; r < nrpc; ++r) {
        const uint32_t nw = guber::wire_windows_of(lens[r]);
        h_wfirst[r] = windows; windows += nw; multi = multi || nw > 1;
        pos = (pos + 15) & ~(size_t)15;
        if (pos + lens[r] + 16 > d->max_bytes) return fail(GUBER_E_WIRE_FULL, "payload bytes exceed the decoder's buffer");
        if (lens[r] && !msgs[r]) return fail(GUBER_E_INVALID_ARG, "null payload");
        memcpy(d->h_buf.p + pos, msgs[r], lens[r]);
        h_off[r] = (uint32_t)pos; h_len[r] = lens[r];
        pos += lens[r];
    }
    h_wfirst[nrpc] = windows;                                                              // (<= max_windows: the bytes fit)
    memset(d->h_buf.p + pos, 0, 16);
    return wire_dev_decode_staged_locked(d, nrpc, 0, pos + 16, windows, multi, is_owner, max_per_rpc, now_ms, status, first, count, n_items);
}

// The decoder's pinned staging buffer, and the decode of payloads that already lie in it: offs[r] 16-byte aligned, ascending, the
// payloads not overlapping, 16 bytes of room behind the last one (the kernels read whole 16-byte pieces: what lies there is never
// interpreted).  Everything else as guber_wire_dev_decode.
extern "C" int guber_wire_dev_buffer(guber_wire_dev_t* d, uint8_t** buf, size_t* cap) {
    if (!d || !buf || !cap) return fail(GUBER_E_INVALID_ARG, "null argument");
    *buf = d->h_buf.p; *cap = d->max_bytes;
    return GUBER_OK;
}
extern "C" int guber_wire_dev_decode_staged(guber_wire_dev_t* d, const uint32_t* offs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                            uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items) {
    if (!d || (nrpc && (!offs || !lens)) || !n_items) return fail(GUBER_E_INVALID_ARG, "null argument");
    *n_items = 0;
    if (nrpc > d->max_rpcs) return fail(GUBER_E_BATCH_TOO_LARGE, "more RPCs than the decoder was created for");
    guber_engine* e = d->e;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    d->nrpc = nrpc; d->n_items = 0; d->now_ms = now_ms;
    if (!nrpc) return GUBER_OK;
    uint32_t* h_off = d->h_u32.p; uint32_t* h_len = h_off + d->max_rpcs; uint32_t* h_wfirst = h_off + 4 * (size_t)d->max_rpcs + 4;
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
    return wire_dev_decode_staged_locked(d, nrpc, offs[0], end + 16, windows, multi, is_owner, max_per_rpc, now_ms, status, first, count, n_items);
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
extern "C" int guber_wire_dev_eval_front(guber_wire_dev_t* d, guber_front_t* f, guber_result_t* r) {
    if (!d || !f || !r) return fail(GUBER_E_INVALID_ARG, "null argument");
    const uint32_t n = d->n_items;
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
    if (!n) return GUBER_OK;
    if (!r->status || !r->limit || !r->remaining || !r->reset_time || !r->err) return fail(GUBER_E_INVALID_ARG, "result is missing an array");
    if (n > f->cap) return fail(GUBER_E_BATCH_TOO_LARGE, "more items than the front's generations hold");
    const size_t M = d->max_items;
    FrontGen g;
    memset(&g.b, 0, sizeof g.b);
    g.b.n = n; g.b.key_bytes = d->out.key_rows; g.b.hits = d->out.hits; g.b.limit = d->out.limit; g.b.duration = d->out.duration; g.b.burst = d->out.burst;
    g.b.created_at = d->out.created_at; g.b.algorithm = d->out.algorithm; g.b.behavior = d->out.behavior; g.b.is_owner = d->out.is_owner; g.b.now_ms = d->now_ms;
    g.key_stride = d->stride; g.key_len = d->out.key_len;
    guber_result_t dr{};
    dr.status = d->d_out8.p; dr.err = d->d_out8.p + M; dr.limit = d->d_out64.p; dr.remaining = d->d_out64.p + M; dr.reset_time = d->d_out64.p + 2 * M;
    {   // (the decode ran on the decoder's engine's stream and was synchronised there: guber_wire_dev_decode* read its verdicts back)
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
    hipStream_t st = e->stream;
    HIPCHK(hipMemcpyAsync(r->status, d->d_out8.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->err, d->d_out8.p + M, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->limit, d->d_out64.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->remaining, d->d_out64.p + M, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r->reset_time, d->d_out64.p + 2 * M, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
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
