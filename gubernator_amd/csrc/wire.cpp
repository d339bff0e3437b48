// wire.cpp — protobuf wire format <-> SoA batch transcoder (include/guber_wire.h).  Host code, no device work
// except guber_wire_eval (which is guber_eval_batch on the batch's own arrays).
//
// Reference behaviour restated here (not its code — the reference uses generated protobuf-go structs):
//   gubernator.proto:137-203, peers.proto:36-49   message layouts
//   gubernator.go:189-220                          batch cap, per-item validation, CreatedAt default
//   client.go:39-41                                key = name + "_" + unique_key
//   gubernator.go:250-255, workers.go:317-321      error texts of the evaluation
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/guber_wire.h"
#include "guber_wire_parse.h"

namespace {
using namespace guber::wire;

inline size_t varint_size(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; ++n; } return n; }
inline uint8_t* put_varint(uint8_t* p, uint64_t v) {
    while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; }
    *p++ = (uint8_t)v;
    return p;
}

}  // namespace

struct guber_wire_batch {
    uint32_t cap_items = 0, cap_keys = 0, flags = 0, n = 0, key_used = 0;
    int64_t now_ms = 0;
    bool any_greg = false;
    void* block = nullptr;                // one allocation for every array
    uint8_t* key_bytes = nullptr; uint32_t* key_off = nullptr;
    int64_t *hits = nullptr, *limit = nullptr, *duration = nullptr, *burst = nullptr, *created_at = nullptr;
    int64_t *greg_expire = nullptr, *greg_duration = nullptr;
    uint32_t* behavior = nullptr; int32_t* algo_raw = nullptr;
    uint8_t *algorithm = nullptr, *is_owner = nullptr, *pre_err = nullptr;
    int64_t *r_limit = nullptr, *r_remaining = nullptr, *r_reset = nullptr;
    uint8_t *r_status = nullptr, *r_err = nullptr;
    guber_batch_t view{};
    guber_result_t res{};
};

extern "C" int guber_wire_batch_create(uint32_t max_items, uint32_t max_key_bytes, uint32_t flags, guber_wire_batch_t** out) {
    if (!out || max_items == 0 || max_key_bytes == 0) return GUBER_E_INVALID_ARG;
    *out = nullptr;
    guber_wire_batch* b = new (std::nothrow) guber_wire_batch();
    if (!b) return GUBER_E_NOMEM;
    b->cap_items = max_items; b->cap_keys = max_key_bytes; b->flags = flags;
    const size_t M = max_items;
    // layout: 10 x int64[M] | u32 key_off[M+1], behavior[M] | i32 algo_raw[M] | 5 x u8[M] | key bytes (+16 readable past the end)
    const size_t bytes = 10 * 8 * M + 4 * (M + 1) + 4 * M + 4 * M + 5 * M + 64 + (size_t)max_key_bytes + 16;
    b->block = (flags & GUBER_WIRE_PINNED) ? guber_alloc_pinned(bytes) : calloc(1, bytes);
    if (!b->block) { delete b; return (flags & GUBER_WIRE_PINNED) ? GUBER_E_NO_DEVICE : GUBER_E_NOMEM; }
    memset(b->block, 0, bytes);
    uint8_t* p = (uint8_t*)b->block;
    int64_t** i64s[] = {&b->hits, &b->limit, &b->duration, &b->burst, &b->created_at, &b->greg_expire, &b->greg_duration,
                        &b->r_limit, &b->r_remaining, &b->r_reset};
    for (auto f : i64s) { *f = (int64_t*)p; p += 8 * M; }
    b->key_off = (uint32_t*)p; p += 4 * (M + 1);
    b->behavior = (uint32_t*)p; p += 4 * M;
    b->algo_raw = (int32_t*)p; p += 4 * M;
    uint8_t** u8s[] = {&b->algorithm, &b->is_owner, &b->pre_err, &b->r_status, &b->r_err};
    for (auto f : u8s) { *f = p; p += M; }
    p = (uint8_t*)(((uintptr_t)p + 63) & ~(uintptr_t)63);
    b->key_bytes = p;
    guber_wire_batch_reset(b, 0);
    *out = b;
    return GUBER_OK;
}

extern "C" void guber_wire_batch_destroy(guber_wire_batch_t* b) {
    if (!b) return;
    if (b->block) { if (b->flags & GUBER_WIRE_PINNED) guber_free_pinned(b->block); else free(b->block); }
    delete b;
}

extern "C" void guber_wire_batch_reset(guber_wire_batch_t* b, int64_t now_ms) {
    if (!b) return;
    b->n = 0; b->key_used = 0; b->now_ms = now_ms; b->any_greg = false;
    b->key_off[0] = 0;
}

extern "C" uint32_t guber_wire_batch_size(const guber_wire_batch_t* b) { return b ? b->n : 0; }

extern "C" int guber_wire_decode_requests(guber_wire_batch_t* b, const uint8_t* msg, size_t len, uint32_t max_per_rpc,
                                          uint8_t is_owner, uint32_t* first, uint32_t* count) {
    if (!b || (!msg && len) || !first || !count) return GUBER_E_INVALID_ARG;
    *first = b->n; *count = 0;
    // One pass over the payload, writing straight into the SoA at [n0, ...): the batch's item / key counters are only
    // advanced at the very end, so a malformed, over-long or non-fitting payload leaves the batch exactly as it was
    // (whatever was written beyond the counters is overwritten by the next successful decode).
    const uint32_t n0 = b->n, k0 = b->key_used;
    uint32_t i = n0, kused = k0;
    bool full = false, any_greg = b->any_greg;
    const uint8_t* p = msg; const uint8_t* end = msg + len;
    while (p < end) {
        uint64_t tag, v;
        if (*p < 0x80) tag = *p++;                                  // one-byte tag: the common case
        else if (!get_varint(p, end, tag)) return GUBER_E_WIRE_MALFORMED;
        const uint32_t wt = (uint32_t)(tag & 7);
        if ((tag >> 3) == 0 || (tag >> 3) > 0x1fffffffull) return GUBER_E_WIRE_MALFORMED;
        if ((tag >> 3) != 1 || wt != 2) {
            if (!skip_field(p, end, wt, tag >> 3)) return GUBER_E_WIRE_MALFORMED;
            continue;
        }
        if (p < end && *p < 0x80) v = *p++;
        else if (!get_varint(p, end, v)) return GUBER_E_WIRE_MALFORMED;
        if ((uint64_t)(end - p) < v) return GUBER_E_WIRE_MALFORMED;
        ReqFields f;
        if (!parse_req(p, p + v, f)) return GUBER_E_WIRE_MALFORMED;
        p += v;
        if (full) { ++i; continue; }                                // keep validating and counting; nothing is stored any more
        uint8_t pre = GUBER_WIRE_PRE_OK;
        if (f.unique_key.n == 0) pre = GUBER_WIRE_PRE_EMPTY_UNIQUE_KEY;          // gubernator.go:208-212
        else if (f.name.n == 0) pre = GUBER_WIRE_PRE_EMPTY_NAME;                 // gubernator.go:213-217
        const uint64_t klen = pre == GUBER_WIRE_PRE_OK ? (uint64_t)f.name.n + 1 + f.unique_key.n : 0;
        if (i >= b->cap_items || klen > (uint64_t)(b->cap_keys - kused)) { full = true; ++i; continue; }
        b->pre_err[i] = pre;
        if (klen) {                                                              // client.go:39-41
            uint8_t* k = b->key_bytes + kused;
            memcpy(k, f.name.p, f.name.n); k[f.name.n] = '_'; memcpy(k + f.name.n + 1, f.unique_key.p, f.unique_key.n);
            kused += (uint32_t)klen;
        }
        b->key_off[i + 1] = kused;
        b->hits[i] = f.hits; b->limit[i] = f.limit; b->duration[i] = f.duration; b->burst[i] = f.burst;
        b->created_at[i] = f.created_at ? f.created_at : b->now_ms;              // gubernator.go:218-220
        b->algo_raw[i] = (int32_t)f.algorithm;
        b->algorithm[i] = (f.algorithm == 0 || f.algorithm == 1) ? (uint8_t)f.algorithm : 255;
        b->behavior[i] = (uint32_t)f.behavior;
        b->is_owner[i] = is_owner ? 1 : 0;
        b->greg_expire[i] = 0; b->greg_duration[i] = 0;
        if (pre == GUBER_WIRE_PRE_OK && ((uint32_t)f.behavior & GUBER_BEHAVIOR_DURATION_IS_GREGORIAN)) {   // interval.go:84-148 at clock.Now()
            any_greg = true;
            int64_t e = 0, d = 0;
            int rc = guber_gregorian_expiration(b->now_ms * 1000000, f.duration, &e);
            if (rc == 0) rc = guber_gregorian_duration(b->now_ms * 1000000, f.duration, &d);
            b->greg_expire[i] = e; b->greg_duration[i] = rc ? rc : d;
        }
        ++i;
    }
    const uint32_t items = i - n0;
    if (max_per_rpc && items > max_per_rpc) { *count = items; return GUBER_E_WIRE_TOO_LARGE; }
    if (full) return GUBER_E_WIRE_FULL;
    *count = items;
    b->n = i; b->key_used = kused; b->any_greg = any_greg;
    memset(b->key_bytes + kused, 0, 16);
    return GUBER_OK;
}

extern "C" const guber_batch_t* guber_wire_batch_view(guber_wire_batch_t* b) {
    if (!b) return nullptr;
    guber_batch_t& v = b->view;
    v = guber_batch_t{};
    v.n = b->n; v.key_bytes = b->key_bytes; v.key_off = b->key_off; v.hits = b->hits; v.limit = b->limit;
    v.duration = b->duration; v.burst = b->burst; v.created_at = b->created_at; v.algorithm = b->algorithm;
    v.behavior = b->behavior; v.is_owner = b->is_owner; v.now_ms = b->now_ms;
    if (b->any_greg) { v.greg_expire = b->greg_expire; v.greg_duration = b->greg_duration; }
    return &v;
}

extern "C" guber_result_t* guber_wire_batch_result(guber_wire_batch_t* b) {
    if (!b) return nullptr;
    guber_result_t& r = b->res;
    r.status = b->r_status; r.limit = b->r_limit; r.remaining = b->r_remaining; r.reset_time = b->r_reset; r.err = b->r_err;
    return &r;
}

extern "C" const uint8_t* guber_wire_batch_pre_errors(const guber_wire_batch_t* b) { return b ? b->pre_err : nullptr; }

extern "C" int guber_wire_eval(guber_engine_t* e, guber_wire_batch_t* b) {
    if (!e || !b) return GUBER_E_INVALID_ARG;
    return guber_eval_batch(e, guber_wire_batch_view(b), guber_wire_batch_result(b));
}

namespace {
// error text of item i ("" = none)
std::string item_error(const guber_wire_batch* b, uint32_t i, int wrap) {
    if (b->pre_err[i] == GUBER_WIRE_PRE_EMPTY_UNIQUE_KEY) return "field 'unique_key' cannot be empty";
    if (b->pre_err[i] == GUBER_WIRE_PRE_EMPTY_NAME) return "field 'namespace' cannot be empty";
    const uint8_t code = b->r_err[i];
    if (code == GUBER_ITEM_OK) return std::string();
    char buf[256];
    if (code == GUBER_ITEM_E_INVALID_ALGORITHM) snprintf(buf, sizeof buf, guber_item_strerror(code), (int)b->algo_raw[i]);   // workers.go:318
    else snprintf(buf, sizeof buf, "%s", guber_item_strerror(code));
    if (!wrap) return buf;
    std::string s = "Error while apply rate limit for '";                                                                     // gubernator.go:250-255
    s.append((const char*)b->key_bytes + b->key_off[i], b->key_off[i + 1] - b->key_off[i]);
    s += "': "; s += buf;
    return s;
}
struct RespFields { uint64_t status, limit, remaining, reset; };
inline size_t resp_body_size(const RespFields& f, size_t err_len) {
    size_t n = 0;
    if (f.status) n += 1 + varint_size(f.status);
    if (f.limit) n += 1 + varint_size(f.limit);
    if (f.remaining) n += 1 + varint_size(f.remaining);
    if (f.reset) n += 1 + varint_size(f.reset);
    if (err_len) n += 1 + varint_size(err_len) + err_len;
    return n;
}
inline RespFields resp_fields(const guber_wire_batch* b, uint32_t i, bool has_error) {
    if (has_error) return RespFields{0, 0, 0, 0};              // the reference answers {Error: ...} only
    return RespFields{(uint64_t)b->r_status[i], (uint64_t)b->r_limit[i], (uint64_t)b->r_remaining[i], (uint64_t)b->r_reset[i]};
}
}  // namespace

extern "C" size_t guber_wire_encode_bound(const guber_wire_batch_t* b, uint32_t first, uint32_t count) {
    if (!b || first > b->n || count > b->n - first) return 0;
    // tag + length (<= 3) + 4 varint fields (<= 11 each) + error field: wrapper text + key + message (<= 256)
    size_t bound = 0;
    for (uint32_t i = first; i < first + count; ++i) bound += 4 + 44 + 3 + 40 + 256 + (b->key_off[i + 1] - b->key_off[i]);
    return bound;
}

extern "C" int guber_wire_encode_responses(const guber_wire_batch_t* b, uint32_t first, uint32_t count, int wrap_errors,
                                           uint8_t* out, size_t cap, size_t* len) {
    if (!b || !len || (!out && cap) || first > b->n || count > b->n - first) return GUBER_E_INVALID_ARG;
    size_t used = 0;
    bool overflow = false;
    for (uint32_t i = first; i < first + count; ++i) {
        std::string err;
        if (b->pre_err[i] != GUBER_WIRE_PRE_OK || b->r_err[i] != GUBER_ITEM_OK) err = item_error(b, i, wrap_errors);
        const RespFields f = resp_fields(b, i, !err.empty());
        const size_t body = resp_body_size(f, err.size());
        const size_t total = 1 + varint_size(body) + body;
        if (!overflow && used + total <= cap) {
            uint8_t* p = out + used;
            *p++ = 0x0a;                                        // field 1, LEN
            p = put_varint(p, body);
            if (f.status) { *p++ = 0x08; p = put_varint(p, f.status); }
            if (f.limit) { *p++ = 0x10; p = put_varint(p, f.limit); }
            if (f.remaining) { *p++ = 0x18; p = put_varint(p, f.remaining); }
            if (f.reset) { *p++ = 0x20; p = put_varint(p, f.reset); }
            if (!err.empty()) { *p++ = 0x2a; p = put_varint(p, err.size()); memcpy(p, err.data(), err.size()); p += err.size(); }
        } else {
            overflow = true;
        }
        used += total;
    }
    *len = used;
    return overflow ? GUBER_E_NOMEM : GUBER_OK;
}

// ---- UpdatePeerGlobals ---------------------------------------------------------------------------------------------
struct guber_wire_items {
    uint32_t cap_items = 0, cap_keys = 0;
    std::vector<guber_item_t> items;
    std::vector<uint8_t> keys;
};

extern "C" int guber_wire_items_create(uint32_t max_items, uint32_t max_key_bytes, guber_wire_items_t** out) {
    if (!out || max_items == 0 || max_key_bytes == 0) return GUBER_E_INVALID_ARG;
    *out = nullptr;
    guber_wire_items* w = new (std::nothrow) guber_wire_items();
    if (!w) return GUBER_E_NOMEM;
    try { w->items.resize(max_items); w->keys.resize((size_t)max_key_bytes + 16); } catch (...) { delete w; return GUBER_E_NOMEM; }
    w->cap_items = max_items; w->cap_keys = max_key_bytes;
    *out = w;
    return GUBER_OK;
}
extern "C" void guber_wire_items_destroy(guber_wire_items_t* w) { delete w; }

namespace {
struct RespFieldsIn { int64_t status = 0, limit = 0, remaining = 0, reset_time = 0; };
// RateLimitResp (gubernator.proto:189-203) as it arrives inside UpdatePeerGlobal.status; error / metadata are validated and ignored
bool parse_resp(const uint8_t* p, const uint8_t* end, RespFieldsIn& f) {
    while (p < end) {
        uint64_t tag, v;
        if (!get_varint(p, end, tag)) return false;
        const uint32_t wt = (uint32_t)(tag & 7);
        const uint64_t field = tag >> 3;
        if (field == 0 || field > 0x1fffffffull) return false;
        if (wt == 0 && field >= 1 && field <= 4) {
            if (!get_varint(p, end, v)) return false;
            if (field == 1) f.status = (int64_t)(int32_t)v;
            else if (field == 2) f.limit = (int64_t)v;
            else if (field == 3) f.remaining = (int64_t)v;
            else f.reset_time = (int64_t)v;
        } else if (field == 5 && wt == 2) {
            if (!get_varint(p, end, v) || (uint64_t)(end - p) < v || v > 0xffffffffull || !valid_utf8(p, (uint32_t)v)) return false;
            p += v;
        } else if (field == 6 && wt == 2) {                       // metadata map entry: structure and UTF-8 as the runtimes check them
            if (!get_varint(p, end, v) || (uint64_t)(end - p) < v) return false;
            const uint8_t* q = p; const uint8_t* qe = p + v;
            while (q < qe) {
                uint64_t t2, l2;
                if (!get_varint(q, qe, t2)) return false;
                const uint32_t w2 = (uint32_t)(t2 & 7);
                const uint64_t f2 = t2 >> 3;
                if (f2 == 0 || f2 > 0x1fffffffull) return false;
                if ((f2 == 1 || f2 == 2) && w2 == 2) {
                    if (!get_varint(q, qe, l2) || (uint64_t)(qe - q) < l2 || l2 > 0xffffffffull || !valid_utf8(q, (uint32_t)l2)) return false;
                    q += l2;
                } else if (!skip_field(q, qe, w2, f2)) return false;
            }
            p = qe;
        } else if (!skip_field(p, end, wt, field)) {
            return false;
        }
    }
    return true;
}
}  // namespace

extern "C" int guber_wire_decode_globals(guber_wire_items_t* w, const uint8_t* msg, size_t len, int64_t now_ms,
                                         const guber_item_t** out, uint32_t* count) {
    if (!w || (!msg && len) || !out || !count) return GUBER_E_INVALID_ARG;
    *out = w->items.data(); *count = 0;
    uint32_t n = 0, kused = 0;
    const uint8_t* p = msg; const uint8_t* end = msg + len;
    while (p < end) {
        uint64_t tag, v;
        if (!get_varint(p, end, tag)) return GUBER_E_WIRE_MALFORMED;
        const uint32_t wt = (uint32_t)(tag & 7);
        if ((tag >> 3) == 0 || (tag >> 3) > 0x1fffffffull) return GUBER_E_WIRE_MALFORMED;
        if ((tag >> 3) != 1 || wt != 2) {
            if (!skip_field(p, end, wt, tag >> 3)) return GUBER_E_WIRE_MALFORMED;
            continue;
        }
        if (!get_varint(p, end, v) || (uint64_t)(end - p) < v) return GUBER_E_WIRE_MALFORMED;
        const uint8_t* q = p; const uint8_t* qe = p + v;
        p = qe;
        Span key; RespFieldsIn st; int64_t algorithm = 0, duration = 0;
        while (q < qe) {                                          // UpdatePeerGlobal
            uint64_t t2, x;
            if (!get_varint(q, qe, t2)) return GUBER_E_WIRE_MALFORMED;
            const uint32_t w2 = (uint32_t)(t2 & 7);
            const uint64_t f2 = t2 >> 3;
            if (f2 == 0 || f2 > 0x1fffffffull) return GUBER_E_WIRE_MALFORMED;
            if (f2 == 1 && w2 == 2) {
                if (!get_varint(q, qe, x) || (uint64_t)(qe - q) < x || x > 0xffffffffull || !valid_utf8(q, (uint32_t)x)) return GUBER_E_WIRE_MALFORMED;
                key.p = q; key.n = (uint32_t)x; q += x;
            } else if (f2 == 2 && w2 == 2) {
                if (!get_varint(q, qe, x) || (uint64_t)(qe - q) < x) return GUBER_E_WIRE_MALFORMED;
                // a repeated occurrence of a singular message field merges into the previous one: parse into the same struct
                if (!parse_resp(q, q + x, st)) return GUBER_E_WIRE_MALFORMED;
                q += x;
            } else if (w2 == 0 && (f2 == 3 || f2 == 4 || f2 == 5)) {
                if (!get_varint(q, qe, x)) return GUBER_E_WIRE_MALFORMED;
                if (f2 == 3) algorithm = (int64_t)(int32_t)x;
                else if (f2 == 4) duration = (int64_t)x;          // 5 = created_at: not used by the receiver (gubernator.go:427 takes now)
            } else if (!skip_field(q, qe, w2, f2)) {
                return GUBER_E_WIRE_MALFORMED;
            }
        }
        if (n >= w->cap_items || key.n > w->cap_keys - kused) return GUBER_E_WIRE_FULL;
        guber_item_t& it = w->items[n];
        memset(&it, 0, sizeof(it));
        if (key.n) memcpy(w->keys.data() + kused, key.p, key.n);
        it.key = w->keys.data() + kused; it.key_len = key.n;
        kused += key.n;
        it.expire_at = st.reset_time;                                             // gubernator.go:430
        if (algorithm == GUBER_ALGO_LEAKY_BUCKET) {                               // :435-442
            it.algorithm = GUBER_ALGO_LEAKY_BUCKET;
            it.remaining_f = (double)st.remaining; it.limit = st.limit; it.duration = duration; it.burst = st.limit; it.stamp = now_ms;
        } else if (algorithm == GUBER_ALGO_TOKEN_BUCKET) {                        // :443-451
            it.algorithm = GUBER_ALGO_TOKEN_BUCKET;
            it.status = (uint8_t)st.status; it.limit = st.limit; it.duration = duration; it.remaining = st.remaining; it.stamp = now_ms;
        } else {
            it.algorithm = (uint8_t)(algorithm < 0 || algorithm > 254 ? 255 : algorithm);   // no Value: the switch at :434 matches nothing
        }
        ++n;
    }
    *count = n;
    return GUBER_OK;
}

extern "C" int guber_wire_encode_globals(const uint8_t* key_bytes, const uint32_t* key_off, const uint8_t* algorithm,
                                         const int64_t* duration, const int64_t* created_at, const guber_result_t* status, uint32_t n,
                                         uint8_t* out, size_t cap, size_t* len) {
    if (!len || (n && (!key_bytes || !key_off || !algorithm || !duration || !created_at || !status)) || (!out && cap)) return GUBER_E_INVALID_ARG;
    size_t used = 0;
    bool overflow = false;
    for (uint32_t i = 0; i < n; ++i) {
        if (status->err[i] != GUBER_ITEM_OK) continue;                            // global.go:246-249: status read failed -> skipped
        const RespFields f{(uint64_t)status->status[i], (uint64_t)status->limit[i], (uint64_t)status->remaining[i], (uint64_t)status->reset_time[i]};
        const size_t sbody = resp_body_size(f, 0);
        const uint32_t klen = key_off[i + 1] - key_off[i];
        size_t body = 0;
        if (klen) body += 1 + varint_size(klen) + klen;
        body += 1 + varint_size(sbody) + sbody;                                   // the status message is always present (non-nil pointer)
        if (algorithm[i]) body += 1 + varint_size(algorithm[i]);
        if (duration[i]) body += 1 + varint_size((uint64_t)duration[i]);
        if (created_at[i]) body += 1 + varint_size((uint64_t)created_at[i]);
        const size_t total = 1 + varint_size(body) + body;
        if (!overflow && used + total <= cap) {
            uint8_t* p = out + used;
            *p++ = 0x0a; p = put_varint(p, body);
            if (klen) { *p++ = 0x0a; p = put_varint(p, klen); memcpy(p, key_bytes + key_off[i], klen); p += klen; }
            *p++ = 0x12; p = put_varint(p, sbody);
            if (f.status) { *p++ = 0x08; p = put_varint(p, f.status); }
            if (f.limit) { *p++ = 0x10; p = put_varint(p, f.limit); }
            if (f.remaining) { *p++ = 0x18; p = put_varint(p, f.remaining); }
            if (f.reset) { *p++ = 0x20; p = put_varint(p, f.reset); }
            if (algorithm[i]) { *p++ = 0x18; p = put_varint(p, algorithm[i]); }
            if (duration[i]) { *p++ = 0x20; p = put_varint(p, (uint64_t)duration[i]); }
            if (created_at[i]) { *p++ = 0x28; p = put_varint(p, (uint64_t)created_at[i]); }
        } else {
            overflow = true;
        }
        used += total;
    }
    *len = used;
    return overflow ? GUBER_E_NOMEM : GUBER_OK;
}
