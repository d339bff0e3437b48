// guber_wire_parse.h — the protobuf wire-format primitives of the request path (varints, unknown-field skipping, UTF-8
// validation, one RateLimitReq), written once and compiled for the host transcoder (wire.cpp), for the device decoder
// (guber_kernels_wire.h: k_wire_scan / k_wire_fill) and, in the memory-safety fuzz (tools/wire_fuzz_asan.cpp), under
// AddressSanitizer: every read is bounded by [p, end).
//   gubernator.proto:137-182  RateLimitReq: name 1, unique_key 2 (LEN); hits 3, limit 4, duration 5 (int64); algorithm 6,
//   behavior 7 (enum); burst 8 (int64); metadata 9 (map entry); created_at 10 (optional int64)
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(GUBER_FAKEHIP)
#include <hip/hip_runtime.h>
#define GW_HD __host__ __device__ inline
#else
#define GW_HD inline
#endif

namespace guber { namespace wire {

struct Span { const uint8_t* p = nullptr; uint32_t n = 0; };
struct ReqFields {
    Span name, unique_key;
    int64_t hits = 0, limit = 0, duration = 0, burst = 0, created_at = 0;
    int64_t algorithm = 0, behavior = 0;
};

GW_HD bool get_varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
    uint64_t r = 0;
    for (int shift = 0; shift < 70; shift += 7) {
        if (p >= end) return false;
        const uint8_t b = *p++;
        if (shift == 63 && (b & 0xfe)) return false;            // more than 64 bits
        r |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) { v = r; return true; }
    }
    return false;                                              // longer than 10 bytes
}
// Skip one unknown field whose tag (field number `field`, wire type `wt`) has just been read.  Groups (wire types 3 / 4, proto2
// leftovers) are skipped the way the protobuf runtimes do: everything up to the matching END_GROUP tag of the same field number,
// nested groups included (iteratively: the open groups' field numbers on a small stack; deeper than 64 = malformed).
GW_HD bool skip_field(const uint8_t*& p, const uint8_t* end, uint32_t wt, uint64_t field) {
    uint64_t v;
    switch (wt) {
    case 0: return get_varint(p, end, v);
    case 1: if (end - p < 8) return false; p += 8; return true;
    case 2: if (!get_varint(p, end, v) || (uint64_t)(end - p) < v) return false; p += v; return true;
    case 5: if (end - p < 4) return false; p += 4; return true;
    case 3: {
        uint32_t open[66];
        int depth = 0;
        open[depth++] = (uint32_t)field;
        while (depth > 0) {
            uint64_t tag;
            if (!get_varint(p, end, tag)) return false;
            const uint32_t w2 = (uint32_t)(tag & 7);
            const uint64_t f2 = tag >> 3;
            if (f2 == 0 || f2 > 0x1fffffffull) return false;
            if (w2 == 4) { if (f2 != open[depth - 1]) return false; --depth; continue; }   // END_GROUP must close THIS group
            if (w2 == 3) { if (depth >= 65) return false; open[depth++] = (uint32_t)f2; continue; }
            if (w2 == 0) { if (!get_varint(p, end, v)) return false; }
            else if (w2 == 1) { if (end - p < 8) return false; p += 8; }
            else if (w2 == 2) { if (!get_varint(p, end, v) || (uint64_t)(end - p) < v) return false; p += v; }
            else if (w2 == 5) { if (end - p < 4) return false; p += 4; }
            else return false;
        }
        return true;
    }
    default: return false;                                     // stray END_GROUP, wire types 6 / 7
    }
}
// proto3 string fields must be valid UTF-8 (the Go runtime rejects the message otherwise)
GW_HD bool valid_utf8(const uint8_t* s, uint32_t n) {
    uint32_t i = 0;
    while (i < n) {
        if (i + 8 <= n) {                                        // ASCII fast path, 8 bytes at a time
            uint64_t w;
            memcpy(&w, s + i, 8);
            if (!(w & 0x8080808080808080ull)) { i += 8; continue; }
        }
        const uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        uint32_t need, cp;
        if ((c & 0xe0) == 0xc0) { need = 1; cp = c & 0x1f; }
        else if ((c & 0xf0) == 0xe0) { need = 2; cp = c & 0x0f; }
        else if ((c & 0xf8) == 0xf0) { need = 3; cp = c & 0x07; }
        else return false;
        if (i + need >= n) return false;                        // truncated sequence
        for (uint32_t k = 1; k <= need; ++k) {
            const uint8_t cc = s[i + k];
            if ((cc & 0xc0) != 0x80) return false;
            cp = (cp << 6) | (cc & 0x3f);
        }
        if ((need == 1 && cp < 0x80) || (need == 2 && cp < 0x800) || (need == 3 && cp < 0x10000)) return false;   // overlong
        if (cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return false;
        i += need + 1;
    }
    return true;
}
GW_HD bool parse_req(const uint8_t* p, const uint8_t* end, ReqFields& f) {
    while (p < end) {
        uint64_t tag, v;
        if (*p < 0x80) tag = *p++;                                  // one-byte tag: every field of these messages
        else if (!get_varint(p, end, tag)) return false;
        const uint32_t wt = (uint32_t)(tag & 7);
        const uint64_t field = tag >> 3;
        if (field == 0 || field > 0x1fffffffull) return false;
        if ((field == 1 || field == 2) && wt == 2) {
            if (p < end && *p < 0x80) v = *p++;
            else if (!get_varint(p, end, v)) return false;
            if ((uint64_t)(end - p) < v || v > 0xffffffffull) return false;
            if (!valid_utf8(p, (uint32_t)v)) return false;
            Span& s = field == 1 ? f.name : f.unique_key;
            s.p = p; s.n = (uint32_t)v;
            p += v;
        } else if (wt == 0 && (field == 3 || field == 4 || field == 5 || field == 6 || field == 7 || field == 8 || field == 10)) {
            if (p < end && *p < 0x80) v = *p++;
            else if (!get_varint(p, end, v)) return false;
            switch (field) {
            case 3: f.hits = (int64_t)v; break;
            case 4: f.limit = (int64_t)v; break;
            case 5: f.duration = (int64_t)v; break;
            case 6: f.algorithm = (int64_t)(int32_t)v; break;      // enums are int32 on the wire (sign-extended varint)
            case 7: f.behavior = (int64_t)(int32_t)v; break;
            case 8: f.burst = (int64_t)v; break;
            case 10: f.created_at = (int64_t)v; break;
            }
        } else if (field == 9 && wt == 2) {
            // metadata map entry (string key = 1, string value = 2): not used on the path, but a malformed entry makes
            // the runtimes reject the whole message, so its structure and UTF-8 are checked
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
                } else if (!skip_field(q, qe, w2, f2)) {
                    return false;
                }
            }
            p = qe;
        } else if (!skip_field(p, end, wt, field)) {
            return false;
        }
    }
    return true;
}

}}  // namespace guber::wire
