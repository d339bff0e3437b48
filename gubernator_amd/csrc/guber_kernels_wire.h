// guber_kernels_wire.h — the protobuf wire format decoded ON THE DEVICE: serialized GetRateLimitsReq / GetPeerRateLimitsReq payloads
// (gubernator.proto:137-182, peers.proto:36-44: `repeated RateLimitReq = 1`) straight into the structure of arrays the batch
// pipelines evaluate, with the per-item validation of V1Instance.GetRateLimits (gubernator.go:189-220).  The host transcoder
// (wire.cpp) does 21 M items/s per thread; the kernels it feeds do billions.
//
// A protobuf message is a chain: where record k+1 starts is known only after record k's length has been read.  So:
//   k_wire_scan   one wave per RPC payload walks that chain — headers only (tag + length, 2-3 bytes per record), read from a window
//                 of the payload staged in LDS, every lane computing the same thing (broadcast reads, no divergence) — and leaves
//                 every record's offset and length, the RPC's item count and its status (ok / malformed / too many items);
//                 (the usual payload — nothing but plain records — has its chain found in parallel first: k_wire_win_a / k_wire_win_b below);
//                 the launch's last workgroup: where each RPC's items start in the batch (exclusive scan of the counts of the RPCs that are ok);
//   k_wire_fill   one thread per item: the record body through guber::wire::parse_req — the SAME source the host transcoder and its
//                 AddressSanitizer fuzz compile (guber_wire_parse.h: every read bounded by the record) — into the arrays: HashKey
//                 bytes `name + "_" + unique_key` (client.go:39-41) as one row per item, the CreatedAt default, the algorithm code,
//                 the validation outcome.  A record whose body is malformed marks its whole RPC malformed, as the runtimes reject the
//                 whole message; its items stay in place as dead slots (empty key: they never reach a bucket).
// The top-level framing decisions live in scan_toplevel(), a host/device template over a byte source, so that the fuzz runs them too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "guber_wire_parse.h"

namespace guber {

constexpr int32_t WIRE_OK = 0, WIRE_MALFORMED = -20, WIRE_TOO_LARGE = -21;      // = GUBER_E_WIRE_* (include/guber_wire.h)
constexpr uint8_t WIRE_PRE_DEAD = 255;                                           // an item of an RPC that turned out malformed

// The top level of one payload: field 1 (LEN) = one RateLimitReq; anything else is an unknown field, skipped by wire type.
// Rd::peek8(pos) = the 8 bytes at pos (zero beyond the payload); Rd::slow(pos) = a plain pointer into the payload for the rare
// constructs that are walked byte by byte.  emit(k, body_offset, body_len) is called for every record, in order.
// scan_one: ONE top-level field at `pos` (advances pos; counts a record); scan_toplevel: the whole payload.
template <class Rd, class Emit>
GW_HD int32_t scan_one(Rd& rd, uint32_t len, uint32_t& pos, uint32_t max_items, uint32_t& count, Emit& emit) {
    uint64_t w = rd.peek8(pos);
    uint64_t tag; uint32_t used;
    if (!(w & 0x80)) { tag = w & 0x7f; used = 1; }
    else {                                                           // a multi-byte tag: by the book
        const uint8_t* p = rd.slow(pos); const uint8_t* e = rd.slow(len);
        if (!wire::get_varint(p, e, tag)) return WIRE_MALFORMED;
        used = (uint32_t)(p - rd.slow(pos));
        w = rd.peek8(pos + used) << 8;                               // (keeps "w >> 8" below = the bytes after the tag)
    }
    const uint32_t wt = (uint32_t)(tag & 7);
    const uint64_t field = tag >> 3;
    if (field == 0 || field > 0x1fffffffull) return WIRE_MALFORMED;
    if (field != 1 || wt != 2) {                                     // unknown top-level field
        const uint8_t* p = rd.slow(pos + used); const uint8_t* e = rd.slow(len);
        if (!wire::skip_field(p, e, wt, field)) return WIRE_MALFORMED;
        pos = (uint32_t)(p - rd.slow(0));
        return WIRE_OK;
    }
    uint64_t L; uint32_t lused;
    const uint64_t v = w >> 8;                                       // the bytes after the tag
    if (!(v & 0x80)) { L = v & 0x7f; lused = 1; }
    else if (!(v & 0x8000)) { L = (v & 0x7f) | ((v >> 1) & 0x3f80); lused = 2; }
    else {
        const uint8_t* p = rd.slow(pos + used); const uint8_t* e = rd.slow(len);
        if (!wire::get_varint(p, e, L)) return WIRE_MALFORMED;
        lused = (uint32_t)(p - rd.slow(pos + used));
    }
    const uint64_t body = (uint64_t)pos + used + lused;
    if (body > len || L > (uint64_t)len - body) return WIRE_MALFORMED;
    if (count < max_items) emit(count, (uint32_t)body, (uint32_t)L);
    ++count;
    pos = (uint32_t)(body + L);
    return WIRE_OK;
}
template <class Rd, class Emit>
GW_HD int32_t scan_toplevel(Rd& rd, uint32_t len, uint32_t max_items, uint32_t& count, Emit emit) {
    uint32_t pos = 0;
    count = 0;
    while (pos < len) {
        const int32_t st = scan_one(rd, len, pos, max_items, count, emit);
        if (st != WIRE_OK) return st;
    }
    return WIRE_OK;
}
// a byte source over plain memory (the device's rare path: constructs the windowed walk does not shortcut; the host's only path)
struct MemReader {
    const uint8_t* g; uint32_t len;
    GW_HD uint64_t peek8(uint32_t pos) const { uint64_t w = 0; for (uint32_t k = 0; k < 8 && pos + k < len; ++k) w |= (uint64_t)g[pos + k] << (8 * k); return w; }
    GW_HD const uint8_t* slow(uint32_t pos) const { return g + pos; }
};

struct WireIn {
    const uint8_t* buf;             // the payloads back to back (16 readable bytes past the end)
    const uint32_t* rpc_off;        // [nrpc] where each payload starts in buf (16-byte aligned)
    const uint32_t* rpc_len;        // [nrpc]
    const uint8_t* rpc_owner;       // [nrpc] RateLimitReqState.IsOwner of the RPC's items
    uint32_t nrpc, cap_per_rpc;     // records per RPC the scratch holds
    uint32_t max_per_rpc;           // 0 = no cap (gubernator.go:40 passes 1000)
    uint32_t cap_items;             // items the output arrays hold
};
struct WireScratch {
    uint32_t* rec_off; uint32_t* rec_len; uint32_t* count; int32_t* status; uint32_t* first;
    const uint32_t* wfirst;         // [nrpc + 1] the windows of the payloads, numbered through the batch (k_wire_win_*): where each payload's start
    uint2* went;                    // [windows][WP_ENT] k_wire_win_a -> k_wire_win_b
    uint32_t* done;                 // workgroups of k_wire_scan that have finished (zero between launches)
};
struct WireOut {
    uint8_t* key_rows; uint32_t key_stride; uint32_t* key_len;
    int64_t *hits, *limit, *duration, *burst, *created_at; uint32_t* behavior; int32_t* algo_raw;
    uint8_t *algorithm, *is_owner, *pre_err; uint32_t* item_rpc;
    int64_t now_ms;
    // the verdicts in DEVICE-VISIBLE HOST memory, written by the decode's last kernel (no copies behind it): first[nrpc + 1] | count[nrpc] | status[nrpc]
    uint32_t* rep_first; uint32_t* rep_count; int32_t* rep_status;
};

constexpr uint32_t WIRE_WIN = 8192;
// One wave per RPC payload.  Every lane walks the same chain; what it reads from the LDS window goes through readfirstlane, so the
// walk runs on the scalar unit (uniform branches, no exec-mask bookkeeping).  The usual record — tag 0x0a, a one- or two-byte
// length — is decided from four bytes; everything else (multi-byte tags, unknown fields, long lengths) goes through scan_one over
// plain memory: the shared, fuzzed code.
__device__ __forceinline__ void wire_scan_body(const WireIn& in, const WireScratch& sc) {
    __shared__ uint32_t win[WIRE_WIN / 4 + 8];
    const uint32_t r = blockIdx.x;
    const uint32_t off = in.rpc_off[r], len = in.rpc_len[r];
    const uint8_t* g = in.buf + off;
    const uint32_t lim = (len + 15u) & ~15u;                         // whole 16-byte chunks: the bytes past the payload are padding or the next payload's
    uint32_t* ro = sc.rec_off + (size_t)r * in.cap_per_rpc; uint32_t* rl = sc.rec_len + (size_t)r * in.cap_per_rpc;
    uint32_t pos = 0, count = 0, wbase = 0, wend = 0;
    int32_t st = WIRE_OK;
    // records are collected in LDS and leave in whole lines of 64 (one coalesced store per line): no store sits in the walk's chain
    __shared__ uint32_t lro[64], lrl[64];
    auto flush = [&](uint32_t upto) {                                // records [upto - n, upto) of the line, n = ((upto - 1) & 63) + 1
        const uint32_t base = (upto - 1u) & ~63u, n = upto - base;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        if (threadIdx.x < n) { ro[base + threadIdx.x] = lro[threadIdx.x]; rl[base + threadIdx.x] = lrl[threadIdx.x]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier();
    };
    auto emit = [&](uint32_t k, uint32_t bo, uint32_t bl) {
        lro[k & 63u] = off + bo; lrl[k & 63u] = bl;                  // (every lane writes the same value)
        if ((k & 63u) == 63u) flush(k + 1u);
    };
    while (pos < len) {
        if (pos < wbase || pos + 4 > wend) {                         // the walk left the window: the whole wave fetches the next one
            wbase = pos & ~15u; wend = wbase + WIRE_WIN;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier();
            uint4 v[WIRE_WIN / (64 * 16)];
#pragma unroll
            for (uint32_t c = 0; c < WIRE_WIN / (64 * 16); ++c) {
                const uint32_t o = (c * 64 + threadIdx.x) * 16;
                v[c] = make_uint4(0, 0, 0, 0);
                if (wbase + o < lim) v[c] = *(const uint4*)(g + wbase + o);
            }
#pragma unroll
            for (uint32_t c = 0; c < WIRE_WIN / (64 * 16); ++c) *(uint4*)((unsigned char*)win + (c * 64 + threadIdx.x) * 16) = v[c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        const uint32_t o = pos - wbase;
        const uint32_t avail = len - pos;
        uint32_t L = 0, hdr = 0;
        {
            const uint32_t w0 = win[o >> 2], w1 = win[(o >> 2) + 1];
            uint32_t h = (uint32_t)((((unsigned long long)w1 << 32) | w0) >> ((o & 3u) * 8));
            h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
            if (avail < 4) h &= (1u << (8 * avail)) - 1u;             // zero beyond the payload
            if ((h & 0xffu) == 0x0au) {
                if (!(h & 0x8000u)) { L = (h >> 8) & 0x7fu; hdr = 2; }
                else if (!(h & 0x800000u)) { L = ((h >> 8) & 0x7fu) | ((h >> 9) & 0x3f80u); hdr = 3; }
            }
        }
        if (hdr) {
            if (hdr > avail || L > avail - hdr) { st = WIRE_MALFORMED; break; }
            if (count < in.cap_per_rpc) emit(count, pos + hdr, L);
            ++count;
            pos += hdr + L;
        } else {
            MemReader rd{g, len};
            st = scan_one(rd, len, pos, in.cap_per_rpc, count, emit);
            pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos); count = (uint32_t)__builtin_amdgcn_readfirstlane((int)count);
            if (st != WIRE_OK) break;
        }
    }
    {
        const uint32_t kept = count < in.cap_per_rpc ? count : in.cap_per_rpc;
        if (kept & 63u) flush(kept);                                 // the last, partial line
    }
    if (st == WIRE_OK && ((in.max_per_rpc && count > in.max_per_rpc) || count > in.cap_per_rpc)) st = WIRE_TOO_LARGE;   // gubernator.go:189-193
    if (threadIdx.x == 0) { sc.count[r] = count; sc.status[r] = st; }
}
// (the kernel: k_wire_scan below — it also numbers the batch)

// ---- the chain of a payload found IN PARALLEL (round 5) ---------------------------------------------------------------------------
// The serial walk above costs one dependent LDS read per record: 255 ns each with 64 waves on the chip, 255 us for a batch of 64
// RPCs of 1 000 items (the reference's maximum, gubernator.go:40) — 230 M items/s where the kernels behind it take billions.  Here a
// payload is cut into WINDOWS of 8 KB at fixed positions and every window gets a WORKGROUP:
//   * every position of the window decides what the chain would do IF it came by ("a record of the usual form starts here: header
//     2 | 3 bytes, body L", from the same four bytes the serial walk looks at) — 8 192 positions at once;
//   * next[o] = where the chain goes from o; the positions the chain really visits — those reachable from the position the chain
//     ENTERS the window at — are found by pointer doubling: in round k every visited position marks the one 2^k steps on and every
//     position's pointer is squared (next = next o next); a window of 1 000-item RPCs (~290 records) is done in nine rounds of 32 LDS
//     steps per thread instead of 290 dependent ones;
//   * the visited positions in ascending order ARE the records in order: a popcount prefix over the 128 words of the bitmap numbers them.
// Where the chain enters a window, and how many records came before, is what the window before it knows.  So that the windows of a
// payload need not wait for each other, a first launch (k_wire_win_a) answers that question for EVERY position a chain could enter
// at: the same pointer doubling without the marks, with the number of records riding on the pointers ({where, how many} composed:
// exact whatever order the words are updated in, so it runs in place), leaves for each of a window's first WP_ENT positions where a
// chain entering THERE leaves the window and how many records it passes.  The second launch (k_wire_win_b: a workgroup per window
// again) follows those answers from position 0 of the payload through the windows before its own — one thread, one dependent load per
// window, beside the loads of its window — and then marks, numbers and writes its own records: four windows of a 1 000-item payload
// run side by side instead of one after the other.
// Anything that is not the usual form where the chain really passes — a multi-byte tag, an unknown field, a length of three or more
// bytes, a record that overruns the payload, the last three bytes of a payload, a record of more than WP_ENT bytes across a window's
// edge — leaves the WHOLE payload to the serial walk (status WIRE_SERIAL, set by the workgroup of the payload's last window, which
// sees every window's answer; k_wire_scan runs behind this kernel for exactly those): one source of truth for every verdict that is
// not "ok" and for every unusual construct (scan_one, shared with the host transcoder and its fuzz).  The windows in front of such a
// place have written their records by then: the same values the serial walk writes again.
constexpr int32_t WIRE_SERIAL = 1;
constexpr uint32_t WP_WIN = 8192;                                  // bytes per window (the window's workgroup also reads the 16 bytes behind it)
constexpr uint32_t WP_ENT = 1024;                                  // positions of a window a chain may enter at (= the longest record across an edge)
constexpr uint32_t WP_T = 1024;                                    // threads per window: sixteen waves, four per SIMD — the rounds are chains of dependent LDS reads, other waves fill the waits
constexpr uint32_t WP_NONE = 0xffffffffu, WP_SKIP = 0xfffffffeu;
constexpr uint32_t WP_MIN = 1024;                                  // a payload shorter than this (three dozen records) is the serial walk's from the start: a wave is enough for it
GW_HD uint32_t wire_windows_of(uint32_t len) { return len < WP_MIN ? 0u : len / WP_WIN + 1u; }     // (position `len`, the chain's end, lies in the last window)
// inclusive prefix sum over a wave's 64 lanes, all active (this header stands alone — the host transcoder's fuzz compiles it without
// the batch kernels' headers — so it carries its own copy of guber_table.h's wave_incl_scan_i32: DPP on the device, shuffles elsewhere)
__device__ __forceinline__ uint32_t wire_wave_incl_scan(uint32_t x) {
    int v = (int)x;
#if defined(__HIP_DEVICE_COMPILE__)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
#else
    const int lane = (int)(threadIdx.x & 63);
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, (unsigned)o, 64); if (lane >= o) v += t; }
#endif
    return (uint32_t)v;
}
__device__ __forceinline__ void wire_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// the workgroup's window: blockIdx -> (payload, window) through wfirst[] (where each payload's windows start: the host knows the lengths)
struct WireWindow {
    uint32_t r, w, nwin, off, len, wbase, rest;                    // rest = payload bytes from the window's first position on
    const uint8_t* g;
};
__device__ __forceinline__ WireWindow wire_window(const WireIn& in, const WireScratch& sc) {
    uint32_t lo = 0, hi = in.nrpc;                                   // the last payload whose first window is <= blockIdx: it has windows, the next one's start behind blockIdx
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sc.wfirst[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    WireWindow W;
    W.r = lo; W.w = blockIdx.x - sc.wfirst[lo];
    W.off = in.rpc_off[lo]; W.len = in.rpc_len[lo];
    W.nwin = W.len / WP_WIN + 1u;
    W.wbase = W.w * WP_WIN; W.rest = W.len - W.wbase;
    W.g = in.buf + W.off;
    return W;
}
// the window's bytes + the 16 behind them (a position decides from its four bytes) -> LDS; zero beyond the payload's 16-byte chunks
__device__ __forceinline__ void wire_window_load(const WireWindow& W, uint32_t* win) {
    const uint32_t t = threadIdx.x, lim = (W.len + 15u) & ~15u;
    if (t <= WP_WIN / 16) {
        const uint32_t o = t * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o < W.rest + 16u && W.wbase + o < lim) v = *(const uint4*)(W.g + W.wbase + o);
        *(uint4*)((unsigned char*)win + o) = v;
    }
    if (t < 4) win[WP_WIN / 4 + 4 + t] = 0u;
}
// what position o is to the chain: a record followed by a position of this window / a record that leaves the window / the payload's
// end / anything else.  The usual form is decided from the four bytes at o exactly as the serial walk decides it.
enum : uint32_t { N_REC_NEXT = 0, N_REC_EXIT = 1, N_END = 3, N_BAD = 4 };
__device__ __forceinline__ uint32_t wire_node(const WireWindow& W, const uint32_t* win, uint32_t o, uint32_t& hdr, uint32_t& L, uint32_t& nxt) {
    hdr = L = 0; nxt = o;
    if (o >= W.rest) return o == W.rest ? N_END : N_BAD;
    if (W.rest - o < 4) return N_BAD;                              // (a payload's last three bytes: the serial walk's)
    const uint32_t w0 = win[o >> 2], w1 = win[(o >> 2) + 1];
    const uint32_t h = (uint32_t)((((unsigned long long)w1 << 32) | w0) >> ((o & 3u) * 8));
    if ((h & 0xffu) != 0x0au) return N_BAD;
    if (!(h & 0x8000u)) { L = (h >> 8) & 0x7fu; hdr = 2; }
    else if (!(h & 0x800000u)) { L = ((h >> 8) & 0x7fu) | ((h >> 9) & 0x3f80u); hdr = 3; }
    else return N_BAD;
    if (hdr + L > W.rest - o) return N_BAD;                        // overruns the payload: the serial walk says malformed
    nxt = o + hdr + L;
    return nxt < WP_WIN ? N_REC_NEXT : N_REC_EXIT;
}

// first launch: for every window but a payload's last, went[window][p] = {where a chain that enters at p leaves (a position of the
// payload; WP_NONE: it meets something the serial walk must see), records it passes}, p < WP_ENT
__global__ __launch_bounds__(WP_T) void k_wire_win_a(WireIn in, WireScratch sc) {
    __shared__ alignas(16) uint32_t win[WP_WIN / 4 + 8];
    __shared__ uint32_t JC[WP_WIN];                                 // position -> where 2^k or more steps on (a terminal: itself) | records passed << 16
    __shared__ uint32_t s_more[3];
    static_assert(WP_T > WP_WIN / 16 && WP_WIN % WP_T == 0 && WP_ENT == WP_T, "a thread per 16-byte chunk, per entry");
    const WireWindow W = wire_window(in, sc);
    if (W.w + 1u >= W.nwin) return;                                  // nothing comes after a payload's last window
    const uint32_t t = threadIdx.x;
    constexpr uint32_t NB = WP_WIN / WP_T;
    wire_window_load(W, win);
    if (t == 0) s_more[0] = 0u;
    wire_barrier();
#pragma unroll
    for (uint32_t b = 0; b < NB; ++b) {
        const uint32_t o = b * WP_T + t;
        uint32_t hdr, L, nxt;
        JC[o] = wire_node(W, win, o, hdr, L, nxt) == N_REC_NEXT ? (nxt | (1u << 16)) : o;
    }
    wire_barrier();
    for (uint32_t round = 0;; ++round) {
        if (t == 0) s_more[(round + 1) % 3] = 0u;
        bool more = false;
#pragma unroll
        for (uint32_t b0 = 0; b0 < NB; b0 += 4) {
            uint32_t a[4], c[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) a[u] = JC[(b0 + u) * WP_T + t];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) c[u] = JC[a[u] & 0xffffu];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t nv = (c[u] & 0xffffu) | ((a[u] & 0xffff0000u) + (c[u] & 0xffff0000u));
                if (nv != a[u]) { JC[(b0 + u) * WP_T + t] = nv; if (b0 + u == 0) more = true; }   // (done when the ENTRIES stand still: each is at a terminal then)
            }
        }
        if (more) s_more[round % 3] = 1u;
        wire_barrier();
        if (!s_more[round % 3]) break;                               // (uniform: read after the barrier; the flag is reused three rounds later)
    }
    {
        const uint32_t v = JC[t], tn = v & 0xffffu;
        uint32_t hdr, L, nxt;
        uint2 e = make_uint2(WP_NONE, 0u);
        if (wire_node(W, win, tn, hdr, L, nxt) == N_REC_EXIT) e = make_uint2(W.wbase + nxt, (v >> 16) + 1u);
        sc.went[(size_t)blockIdx.x * WP_ENT + t] = e;
    }
}

// second launch: the window's records
__global__ __launch_bounds__(WP_T) void k_wire_win_b(WireIn in, WireScratch sc) {
    __shared__ alignas(16) uint32_t win[WP_WIN / 4 + 8];
    __shared__ uint16_t J[2][WP_WIN];                             // position -> the position 2^k steps on (a terminal points at itself)
    __shared__ unsigned long long reach[WP_WIN / 64];             // the positions the chain visits
    __shared__ uint32_t wsum[WP_T / 64], s_more[3], s_bad, s_entry, s_base;
    static_assert(WP_T >= WP_WIN / 64, "a thread per bitmap word");
    const WireWindow W = wire_window(in, sc);
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const bool last = W.w + 1u == W.nwin;
    uint32_t* ro = sc.rec_off + (size_t)W.r * in.cap_per_rpc; uint32_t* rl = sc.rec_len + (size_t)W.r * in.cap_per_rpc;
    const uint32_t wlen = W.rest < WP_WIN ? W.rest : WP_WIN;
    const uint32_t nb = (wlen + WP_T) / WP_T < WP_WIN / WP_T ? (wlen + WP_T) / WP_T : WP_WIN / WP_T;   // positions 0 .. wlen, in rows of WP_T
    wire_window_load(W, win);
    if (t < WP_WIN / 64) reach[t] = 0ull;
    if (t == 0) { s_more[0] = 0u; s_bad = 0u; }
    if (t == WP_T - 1u) {                                            // where the chain enters this window, and the records before it
        uint32_t pos = 0, cnt = 0, wc = 0;
        bool bad = false;
        while (wc < W.w) {
            const uint32_t o = pos - wc * WP_WIN;
            if (o >= WP_ENT) { bad = true; break; }
            const uint2 e = sc.went[(size_t)(blockIdx.x - W.w + wc) * WP_ENT + o];
            if (e.x == WP_NONE) { bad = true; break; }
            cnt += e.y; pos = e.x; wc = pos / WP_WIN;
        }
        s_entry = bad ? WP_NONE : wc > W.w ? WP_SKIP : pos - W.wbase;
        s_base = cnt;
    }
    wire_barrier();
    const uint32_t entry = s_entry, base = s_base;
    if (entry == WP_NONE) { if (last && t == 0) { sc.count[W.r] = 0u; sc.status[W.r] = WIRE_SERIAL; } return; }
    if (entry == WP_SKIP) return;                                    // a record longer than a window passes over this one (never the last)
    for (uint32_t b = 0; b < nb; ++b) {                              // where the chain goes from every position (terminals: themselves)
        const uint32_t o = b * WP_T + t;
        uint32_t hdr, L, nxt;
        const uint32_t k = wire_node(W, win, o, hdr, L, nxt);
        J[0][o] = (uint16_t)(k == N_REC_NEXT ? nxt : o);
    }
    if (t == 0) reach[entry >> 6] = 1ull << (entry & 63u);
    wire_barrier();
    uint32_t cur = 0;
    for (uint32_t round = 0;; ++round) {                             // pointer doubling: after round k the chain's first 2^(k+1) positions are marked
        if (t == 0) s_more[(round + 1) % 3] = 0u;
        bool more = false;
        for (uint32_t b0 = 0; b0 < nb; b0 += 4) {                    // (four rows at a time: the dependent reads of a row overlap the next rows')
            uint32_t j[4], j2[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) j[u] = b0 + u < nb ? J[cur][(b0 + u) * WP_T + t] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) j2[u] = J[cur][j[u]];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (b0 + u >= nb) continue;                          // (uniform)
                J[cur ^ 1][(b0 + u) * WP_T + t] = (uint16_t)j2[u];
                const unsigned long long mine = reach[(b0 + u) * (WP_T / 64) + wave];   // (the word of this wave's 64 positions: a broadcast read)
                if ((mine >> lane) & 1ull) {
                    if (!((reach[j[u] >> 6] >> (j[u] & 63u)) & 1ull)) { atomicOr(&reach[j[u] >> 6], 1ull << (j[u] & 63u)); more = true; }
                }
            }
        }
        if (more) s_more[round % 3] = 1u;
        wire_barrier();
        cur ^= 1;
        if (!s_more[round % 3]) break;                               // (uniform: read after the barrier; the flag is reused three rounds later)
    }
    // the visited positions, in order: records (numbered by a prefix over the bitmap's words), and the one terminal
    uint32_t mine_n = 0;
    const unsigned long long word = t < WP_WIN / 64 ? reach[t] : 0ull;
    for (unsigned long long m = word; m; m &= m - 1ull) {
        const uint32_t o = t * 64 + (uint32_t)__ffsll((unsigned long long)m) - 1u;
        uint32_t hdr, L, nxt;
        const uint32_t k = wire_node(W, win, o, hdr, L, nxt);
        if (k == N_BAD) s_bad = 1u;
        else if (k != N_END) mine_n++;
    }
    const uint32_t incl = wire_wave_incl_scan(mine_n);
    if (lane == 63) wsum[wave] = incl;
    wire_barrier();
    if (s_bad) { if (last && t == 0) { sc.count[W.r] = 0u; sc.status[W.r] = WIRE_SERIAL; } return; }   // (uniform; a window in front of the last: the last one sees it in went[])
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < WP_WIN / 64 / 64; ++w) { before += w < wave ? wsum[w] : 0u; total += wsum[w]; }   // (the bitmap's words live in the first two waves)
    uint32_t k = base + before + incl - mine_n;
    for (unsigned long long m = word; m; m &= m - 1ull) {
        const uint32_t o = t * 64 + (uint32_t)__ffsll((unsigned long long)m) - 1u;
        uint32_t hdr, L, nxt;
        const uint32_t kind = wire_node(W, win, o, hdr, L, nxt);
        if (kind == N_REC_NEXT || kind == N_REC_EXIT) {
            if (k < in.cap_per_rpc) { ro[k] = W.off + W.wbase + o + hdr; rl[k] = L; }
            ++k;
        }
    }
    if (last && t == 0) {
        const uint32_t count = base + total;
        sc.count[W.r] = count;
        sc.status[W.r] = ((in.max_per_rpc && count > in.max_per_rpc) || count > in.cap_per_rpc) ? WIRE_TOO_LARGE : WIRE_OK;   // gubernator.go:189-193
    }
}

// The numbering of the batch: first[r] = where RPC r's items start (RPCs that are not ok contribute nothing), first[nrpc] = the batch
// size; an RPC that would not fit the arrays any more is turned away as too large.  By one workgroup of WAVES waves; the loads of
// several rows of payloads are in flight together (a row at a time: 2 us of latency per row).
template <uint32_t WAVES>
__device__ __forceinline__ void wire_number_batch(const WireIn& in, const WireScratch& sc) {
    __shared__ uint32_t wtot[WAVES];
    constexpr uint32_t U = WAVES == 1 ? 16 : 2, ROW = 64 * WAVES;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < in.nrpc; base += ROW * U) {
        int32_t st[U]; uint32_t cn[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t r = base + u * ROW + t;
            st[u] = r < in.nrpc ? sc.status[r] : WIRE_MALFORMED;
            cn[u] = r < in.nrpc ? sc.count[r] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t r = base + u * ROW + t;
            const uint32_t c = st[u] == WIRE_OK ? cn[u] : 0u;        // (rows behind the last payload: zeros)
            const uint32_t incl = wire_wave_incl_scan(c);
            uint32_t before = 0, tot = (uint32_t)__shfl((int)incl, 63, 64);
            if (WAVES > 1) {
                if (lane == 63u) wtot[wave] = incl;
                wire_barrier();
                tot = 0;
                for (uint32_t w = 0; w < WAVES; ++w) { before += w < wave ? wtot[w] : 0u; tot += wtot[w]; }
                wire_barrier();
            }
            const uint32_t excl = carry + before + incl - c;
            if (r < in.nrpc) {
                if (c && excl + c > in.cap_items) sc.status[r] = WIRE_TOO_LARGE;   // (rare: flagged; its slots stay unused — see k_wire_fill)
                sc.first[r] = excl;
            }
            carry += tot;
        }
    }
    if (t == 0) sc.first[in.nrpc] = carry < in.cap_items ? carry : in.cap_items;
}
// The serial walk — only_flagged: behind k_wire_win_b, only the payloads it left (status WIRE_SERIAL, or shorter than WP_MIN: no windows);
// 0 = every payload — and, with `number`, the numbering of the batch by the LAST workgroup of the launch to finish (a launch of its own
// costs 5 us where the whole decode of 64 RPCs takes 60; every workgroup's ticket is a device-scope atomic on one word, 20-25 ns each
// one after the other — 100 us for 4 000 payloads —, so batches of more than WIRE_NUMBER_FUSED payloads get k_wire_prefix instead).
constexpr uint32_t WIRE_NUMBER_FUSED = 256;
__global__ __launch_bounds__(64) void k_wire_scan(WireIn in, WireScratch sc, uint32_t only_flagged, uint32_t number) {
    if (!only_flagged || in.rpc_len[blockIdx.x] < WP_MIN || sc.status[blockIdx.x] == WIRE_SERIAL) wire_scan_body(in, sc);
    if (!number) return;
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) {
        __threadfence();                                             // this payload's count and status before the ticket
        s_last = atomicAdd(sc.done, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    wire_barrier();
    if (!s_last) return;
    __threadfence();                                                 // (what the other workgroups wrote before their tickets is visible from here)
    wire_number_batch<1>(in, sc);
    if (threadIdx.x == 0) *sc.done = 0u;
}
__global__ __launch_bounds__(1024) void k_wire_prefix(WireIn in, WireScratch sc) { wire_number_batch<16>(in, sc); }

// One thread per item.  (Measured in round 5, profiles/r05_m_*: of its 16 us for 64 000 items 6 are parse_req, the rest is the chain of
// dependent first-touch loads first[] -> status / record -> bytes at ~1.5 us each; parsing from a copy of the records in LDS, the search
// of first[] in LDS and 8-byte row stores changed nothing and were taken out again.)
__global__ __launch_bounds__(256) void k_wire_fill(WireIn in, WireScratch sc, WireOut out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t total = sc.first[in.nrpc];
    if (i >= total) return;
    uint32_t lo = 0, hi = in.nrpc;                                   // the RPC whose slice holds item i: the last r with first[r] <= i that has items
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sc.first[mid] <= i) lo = mid; else hi = mid; }
    const uint32_t r = lo, k = i - sc.first[r];
    out.item_rpc[i] = r;
    const bool dead_rpc = sc.status[r] != WIRE_OK || k >= sc.count[r];
    wire::ReqFields f;
    bool ok = !dead_rpc;
    if (ok) {
        const uint8_t* p = in.buf + sc.rec_off[(size_t)r * in.cap_per_rpc + k];
        ok = wire::parse_req(p, p + sc.rec_len[(size_t)r * in.cap_per_rpc + k], f);
        if (!ok) atomicExch((int*)&sc.status[r], WIRE_MALFORMED);      // the whole message is rejected, as the runtimes do
    }
    uint8_t pre = 0;
    uint32_t klen = 0;
    if (!ok) pre = WIRE_PRE_DEAD;
    else if (f.unique_key.n == 0) pre = 1;                            // gubernator.go:208-212 "field 'unique_key' cannot be empty"
    else if (f.name.n == 0) pre = 2;                                  // gubernator.go:213-217 "field 'namespace' cannot be empty"
    else {
        const uint64_t kl = (uint64_t)f.name.n + 1 + f.unique_key.n;
        klen = kl > 0xffffffffull ? 0xffffffffu : (uint32_t)kl;
        if (klen + 8 <= out.key_stride) {                             // client.go:39-41 (a longer key is refused by the engine: max_key_bytes)
            uint8_t* row = out.key_rows + (size_t)i * out.key_stride;
            for (uint32_t b = 0; b < f.name.n; ++b) row[b] = f.name.p[b];
            row[f.name.n] = '_';
            for (uint32_t b = 0; b < f.unique_key.n; ++b) row[f.name.n + 1 + b] = f.unique_key.p[b];
            for (uint32_t b = klen; b < ((klen + 7u) & ~7u); ++b) row[b] = 0;
        }
    }
    out.key_len[i] = klen;
    out.pre_err[i] = pre;
    out.hits[i] = f.hits; out.limit[i] = f.limit; out.duration[i] = f.duration; out.burst[i] = f.burst;
    out.created_at[i] = f.created_at ? f.created_at : out.now_ms;      // gubernator.go:218-220
    out.algo_raw[i] = (int32_t)f.algorithm;
    out.algorithm[i] = (f.algorithm == 0 || f.algorithm == 1) ? (uint8_t)f.algorithm : 255;
    out.behavior[i] = (uint32_t)f.behavior;
    out.is_owner[i] = in.rpc_owner ? (in.rpc_owner[r] ? 1 : 0) : 1;
}

// items of an RPC that a later record showed to be malformed (or that did not fit): dead slots — an empty key never reaches a bucket
// ... and the decode's report: the verdicts per payload go to host memory from here (three copy commands less behind the decode; k_wire_fill
// may still have turned a payload away, so this is the first kernel that can), and the numbering's ticket counter is left at zero whatever
// happened before (a memset command per decode did that)
__global__ __launch_bounds__(256) void k_wire_kill(WireIn in, WireScratch sc, WireOut out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (out.rep_first) {
        for (uint32_t r = i; r <= in.nrpc; r += gridDim.x * 256u) {
            out.rep_first[r] = sc.first[r];
            if (r < in.nrpc) { out.rep_count[r] = sc.count[r]; out.rep_status[r] = sc.status[r]; }
        }
    }
    if (i == 0) *sc.done = 0u;
    if (i >= sc.first[in.nrpc]) return;
    if (sc.status[out.item_rpc[i]] != WIRE_OK) { out.key_len[i] = 0; out.pre_err[i] = WIRE_PRE_DEAD; }
}

// ---- the answers as GetRateLimitsResp bytes, encoded where they are -----------------------------------------------------------------
// gubernator.proto:184-203: GetRateLimitsResp = repeated RateLimitResp responses = 1; RateLimitResp = status 1, limit 2, remaining 3,
// reset_time 4 (varints, zero fields not written), error 5, metadata 6.  One workgroup per RPC turns its slice of the answers (arrival
// order, HBM) into the bytes the runtimes write for it — `0x0a len body` per item, at most 37 bytes — and leaves them in DEVICE-VISIBLE
// HOST memory at wire_enc_off(first, r), their number in enc_len[r]: all that is left for the RPC's caller thread is one memcpy.  An
// RPC with an item error (the text needs the item's key and raw algorithm: the host transcoder writes it) is not encoded: its raw answers
// go to the host arrays instead and enc_len[r] = WIRE_ENC_RAW.  Byte-for-byte the host transcoder's output (csrc/wire.cpp), which is
// checked against the protobuf runtime.
constexpr uint32_t WIRE_ENC_ITEM_MAX = 37;                           // 0x0a, one length byte, status (2), three int64 fields (11 each)
constexpr uint32_t WIRE_ENC_RAW = 0xffffffffu;
constexpr uint32_t WE_T = 256, WE_PER = 4, WE_CH = WE_T * WE_PER;  // a workgroup encodes its RPC in pieces of 1 024 items
// where RPC r's bytes start: 16-byte aligned, and the regions of two RPCs never meet (count x 37 + 17 bytes apart at least, the last
// piece is written in whole 16-byte words)
GW_HD size_t wire_enc_off(uint32_t first, uint32_t r) { return ((size_t)first * WIRE_ENC_ITEM_MAX + (size_t)r * 32u) & ~(size_t)15; }
GW_HD size_t wire_enc_bytes(uint32_t max_items, uint32_t max_rpcs) { return (size_t)max_items * WIRE_ENC_ITEM_MAX + (size_t)max_rpcs * 32u + 64u; }
struct WireEnc {
    uint32_t nrpc;
    const uint32_t *first, *count; const int32_t* status;            // the decode's verdicts (HBM)
    // the answers (HBM): item i's are at fwd[i] — the front's shares, as the engines left them: this kernel IS the answers' last hop
    // (k_fr_out's place, guber_front.h front_out) — or, fwd null, at i
    const uint32_t* fwd;
    const uint8_t *d_status, *d_err; const int64_t *d_limit, *d_remaining, *d_reset;
    uint8_t* enc; uint32_t* enc_len;                                 // host memory the device writes in place
    uint8_t *h_status, *h_err; int64_t *h_limit, *h_remaining, *h_reset;               // host: the raw answers of the RPCs that are not encoded
};
__device__ __forceinline__ uint32_t wire_varint_len(uint64_t v) {
    uint32_t n = 1;
    while (v >= 0x80ull) { v >>= 7; ++n; }
    return n;
}
__device__ __forceinline__ uint32_t wire_put_field(uint8_t* p, uint8_t tag, uint64_t v) {   // a varint field that is not zero
    uint32_t n = 0;
    p[n++] = tag;
    while (v >= 0x80ull) { p[n++] = (uint8_t)(v | 0x80ull); v >>= 7; }
    p[n++] = (uint8_t)v;
    return n;
}
__global__ __launch_bounds__(WE_T) void k_wire_enc(WireEnc E) {
    __shared__ alignas(16) uint8_t stage[WE_CH * WIRE_ENC_ITEM_MAX + 32];
    __shared__ uint32_t wsum[WE_T / 64];
    __shared__ uint32_t s_bad;
    const uint32_t r = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t first = E.first[r];
    const uint32_t count = E.status[r] == WIRE_OK ? E.count[r] : 0u;  // (a message turned away whole has no answers: its caller gets the verdict)
    if (count == 0u) { if (tid == 0) E.enc_len[r] = 0u; return; }
    if (tid == 0) s_bad = 0u;
    __syncthreads();
    // an item error anywhere in the RPC?  An RPC of one piece (the usual one: at most 1 000 items, gubernator.go:40) finds out from the loads it
    // encodes from; a longer one looks first
    const bool one_piece = count <= WE_CH;
    if (!one_piece) {
        uint32_t bad = 0;
        for (uint32_t k = tid; k < count; k += WE_T) { const uint32_t i = first + k; bad |= E.d_err[E.fwd ? E.fwd[i] : i] != 0 ? 1u : 0u; }
        if (bad) s_bad = 1u;
        __syncthreads();
    }
    auto raw = [&]() {                                                 // the host transcoder's: the raw answers, as they are, in arrival order
        for (uint32_t k = tid; k < count; k += WE_T) {
            const uint32_t i = first + k, j = E.fwd ? E.fwd[i] : i;
            E.h_status[i] = E.d_status[j]; E.h_err[i] = E.d_err[j]; E.h_limit[i] = E.d_limit[j]; E.h_remaining[i] = E.d_remaining[j]; E.h_reset[i] = E.d_reset[j];
        }
        if (tid == 0) E.enc_len[r] = WIRE_ENC_RAW;
    };
    if (!one_piece && s_bad) { raw(); return; }
    uint8_t* const dst = E.enc + wire_enc_off(first, r);
    uint32_t carry = 0, flushed = 0;                                   // bytes at the front of `stage` that wait for a whole word; bytes that have left
    for (uint32_t c0 = 0; c0 < count; c0 += WE_CH) {
        // thread t: items c0 + 4 t .. c0 + 4 t + 3 (neighbours in the output: one offset per thread)
        const uint32_t i0 = c0 + tid * WE_PER;
        uint32_t st[WE_PER]; uint64_t lim[WE_PER], rem[WE_PER], rst[WE_PER];
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < WE_PER; ++k) {
            st[k] = 0; lim[k] = 0; rem[k] = 0; rst[k] = 0;
            if (i0 + k < count) {
                const uint32_t i = E.fwd ? E.fwd[first + i0 + k] : first + i0 + k;
                st[k] = E.d_status[i]; lim[k] = (uint64_t)E.d_limit[i]; rem[k] = (uint64_t)E.d_remaining[i]; rst[k] = (uint64_t)E.d_reset[i];
                if (one_piece && E.d_err[i] != 0) s_bad = 1u;
                mine += 2u + (st[k] ? 1u + wire_varint_len(st[k]) : 0u) + (lim[k] ? 1u + wire_varint_len(lim[k]) : 0u) +
                        (rem[k] ? 1u + wire_varint_len(rem[k]) : 0u) + (rst[k] ? 1u + wire_varint_len(rst[k]) : 0u);
            }
        }
        const uint32_t incl = wire_wave_incl_scan(mine);
        if (lane == 63u) wsum[wave] = incl;
        __syncthreads();
        if (one_piece && s_bad) { raw(); return; }
        uint32_t base = carry + incl - mine, chunk = 0;
#pragma unroll
        for (uint32_t w = 0; w < WE_T / 64; ++w) { const uint32_t s = wsum[w]; if (w < wave) base += s; chunk += s; }
        uint8_t* o = stage + base;
#pragma unroll
        for (uint32_t k = 0; k < WE_PER; ++k) {
            if (i0 + k < count) {
                uint8_t* body = o + 2;
                uint32_t b = 0;
                if (st[k]) b += wire_put_field(body + b, 0x08, st[k]);
                if (lim[k]) b += wire_put_field(body + b, 0x10, lim[k]);
                if (rem[k]) b += wire_put_field(body + b, 0x18, rem[k]);
                if (rst[k]) b += wire_put_field(body + b, 0x20, rst[k]);
                o[0] = 0x0a; o[1] = (uint8_t)b;                        // (a body is at most 35 bytes: its length is one byte)
                o += 2 + b;
            }
        }
        const uint32_t total = carry + chunk;
        const bool last = c0 + WE_CH >= count;
        const uint32_t full = last ? (total + 15u) & ~15u : total & ~15u;
        if (last && tid < full - total) stage[total + tid] = 0;       // (the last word's padding: never read as part of the response)
        __syncthreads();
        for (uint32_t q = tid * 16u; q < full; q += WE_T * 16u) *(uint4*)(dst + flushed + q) = *(const uint4*)(stage + q);
        if (last) { if (tid == 0) E.enc_len[r] = flushed + total; return; }
        uint8_t keep = 0;
        if (tid < total - full) keep = stage[full + tid];
        __syncthreads();
        if (tid < total - full) stage[tid] = keep;
        carry = total - full; flushed += full;
        __syncthreads();
    }
}

}  // namespace guber
