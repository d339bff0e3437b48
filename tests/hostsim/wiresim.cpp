// wiresim.cpp — TEST INFRASTRUCTURE: the device wire decoder's chain walks (guber_kernels_wire.h: the serial k_wire_scan and the
// parallel k_wire_win_a / k_wire_win_b with the serial walk behind them) compiled for the host against fakehip/ and run against the SAME framing code over plain memory
// (scan_toplevel<MemReader>: what the host transcoder's fuzz runs) on generated and mutated payloads: item counts, verdicts and every
// record's offset and length must agree.  Built as its own library (tests/test_wire_scan_devsim.py); nothing in the product includes it.
#include "devsim.cpp"
#include "../../gubernator_amd/csrc/guber_kernels_wire.h"

namespace {
struct Rng { uint64_t s; uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; } uint32_t below(uint32_t n) { return (uint32_t)(next() % n); } };
void put_varint(std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
// one payload: records of field 1 with bodies of assorted lengths (1-, 2- and 3-byte length varints), now and then an unknown
// field of every wire type, a multi-byte tag, a tag with field number 0; then mutations (truncation, byte flips)
// (plain: nothing but records of the usual form — what the parallel walk finishes on its own, windows, window edges and all)
std::vector<uint8_t> make_payload(Rng& g) {
    std::vector<uint8_t> p;
    const uint32_t shape = g.below(8);
    const bool plain = g.below(2) == 0;
    const bool small = g.below(3) != 0;                                 // (plain payloads: few records of kilobytes — a chain enters a window within its first WP_ENT positions or the payload is the serial walk's)
    const uint32_t n = shape == 0 ? 0 : shape < 5 ? g.below(40) : shape < 7 ? 200 + g.below(1300) : g.below(6);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t k = plain ? 6 + g.below(58) : g.below(64);
        if (k == 0) { p.push_back(0x10); put_varint(p, g.next()); continue; }                       // unknown varint field 2
        if (k == 1) { p.push_back(0x1d); for (int b = 0; b < 4; ++b) p.push_back((uint8_t)g.next()); continue; }   // fixed32 field 3
        if (k == 2) { p.push_back(0x21); for (int b = 0; b < 8; ++b) p.push_back((uint8_t)g.next()); continue; }   // fixed64 field 4
        if (k == 3) { p.push_back(0x2a); const uint32_t L = g.below(20); put_varint(p, L); for (uint32_t b = 0; b < L; ++b) p.push_back((uint8_t)g.next()); continue; }   // unknown LEN field 5
        if (k == 4) { p.push_back(0x8a); p.push_back(0x01); const uint32_t L = g.below(10); put_varint(p, L); for (uint32_t b = 0; b < L; ++b) p.push_back(0x41); continue; }   // field 17, two-byte tag
        if (k == 5) { p.push_back(0x8a); p.push_back(0x80); p.push_back(0x00); const uint32_t L = g.below(5); put_varint(p, L); for (uint32_t b = 0; b < L; ++b) p.push_back(0x42); continue; }   // field 1, over-long tag
        uint32_t L = g.below(60);
        const uint32_t big = plain ? 1 + g.below(small ? 3999 : 199) : g.below(200);
        if (plain && g.below(small ? 20000 : 400) == 0) L = 9000 + g.below(7000);                                       // a record longer than a window (two-byte length)
        if (plain && g.below(40) == 0) L = 0;
        if (big == 0) L = 16384 + g.below(300);                                                        // three-byte length
        else if (big < 12) L = 128 + g.below(3000);                                                    // two-byte length
        p.push_back(0x0a);
        if (!plain && g.below(50) == 0) { p.push_back((uint8_t)(L & 0x7f) | 0x80); p.push_back((uint8_t)((L >> 7) & 0x7f) | 0x80); p.push_back((uint8_t)(L >> 14)); }   // non-minimal varint
        else put_varint(p, L);
        for (uint32_t b = 0; b < L; ++b) p.push_back((uint8_t)(g.below(9) ? 0x0a * (g.below(3) == 0) + g.below(200) : 0x0a));   // bodies full of bytes that look like tags
    }
    const uint32_t mut = plain ? (g.below(4) == 0 ? g.below(10) : 9) : g.below(10);
    if (mut == 0 && !p.empty()) p.resize(g.below((uint32_t)p.size()));
    if (mut == 1) for (int f = 0; f < 3 && !p.empty(); ++f) p[g.below((uint32_t)p.size())] = (uint8_t)g.next();
    if (mut == 2 && !p.empty()) p[p.size() - 1 - g.below(std::min<uint32_t>(4, (uint32_t)p.size()))] = 0x0a;
    return p;
}
}  // namespace

extern "C" {
// runs `iters` rounds of 1 .. 6 payloads through k_wire_scan (mode 0) or k_wire_win_a + k_wire_win_b + k_wire_scan for what they leave (mode 1);
// returns the number of payloads that disagree with scan_toplevel<MemReader>; stats: payloads, records, payloads with a verdict other
// than ok, payloads the parallel walk finished on its own, those of them with more than one window
uint64_t ws_fuzz(uint32_t iters, uint64_t seed, int table, uint32_t max_per_rpc, unsigned long long* stats) {
    Rng g{seed * 0x9E3779B97F4A7C15ull + 12345};
    uint64_t bad = 0;
    stats[0] = stats[1] = stats[2] = stats[3] = stats[4] = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t nrpc = 1 + g.below(6), cap = 1600;
        std::vector<std::vector<uint8_t>> pl(nrpc);
        std::vector<uint8_t> buf; std::vector<uint32_t> off(nrpc), len(nrpc);
        for (uint32_t r = 0; r < nrpc; ++r) {
            pl[r] = make_payload(g);
            while (buf.size() & 15) buf.push_back(0xAA);                 // (padding between payloads is arbitrary bytes)
            off[r] = (uint32_t)buf.size(); len[r] = (uint32_t)pl[r].size();
            buf.insert(buf.end(), pl[r].begin(), pl[r].end());
        }
        for (int k = 0; k < 32; ++k) buf.push_back(k < 16 ? 0 : 0x0a);      // 16 readable bytes past the end (+ slack)
        std::vector<uint32_t> ro((size_t)nrpc * cap, 0xdeadbeef), rl((size_t)nrpc * cap, 0xdeadbeef), cnt(nrpc, 0xdeadbeef), first(nrpc + 1, 0);
        std::vector<int32_t> st(nrpc, 12345);
        guber::WireIn in{}; in.buf = buf.data(); in.rpc_off = off.data(); in.rpc_len = len.data(); in.rpc_owner = nullptr;
        in.nrpc = nrpc; in.cap_per_rpc = cap; in.max_per_rpc = max_per_rpc; in.cap_items = nrpc * cap;
        std::vector<uint32_t> wfirst(nrpc + 1, 0);
        for (uint32_t r = 0; r < nrpc; ++r) wfirst[r + 1] = wfirst[r] + guber::wire_windows_of(len[r]);
        const uint32_t windows = wfirst[nrpc];
        std::vector<uint2> went((size_t)windows * guber::WP_ENT, make_uint2(0xabababab, 0xabababab));
        uint32_t done = 0;
        guber::WireScratch sc{ro.data(), rl.data(), cnt.data(), st.data(), first.data(), wfirst.data(), went.data(), &done};
        const bool fused = (it & 1u) == 0;                               // the numbering: by k_wire_scan's last workgroup / by k_wire_prefix
        if (table) {
            bool multi = false;
            for (uint32_t r = 0; r < nrpc; ++r) multi = multi || wfirst[r + 1] - wfirst[r] > 1;
            if (multi) fakehip::launch(dim3(windows), dim3(guber::WP_T), nullptr, [&] { guber::k_wire_win_a(in, sc); });
            if (windows) fakehip::launch(dim3(windows), dim3(guber::WP_T), nullptr, [&] { guber::k_wire_win_b(in, sc); });
            for (uint32_t r = 0; r < nrpc; ++r) { stats[3] += wfirst[r + 1] > wfirst[r] && st[r] != guber::WIRE_SERIAL; stats[4] += wfirst[r + 1] - wfirst[r] > 1 && st[r] != guber::WIRE_SERIAL; }
            fakehip::launch(dim3(nrpc), dim3(64), nullptr, [&] { guber::k_wire_scan(in, sc, 1u, fused ? 1u : 0u); });
        } else fakehip::launch(dim3(nrpc), dim3(64), nullptr, [&] { guber::k_wire_scan(in, sc, 0u, fused ? 1u : 0u); });
        if (!fused) fakehip::launch(dim3(1), dim3(1024), nullptr, [&] { guber::k_wire_prefix(in, sc); });
        // the numbering of the batch by the launch's last workgroup: the exclusive scan of the counts of the payloads that are ok
        {
            uint32_t run = 0;
            bool okp = done == 0;
            for (uint32_t r = 0; r < nrpc; ++r) { okp = okp && first[r] == run; if (st[r] == guber::WIRE_OK) run += cnt[r]; }
            okp = okp && first[nrpc] == run;
            if (!okp) { if (bad < 5) fprintf(stderr, "[wiresim] seed %llu it %u: the batch's numbering is off\n", (unsigned long long)seed, it); bad++; }
        }
        for (uint32_t r = 0; r < nrpc; ++r) {
            std::vector<uint32_t> wo, wl;
            guber::MemReader rd{pl[r].data(), len[r]};
            uint32_t wc = 0;
            int32_t ws = guber::scan_toplevel(rd, len[r], cap, wc, [&](uint32_t, uint32_t bo, uint32_t bl) { wo.push_back(off[r] + bo); wl.push_back(bl); });
            if (ws == guber::WIRE_OK && ((max_per_rpc && wc > max_per_rpc) || wc > cap)) ws = guber::WIRE_TOO_LARGE;
            bool ok = st[r] == ws;
            if (ws != guber::WIRE_MALFORMED) ok = ok && cnt[r] == wc;       // (a malformed payload's count is how far the walk got: not part of the contract)
            if (ok && ws == guber::WIRE_OK)
                for (uint32_t k = 0; k < wc && k < cap; ++k) if (ro[(size_t)r * cap + k] != wo[k] || rl[(size_t)r * cap + k] != wl[k]) { ok = false; break; }
            if (!ok) {
                if (bad < 5) fprintf(stderr, "[wiresim] seed %llu it %u rpc %u (len %u): status %d want %d, count %u want %u\n", (unsigned long long)seed, it, r, len[r], st[r], ws, cnt[r], wc);
                bad++;
            }
            stats[0]++; stats[1] += wc; stats[2] += ws != guber::WIRE_OK;
        }
    }
    return bad;
}
}
