// wire_fuzz_asan.cpp — memory-safety fuzz of the wire front end (the only code of the library that parses bytes from the
// network): mutated GetRateLimitsReq / UpdatePeerGlobalsReq payloads through decode + encode under AddressSanitizer and
// UBSan.  Semantics are checked elsewhere (tests/test_wire_cpu.py against the protobuf runtime); this looks for out-of-bounds
// reads / writes and undefined behaviour only.
// Build + run:  g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -I include \
//                   tools/wire_fuzz_asan.cpp gubernator_amd/csrc/wire.cpp -o /tmp/wire_fuzz && /tmp/wire_fuzz 2000000
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/guber_wire.h"
// the device decoder's framing logic (scan_toplevel: a host/device template) over a bounds-checked host byte source, and the shared
// record parser it hands the bodies to: the SAME source the kernels k_wire_scan / k_wire_fill compile
#define GUBER_FAKEHIP 1
#include "../tests/hostsim/fakehip/hip/hip_runtime.h"
#include "../gubernator_amd/csrc/guber_kernels_wire.h"

// the few non-wire symbols wire.cpp links against, stubbed (no device here)
extern "C" void* guber_alloc_pinned(size_t) { return nullptr; }
extern "C" void guber_free_pinned(void*) {}
extern "C" int guber_gregorian_expiration(int64_t now_ns, int64_t d, int64_t* out) { *out = now_ns / 1000000 + 1000; return d > 5 ? -3 : 0; }
extern "C" int guber_gregorian_duration(int64_t, int64_t d, int64_t* out) { *out = 60000; return d > 5 ? -3 : 0; }
extern "C" const char* guber_item_strerror(uint8_t e) { return e == 1 ? "Invalid rate limit algorithm '%d'" : "some item error"; }
extern "C" int guber_eval_batch(guber_engine_t*, const guber_batch_t*, guber_result_t*) { return 0; }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static void put_varint(std::string& s, uint64_t v) { while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; } s.push_back((char)v); }

static std::string make_requests(int n) {
    std::string payload;
    for (int i = 0; i < n; ++i) {
        std::string r;
        std::string name = (rnd() % 9) ? "ns_" + std::to_string(rnd() % 5) : "";
        std::string uk = (rnd() % 9) ? "acct:\xc3\xa9" + std::to_string(rnd() % 1000) : "";
        r.push_back(0x0a); put_varint(r, name.size()); r += name;
        r.push_back(0x12); put_varint(r, uk.size()); r += uk;
        r.push_back(0x18); put_varint(r, rnd() % 3 ? 1 : rnd());
        r.push_back(0x20); put_varint(r, rnd() % 200);
        r.push_back(0x28); put_varint(r, rnd() % 4 ? 60000 : rnd() % 8);
        if (rnd() % 3 == 0) { r.push_back(0x30); put_varint(r, rnd() % 4); }
        if (rnd() % 3 == 0) { r.push_back(0x38); put_varint(r, rnd() % 64); }
        if (rnd() % 5 == 0) { std::string e = "\x0a\x01k\x12\x01v"; r.push_back(0x4a); put_varint(r, e.size()); r += e; }
        if (rnd() % 4 == 0) { r.push_back(0x50); put_varint(r, rnd()); }
        payload.push_back(0x0a); put_varint(payload, r.size()); payload += r;
    }
    return payload;
}
static std::string make_globals(int n) {
    std::string payload;
    for (int i = 0; i < n; ++i) {
        std::string st;
        st.push_back(0x08); put_varint(st, rnd() % 2); st.push_back(0x10); put_varint(st, rnd() % 100); st.push_back(0x18); put_varint(st, rnd());
        st.push_back(0x20); put_varint(st, 1700000000000ull + rnd() % 100000);
        std::string g, key = "glob_" + std::to_string(rnd() % 100);
        g.push_back(0x0a); put_varint(g, key.size()); g += key;
        g.push_back(0x12); put_varint(g, st.size()); g += st;
        g.push_back(0x18); put_varint(g, rnd() % 3); g.push_back(0x20); put_varint(g, rnd() % 100000); g.push_back(0x28); put_varint(g, rnd());
        payload.push_back(0x0a); put_varint(payload, g.size()); payload += g;
    }
    return payload;
}
static void mutate(std::string& p) {
    const int ops = rnd() % 4;
    for (int k = 0; k < ops && !p.empty(); ++k) {
        const size_t i = rnd() % p.size();
        switch (rnd() % 5) {
        case 0: p[i] ^= (char)(1u << (rnd() % 8)); break;
        case 1: p.insert(p.begin() + i, (char)rnd()); break;
        case 2: p.erase(p.begin() + i); break;
        case 3: p.resize(i); break;
        default: { std::string junk; for (int j = 0, m = 1 + rnd() % 8; j < m; ++j) junk.push_back((char)rnd()); p.insert(i, junk); }
        }
    }
}

namespace fakehip { State S; void yield() {} void barrier() {} unsigned long long wave_exchange(unsigned long long, int, unsigned long long*, unsigned long long*) { return 0; } }
// the device decoder's verdict on one payload: GUBER_OK / GUBER_E_WIRE_MALFORMED and the item count
static int device_logic(const std::vector<uint8_t>& m, uint32_t& count) {
    guber::MemReader rd{m.data(), (uint32_t)m.size()};
    std::vector<std::pair<uint32_t, uint32_t>> recs;
    int32_t st = guber::scan_toplevel(rd, rd.len, 1u << 20, count, [&](uint32_t, uint32_t bo, uint32_t bl) { recs.push_back({bo, bl}); });
    if (st != guber::WIRE_OK) return st;
    for (auto& r : recs) {
        guber::wire::ReqFields f;
        if (!guber::wire::parse_req(m.data() + r.first, m.data() + r.first + r.second, f)) return guber::WIRE_MALFORMED;
    }
    return GUBER_OK;
}

int main(int argc, char** argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 200000;
    guber_wire_batch_t* b = nullptr; guber_wire_items_t* w = nullptr;
    if (guber_wire_batch_create(64, 600, 0, &b) || guber_wire_items_create(24, 200, &w)) return 2;   // tight capacities: FULL paths get exercised
    std::vector<std::string> reqs, globs;
    for (int i = 0; i < 32; ++i) { reqs.push_back(make_requests(1 + rnd() % 12)); globs.push_back(make_globals(1 + rnd() % 10)); }
    long ok = 0, bad = 0, full = 0;
    std::vector<uint8_t> out;
    for (long it = 0; it < iters; ++it) {
        std::string p = reqs[rnd() % reqs.size()];
        mutate(p);
        // exact-size heap copy: any read past the payload is an ASan error
        std::vector<uint8_t> exact(p.begin(), p.end());
        if (rnd() % 7 == 0) guber_wire_batch_reset(b, 1700000000000ll);
        uint32_t first = 0, count = 0;
        const int rc = guber_wire_decode_requests(b, exact.data(), exact.size(), rnd() % 4 ? 0 : 8, rnd() & 1, &first, &count);
        {   // the device decoder's logic must reach the same verdict (capacity aside) and the same item count
            uint32_t dcount = 0;
            const int drc = device_logic(exact, dcount);
            const bool host_ok = rc == GUBER_OK || rc == GUBER_E_WIRE_FULL || rc == GUBER_E_WIRE_TOO_LARGE;
            if ((drc == GUBER_OK) != host_ok || (rc == GUBER_OK && dcount != count)) { printf("device framing disagrees with the host transcoder: host rc %d count %u, device rc %d count %u\n", rc, count, drc, dcount); return 1; }
        }
        if (rc == GUBER_OK) {
            ++ok;
            guber_result_t* r = guber_wire_batch_result(b);
            for (uint32_t i = first; i < first + count; ++i) { r->status[i] = rnd() & 1; r->limit[i] = (int64_t)rnd(); r->remaining[i] = (int64_t)rnd(); r->reset_time[i] = (int64_t)rnd(); r->err[i] = rnd() % 6 == 0 ? 1 + rnd() % 3 : 0; }
            out.resize(guber_wire_encode_bound(b, first, count));
            size_t n = 0;
            if (guber_wire_encode_responses(b, first, count, rnd() & 1, out.data(), out.size(), &n) != GUBER_OK || n > out.size()) { printf("encode failed within its own bound\n"); return 1; }
            std::vector<uint8_t> tight(n ? n - 1 : 0);
            size_t need = 0;
            if (n && guber_wire_encode_responses(b, first, count, 0, tight.data(), tight.size(), &need) != GUBER_E_NOMEM && need > tight.size()) { printf("short buffer not reported\n"); return 1; }
            (void)guber_wire_batch_view(b);
        } else if (rc == GUBER_E_WIRE_FULL) { ++full; guber_wire_batch_reset(b, 1700000000000ll); }
        else ++bad;
        std::string q = globs[rnd() % globs.size()];
        mutate(q);
        std::vector<uint8_t> exact2(q.begin(), q.end());
        const guber_item_t* items = nullptr; uint32_t cnt = 0;
        const int rc2 = guber_wire_decode_globals(w, exact2.data(), exact2.size(), 1700000000000ll, &items, &cnt);
        if (rc2 == GUBER_OK) for (uint32_t i = 0; i < cnt; ++i) { volatile uint8_t sink = items[i].key_len ? items[i].key[items[i].key_len - 1] : 0; (void)sink; }
    }
    printf("wire fuzz: %ld iterations, %ld decoded, %ld rejected, %ld full — no sanitizer report\n", iters, ok, bad, full);
    guber_wire_batch_destroy(b); guber_wire_items_destroy(w);
    return 0;
}
