// guber_host.cpp — host-only parts of the C ABI (include/guber_gpu.h):
//   * ReplicatedConsistentHash as a sorted point table (reference replicated_hash.go:29-119): which
//     GPU / peer owns a key.  Built once per membership change, queried on the host here and by the
//     k_route kernel on device arrays.
//   * GregorianExpiration / GregorianDuration (reference interval.go:84-148) in UTC, used by the host
//     layer to fill guber_batch_t.greg_expire / greg_duration before a batch is shipped.
//   * error strings identical to the reference's.
#include "guber_host.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <vector>

#include "guber_algo.h"

namespace {

// ---- MD5 (RFC 1321); Go's crypto/md5 labels the virtual nodes (replicated_hash.go:81) ------------
struct Md5 {
    uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    static uint32_t rol(uint32_t x, int c) { return (x << c) | (x >> (32 - c)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
            0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
            0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
            0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
            0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
            0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
            0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
            0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int R[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
        uint32_t m[16];
        memcpy(m, p, 64);
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
        for (int i = 0; i < 64; ++i) {
            const int round = i >> 4;
            uint32_t f; int g;
            switch (round) {
            case 0: f = (b & c) | (~b & d); g = i; break;
            case 1: f = (d & b) | (~d & c); g = (5 * i + 1) & 15; break;
            case 2: f = b ^ c ^ d; g = (3 * i + 5) & 15; break;
            default: f = c ^ (b | ~d); g = (7 * i) & 15; break;
            }
            const uint32_t t = d;
            d = c; c = b;
            b = b + rol(a + f + K[i] + m[g], R[round][i & 3]);
            a = t;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d;
    }
    static std::string hex(const std::string& s) {
        Md5 h;
        std::vector<uint8_t> buf(s.begin(), s.end());
        const uint64_t bits = (uint64_t)buf.size() * 8;
        buf.push_back(0x80);
        while (buf.size() % 64 != 56) buf.push_back(0);
        for (int i = 0; i < 8; ++i) buf.push_back((uint8_t)(bits >> (8 * i)));
        for (size_t o = 0; o < buf.size(); o += 64) h.block(buf.data() + o);
        char out[33];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) snprintf(out + 8 * i + 2 * j, 3, "%02x", (h.st[i] >> (8 * j)) & 0xff);
        return std::string(out, 32);
    }
};

}  // namespace

struct guber_ring {
    std::vector<uint64_t> hash;
    std::vector<uint32_t> owner;
    int kind = 0;
    uint64_t id = 0;   // unique per created ring: engines cache the device image by id, not by (reusable) address
};

extern "C" int guber_ring_create(const char* const* peer_names, uint32_t n_peers, uint32_t replicas, int hash_kind,
                                 guber_ring_t** out) {
    if (!out || (!peer_names && n_peers) || (hash_kind != 0 && hash_kind != 1)) return GUBER_E_INVALID_ARG;
    if (replicas == 0) replicas = 512;   // replicated_hash.go:29 defaultReplicas
    struct Pt { uint64_t h; uint32_t o; uint32_t seq; };
    std::vector<Pt> pts;
    pts.reserve((size_t)n_peers * replicas);
    for (uint32_t p = 0; p < n_peers; ++p) {
        const std::string label = Md5::hex(peer_names[p]);             // :81
        for (uint32_t i = 0; i < replicas; ++i) {                      // :82-88
            const std::string s = std::to_string(i) + label;
            const uint64_t h = hash_kind == 1 ? guber::fnv1a_64((const uint8_t*)s.data(), (uint32_t)s.size())
                                              : guber::fnv1_64((const uint8_t*)s.data(), (uint32_t)s.size());
            pts.push_back({h, p, (uint32_t)pts.size()});
        }
    }
    std::sort(pts.begin(), pts.end(), [](const Pt& a, const Pt& b) { return a.h != b.h ? a.h < b.h : a.seq < b.seq; });  // :90
    static std::atomic<uint64_t> next_ring_id{1};
    guber_ring* r = new guber_ring();
    r->id = next_ring_id.fetch_add(1);
    r->kind = hash_kind;
    for (const Pt& p : pts) { r->hash.push_back(p.h); r->owner.push_back(p.o); }
    *out = r;
    return GUBER_OK;
}
extern "C" void guber_ring_destroy(guber_ring_t* r) { delete r; }
extern "C" int guber_ring_kind(const guber_ring_t* r) { return r ? r->kind : 0; }
extern "C" uint64_t guber_ring_id(const guber_ring_t* r) { return r ? r->id : 0; }
extern "C" uint32_t guber_ring_points(const guber_ring_t* r, uint64_t* hashes, uint32_t* owners, uint32_t cap) {
    if (!r) return 0;
    const uint32_t n = (uint32_t)r->hash.size();
    for (uint32_t i = 0; i < n && i < cap; ++i) { if (hashes) hashes[i] = r->hash[i]; if (owners) owners[i] = r->owner[i]; }
    return n;
}
// replicated_hash.go:104-119 Get
extern "C" int guber_ring_route(const guber_ring_t* r, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t n,
                                uint32_t* owner) {
    if (!r || r->hash.empty()) return GUBER_E_INVALID_ARG;   // "unable to pick a peer; pool is empty"
    if (n && (!key_bytes || !key_off || !owner)) return GUBER_E_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        const uint8_t* k = key_bytes + key_off[i];
        const uint32_t len = key_off[i + 1] - key_off[i];
        const uint64_t h = r->kind == 1 ? guber::fnv1a_64(k, len) : guber::fnv1_64(k, len);
        size_t idx = std::lower_bound(r->hash.begin(), r->hash.end(), h) - r->hash.begin();
        if (idx == r->hash.size()) idx = 0;
        owner[i] = r->owner[idx];
    }
    return GUBER_OK;
}

// interval.go:84-148 (the calendar arithmetic lives in guber_algo.h: the kernels evaluate the same code per request)
static guber::TzTable g_host_tz{};                                   // the process's zone (guber_set_timezone); all zero = UTC
static const guber::TzTable* host_tz() { return (g_host_tz.n || g_host_tz.offset0_s) ? &g_host_tz : nullptr; }
const guber::TzTable* guber_host_tz_table() { return &g_host_tz; }
// validate a zone and build its table; nothing is published (guber_set_timezone publishes to the devices first, to the host last)
int guber_host_build_tz(const guber_tz_t* tz, guber::TzTable* out) {
    guber::TzTable t{};
    if (tz) {
        if (tz->n > (uint32_t)guber::TZ_MAX || (tz->n && (!tz->when_s || !tz->offset_s))) return GUBER_E_INVALID_ARG;
        t.n = tz->n; t.offset0_s = tz->offset0_s;
        for (uint32_t k = 0; k < tz->n; ++k) {
            if (k && tz->when_s[k] <= tz->when_s[k - 1]) return GUBER_E_INVALID_ARG;        // ascending
            t.when_s[k] = tz->when_s[k]; t.offset_s[k] = tz->offset_s[k];
        }
    }
    *out = t;
    return GUBER_OK;
}
void guber_host_publish_tz(const guber::TzTable& t) { g_host_tz = t; }
int guber_host_set_tz(const guber_tz_t* tz) {
    guber::TzTable t{};
    const int rc = guber_host_build_tz(tz, &t);
    if (rc == GUBER_OK) g_host_tz = t;
    return rc;
}
extern "C" int guber_gregorian_expiration(int64_t now_ns, int64_t d, int64_t* expire_ms) {
    if (!expire_ms) return GUBER_E_INVALID_ARG;
    return -(int)guber::greg_expiration(now_ns, d, *expire_ms, host_tz());
}
extern "C" int guber_gregorian_duration(int64_t now_ns, int64_t d, int64_t* duration) {
    if (!duration) return GUBER_E_INVALID_ARG;
    return -(int)guber::greg_duration(now_ns, d, *duration, host_tz());
}

extern "C" uint64_t guber_xxhash64(const uint8_t* p, size_t len, uint64_t seed) { return guber::xxhash64(p, (uint32_t)len, seed); }
extern "C" uint64_t guber_fnv1_64(const uint8_t* p, size_t len) { return guber::fnv1_64(p, (uint32_t)len); }
extern "C" uint64_t guber_fnv1a_64(const uint8_t* p, size_t len) { return guber::fnv1a_64(p, (uint32_t)len); }

extern "C" const char* guber_strerror(int code) {
    switch (code) {
    case GUBER_OK: return "ok";
    case GUBER_E_INVALID_ARG: return "invalid argument";
    case GUBER_E_NO_DEVICE: return "no HIP device (the engine has no CPU fallback)";
    case GUBER_E_HIP: return "HIP runtime error";
    case GUBER_E_BATCH_TOO_LARGE: return "batch too large";
    case GUBER_E_TABLE_FULL: return "bucket table full";
    case GUBER_E_NOMEM: return "out of memory / buffer too small";
    case GUBER_E_KEY_TOO_LONG: return "key too long";
    case GUBER_E_NOT_FOUND: return "not found";
    case -20: return "malformed protobuf payload";                                     // GUBER_E_WIRE_MALFORMED
    case -21: return "Requests.RateLimits list too large; max size is '1000'";         // GUBER_E_WIRE_TOO_LARGE, gubernator.go:191
    case -22: return "wire batch full";                                                // GUBER_E_WIRE_FULL
    }
    return "unknown error";
}
// The reference's per-item error texts (workers.go:318, interval.go:93,107,136,147, gubernator.go:208-217).
extern "C" const char* guber_item_strerror(uint8_t e) {
    switch (e) {
    case GUBER_ITEM_OK: return "";
    case GUBER_ITEM_E_INVALID_ALGORITHM: return "Invalid rate limit algorithm '%d'";
    case GUBER_ITEM_E_GREGORIAN_WEEKS: return "`Duration = GregorianWeeks` not yet supported; consider making a PR!`";
    case GUBER_ITEM_E_GREGORIAN_INVALID: return "behavior DURATION_IS_GREGORIAN is set; but `Duration` is not a valid gregorian interval";
    case GUBER_ITEM_E_EMPTY_KEY: return "field 'unique_key' cannot be empty";
    case GUBER_ITEM_E_RETRY: return "internal: retry";
    case GUBER_ITEM_E_TABLE_FULL: return "rate limit table full";
    case GUBER_ITEM_E_KEY_TOO_LONG: return "rate limit key too long";
    }
    return "unknown item error";
}
extern "C" const char* guber_version(void) { return "gubernator-amd 0.1 (gfx950)"; }
