// hostsim.cpp — TEST-ONLY host build of the kernel logic in gubernator_amd/csrc/guber_algo.h.
// It evaluates a batch the way the HIP pipeline does (group a batch's requests by key keeping request
// order; uniform runs answered per rank through eval_uniform_rank(); heterogeneous segments walked
// serially) so that apply()/skip() can be differential-tested against the oracle on a machine with
// no GPU.  Nothing in the product path links this file.
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../gubernator_amd/csrc/guber_algo.h"
#include "../../include/guber_gpu.h"

using namespace guber;

struct HostSim {
    std::unordered_map<std::string, Rec> table;
    uint64_t over = 0, hits = 0, misses = 0;
    int64_t size = 0;
    uint64_t par_segs = 0, ser_segs = 0;
};

static Req load_req(const guber_batch_t* b, uint32_t i) {
    Req r;
    r.hits = b->hits[i]; r.limit = b->limit[i]; r.duration = b->duration[i];
    r.burst = b->burst ? b->burst[i] : 0;
    r.created_at = b->created_at ? b->created_at[i] : b->now_ms;
    r.greg_expire = b->greg_expire ? b->greg_expire[i] : 0;
    r.greg_duration = b->greg_duration ? b->greg_duration[i] : 0;
    r.behavior = b->behavior ? b->behavior[i] : 0;
    r.algorithm = b->algorithm ? b->algorithm[i] : 0;
    r.is_owner = b->is_owner ? b->is_owner[i] : 1;
    // as load_req() of the kernels: calendar values from the batch clock when the host did not precompute them
    if ((r.behavior & BH_GREGORIAN) && !(b->greg_expire && b->greg_duration)) greg_fill(b->now_ms, r.duration, r.greg_expire, r.greg_duration);
    return r;
}
static void store(guber_result_t* res, uint32_t i, const Resp& rl) {
    res->status[i] = rl.status; res->limit[i] = rl.limit; res->remaining[i] = rl.remaining;
    res->reset_time[i] = rl.reset_time; res->err[i] = rl.err;
}

extern "C" {
void* hs_create() { return new HostSim(); }
// multi-request segments evaluated on the parallel (closed-form) path / walked serially
void hs_path_counts(void* h, uint64_t* out) { out[0] = ((HostSim*)h)->par_segs; out[1] = ((HostSim*)h)->ser_segs; }
void hs_destroy(void* h) { delete (HostSim*)h; }

// mode 0: pipeline emulation (uniform -> per-rank closed form, else serial); mode 1: force serial
int hs_eval_batch(void* hp, const guber_batch_t* b, guber_result_t* res, int mode) {
    HostSim* h = (HostSim*)hp;
    std::unordered_map<std::string, std::vector<uint32_t>> segs;
    std::vector<std::string> order;
    for (uint32_t i = 0; i < b->n; i++) {
        std::string k((const char*)b->key_bytes + b->key_off[i], b->key_off[i + 1] - b->key_off[i]);
        auto it = segs.find(k);
        if (it == segs.end()) { order.push_back(k); segs[k] = {i}; } else it->second.push_back(i);
    }
    uint64_t over0 = h->over, hit0 = h->hits, miss0 = h->misses;
    for (auto& k : order) {
        auto& idx = segs[k];
        Rec s0; rec_clear(s0);
        auto it = h->table.find(k);
        if (it != h->table.end()) s0 = it->second;
        Req r0 = load_req(b, idx[0]);
        bool uniform = true, created_only = true;
        for (size_t j = 1; j < idx.size(); j++) {
            Req rj = load_req(b, idx[j]);
            if (!req_eq(rj, r0)) { uniform = false; if (!req_eq_but_created(rj, r0)) { created_only = false; break; } }
        }
        if (!uniform && created_only) {
            if (created_at_irrelevant(s0, r0, b->now_ms)) uniform = true;          // live token bucket
            else if (r0.algorithm == ALGO_LEAKY && !(r0.behavior & BH_GLOBAL)) {   // live leaky bucket, nobody leaks
                bool ok = true;
                for (size_t j = 0; j < idx.size() && ok; j++) ok = leaky_created_harmless(s0, load_req(b, idx[j]), b->now_ms);
                uniform = ok;
            }
        }
        Rec fin = s0;
        if (idx.size() > 1) { if (uniform && mode == 0) h->par_segs++; else h->ser_segs++; }
        if (uniform && mode == 0) {
            for (size_t j = 0; j < idx.size(); j++) {
                Req r = load_req(b, idx[j]);
                Resp out; Rec after;
                uint32_t ev = eval_rank(s0, r, b->now_ms, j, out, after, eval_uniform_rank_1x);   // closed forms first, as the kernels do
                store(res, idx[j], out);
                h->over += (ev & EV_OVER) ? 1 : 0; h->hits += (ev & EV_HIT) ? 1 : 0; h->misses += (ev & EV_MISS) ? 1 : 0;
                if (j + 1 == idx.size()) fin = after;
            }
        } else {
            for (size_t j = 0; j < idx.size(); j++) {
                Req r = load_req(b, idx[j]);
                Resp out;
                uint32_t ev = apply(fin, r, b->now_ms, out);
                store(res, idx[j], out);
                h->over += (ev & EV_OVER) ? 1 : 0; h->hits += (ev & EV_HIT) ? 1 : 0; h->misses += (ev & EV_MISS) ? 1 : 0;
            }
        }
        h->size += (rec_kind(fin) != K_ABSENT) - (rec_kind(s0) != K_ABSENT);
        h->table[k] = fin;
    }
    res->over_limit_count = h->over - over0; res->cache_hits = h->hits - hit0; res->cache_misses = h->misses - miss0;
    res->unexpired_evictions = 0; res->cache_size = h->size;
    return 0;
}

// Differential fuzz of the closed forms (token_fast / leaky_fast) against apply() / skip(): random live buckets and
// requests in the regimes the closed forms accept, every rank of a run.  out[0] = comparisons made on the token closed
// form, out[1] = on the leaky one, out[2] = mismatches.
static uint64_t fz_next(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
void hs_fuzz_closed_forms(uint64_t seed, uint32_t iters, uint64_t* out) {
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    out[0] = out[1] = out[2] = 0;
    static const int64_t kHits[] = {1, 1, 1, 2, 3, 5, 7, 10, 100, 1000, (int64_t)1 << 40, (int64_t)1 << 62};
    static const int64_t kLim[] = {1, 2, 5, 10, 10, 100, 100, 2000, 65536, (int64_t)1 << 40, (int64_t)1 << 62, 0, -5};
    static const int64_t kDur[] = {0, 1, 5, 50, 1000, 30000, 60000, (int64_t)1 << 40};
    for (uint32_t it = 0; it < iters; ++it) {
        const int64_t now = 1700000000000ll + (int64_t)(fz_next(st) % 100000);
        Req r; memset(&r, 0, sizeof r);
        r.hits = kHits[fz_next(st) % 12]; r.limit = kLim[fz_next(st) % 13]; r.duration = kDur[fz_next(st) % 8];
        r.created_at = now + (int64_t)(fz_next(st) % 200) - 100;
        r.behavior = (fz_next(st) % 3 == 0 ? BH_DRAIN_OVER_LIMIT : 0u) | (fz_next(st) % 7 == 0 ? BH_GLOBAL : 0u) |
                     (fz_next(st) % 11 == 0 ? BH_GREGORIAN : 0u);
        r.is_owner = fz_next(st) % 4 != 0;
        r.algorithm = fz_next(st) & 1;
        r.burst = fz_next(st) % 3 == 0 ? (int64_t)(fz_next(st) % 50) : 0;
        Rec s0; rec_clear(s0);
        s0.limit = fz_next(st) % 5 ? r.limit : r.limit + 3; s0.duration = fz_next(st) % 5 ? r.duration : r.duration + 1;
        s0.expire_at = now + (int64_t)(fz_next(st) % 100000) - (fz_next(st) % 16 == 0 ? 200000 : 0);
        s0.invalid_at = fz_next(st) % 9 == 0 ? now + (int64_t)(fz_next(st) % 100) - 50 : 0;
        s0.stamp = now - (int64_t)(fz_next(st) % 100000);
        if (r.algorithm == ALGO_TOKEN) {
            const uint64_t pick = fz_next(st) % 8;
            s0.remaining = pick == 0 ? 0 : pick == 1 ? r.hits : pick == 2 ? (int64_t)((uint64_t)r.hits - 1) : pick == 3 ? (int64_t)((fz_next(st) % 20) * (uint64_t)r.hits)
                         : pick == 4 ? -3 : (int64_t)(fz_next(st) % 300000);
            s0.meta = make_meta(K_TOKEN, fz_next(st) % 5 == 0 ? ST_OVER : ST_UNDER, ALGO_TOKEN);
        } else {
            const int64_t burst = r.burst == 0 ? r.limit : r.burst;
            s0.burst = fz_next(st) % 6 ? burst : burst + 1;
            const uint64_t pick = fz_next(st) % 8;
            double rem = pick == 0 ? 0.0 : pick == 1 ? (double)r.hits : pick == 2 ? (double)r.hits + 0.75 : pick == 3 ? 0.5
                       : pick == 4 ? -2.5 : (double)(fz_next(st) % 200000) / 7.0;
            if (fz_next(st) % 40 == 0) rem = 1e300;
            s0.remaining = f2bits(rem);
            s0.meta = make_meta(K_LEAKY, 0, ALGO_LEAKY);
        }
        static const uint64_t kRank[] = {0, 1, 2, 3, 7, 63, 64, 99, 100, 101, 255, 4096, 65535, 1u << 20};
        for (uint64_t k : kRank) {
            Resp a, b; Rec sa, sb; uint32_t ea = 0;
            bool fast = false;
            if (token_fast_ok(s0, r, now)) { ea = token_fast(s0, r, k, a, sa); fast = true; out[0]++; }
            else if (leaky_fast(s0, r, now, k, a, sa, ea)) { fast = true; out[1]++; }
            if (!fast) break;
            if (k > 4096 && r.hits > 1 && !(r.behavior & BH_DRAIN_OVER_LIMIT)) { /* generic path steps: keep it */ }
            const uint32_t eb = eval_uniform_rank(s0, r, now, k, b, sb);
            {   // the one-site form the kernels inline must agree with the three-site form everywhere
                Resp c; Rec sc;
                const uint32_t ec = eval_uniform_rank_1x(s0, r, now, k, c, sc);
                if (ec != eb || c.status != b.status || c.err != b.err || c.limit != b.limit || c.remaining != b.remaining ||
                    c.reset_time != b.reset_time || !rec_eq(sc, sb))
                    out[2]++;
            }
            if (ea != eb || a.status != b.status || a.err != b.err || a.limit != b.limit || a.remaining != b.remaining ||
                a.reset_time != b.reset_time || !rec_eq(sa, sb))
                out[2]++;
        }
    }
}

int hs_add_item(void* hp, const guber_item_t* in, int* existed) {
    HostSim* h = (HostSim*)hp;
    std::string k((const char*)in->key, in->key_len);
    Rec s; rec_clear(s);
    s.limit = in->limit; s.duration = in->duration; s.stamp = in->stamp; s.burst = in->burst;
    s.expire_at = in->expire_at; s.invalid_at = in->invalid_at;
    if (in->algorithm == ALGO_TOKEN) { s.remaining = in->remaining; s.burst = 0; s.meta = make_meta(K_TOKEN, in->status, ALGO_TOKEN); }   // as rec_from_item (guber_engine.hip)
    else if (in->algorithm == ALGO_LEAKY) { s.remaining = f2bits(in->remaining_f); s.meta = make_meta(K_LEAKY, 0, ALGO_LEAKY); }
    else s.meta = make_meta(K_NIL, 0, in->algorithm);
    auto it = h->table.find(k);
    bool ex = it != h->table.end() && rec_kind(it->second) != K_ABSENT;
    if (!ex) h->size++;
    h->table[k] = s;
    if (existed) *existed = ex;
    return 0;
}
int hs_get_item(void* hp, const uint8_t* key, uint32_t klen, int64_t now_ms, guber_item_t* out, int* found) {
    HostSim* h = (HostSim*)hp;
    std::string k((const char*)key, klen);
    *found = 0;
    auto it = h->table.find(k);
    if (it == h->table.end() || rec_kind(it->second) == K_ABSENT) return 0;
    Rec& s = it->second;
    if (rec_expired(s, now_ms)) { rec_clear(s); h->size--; return 0; }
    *found = 1;
    memset(out, 0, sizeof(*out));
    out->limit = s.limit; out->duration = s.duration; out->stamp = s.stamp; out->burst = s.burst;
    out->expire_at = s.expire_at; out->invalid_at = s.invalid_at;
    if (rec_kind(s) == K_TOKEN) { out->algorithm = ALGO_TOKEN; out->status = (uint8_t)rec_status(s); out->remaining = s.remaining; }
    else if (rec_kind(s) == K_LEAKY) { out->algorithm = ALGO_LEAKY; out->remaining_f = bits2f(s.remaining); }
    else out->algorithm = (uint8_t)rec_algo(s);
    return 0;
}
int64_t hs_size(void* hp) { return ((HostSim*)hp)->size; }
uint64_t hs_xxhash64(const uint8_t* p, uint32_t len, uint64_t seed) { return xxhash64(p, len, seed); }
uint64_t hs_fnv1_64(const uint8_t* p, uint32_t len) { return fnv1_64(p, len); }
uint64_t hs_fnv1a_64(const uint8_t* p, uint32_t len) { return fnv1a_64(p, len); }
}
