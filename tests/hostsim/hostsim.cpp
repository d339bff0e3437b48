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
                uint32_t ev = eval_uniform_rank(s0, r, b->now_ms, j, out, after);
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
