// worker_pool.cpp — see worker_pool.h.
#include "worker_pool.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

namespace gubernator {

GPUWorkerPool::GPUWorkerPool(const guber_config_t& cfg, uint32_t batch_limit, uint32_t batch_wait_us, uint32_t shards)
    : batch_limit_(batch_limit ? batch_limit : 1000), batch_wait_us_(batch_wait_us ? batch_wait_us : 500) {
    if (shards == 0) shards = 1;
    ring_step_ = (1ull << 63) / shards;                              // workers.go:132 hashRingStep
    guber_config_t c = cfg;
    if (c.max_batch < batch_limit_) c.max_batch = batch_limit_;
    if (shards > 1) c.cache_size = c.cache_size / shards + 1;        // workers.go:132 `CacheSize / Workers` per worker
    for (uint32_t i = 0; i < shards; ++i) {
        std::unique_ptr<Shard> sh(new Shard());
        create_rc_ = guber_engine_create(&c, &sh->engine);
        if (create_rc_ != GUBER_OK) { sh->engine = nullptr; break; }
        shards_.push_back(std::move(sh));
    }
    if (create_rc_ != GUBER_OK) {
        for (auto& sh : shards_) guber_engine_destroy(sh->engine);
        shards_.clear();
        return;
    }
    for (auto& sh : shards_) { Shard* p = sh.get(); p->thread = std::thread([this, p] { run(*p); }); }
}

GPUWorkerPool::~GPUWorkerPool() { Close(); }

void GPUWorkerPool::Close() {
    if (closed_) return;
    closed_ = true;
    for (auto& sh : shards_) {
        { std::lock_guard<std::mutex> lk(sh->mu); sh->closing = true; }
        sh->cv.notify_all();
    }
    for (auto& sh : shards_) {
        if (sh->thread.joinable()) sh->thread.join();
        if (sh->engine) { guber_engine_destroy(sh->engine); sh->engine = nullptr; }
    }
}

uint32_t GPUWorkerPool::ShardOf(const std::string& key) const {
    if (shards_.size() <= 1) return 0;
    const uint64_t h63 = guber_xxhash64((const uint8_t*)key.data(), key.size(), 0) >> 1;   // workers.go:153-155 ComputeHash63
    const uint64_t idx = h63 / ring_step_;                                                  // workers.go:180-184 getWorker
    return (uint32_t)std::min<uint64_t>(idx, shards_.size() - 1);
}
uint64_t GPUWorkerPool::batches_flushed() const {
    uint64_t n = 0;
    for (auto& sh : shards_) n += sh->flushed;
    return n;
}

int64_t GPUWorkerPool::NowMs() const {
    if (frozen_ms_) return frozen_ms_;
    using namespace std::chrono;
    return duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();   // MillisecondNow, lrucache.go:106
}

bool GPUWorkerPool::GetRateLimit(const RateLimitReq& r, RateLimitReqState st, RateLimitResp* resp) {
    std::vector<const RateLimitReq*> reqs{&r};
    std::vector<RateLimitReqState> sts{st};
    std::vector<RateLimitResp*> out{resp};
    GetRateLimitMany(reqs, sts, out);
    return resp->error.empty();
}

void GPUWorkerPool::GetRateLimitMany(const std::vector<const RateLimitReq*>& reqs, const std::vector<RateLimitReqState>& st,
                                     std::vector<RateLimitResp*>& out) {
    if (reqs.empty()) return;
    if (closed_ || shards_.empty()) {
        for (auto* r : out) r->error = "worker pool is closed";
        return;
    }
    Call call;
    call.remaining = reqs.size();
    // every request goes to the queue of its key's shard (workers.go:261-291); requests of one key keep their order
    std::vector<std::vector<Pending>> per(shards_.size());
    for (size_t i = 0; i < reqs.size(); ++i)
        per[shards_.size() > 1 ? ShardOf(reqs[i]->HashKey()) : 0].push_back({reqs[i], st[i], out[i], &call});
    for (size_t j = 0; j < per.size(); ++j) {
        if (per[j].empty()) continue;
        Shard& sh = *shards_[j];
        { std::lock_guard<std::mutex> lk(sh.mu); sh.queue.insert(sh.queue.end(), per[j].begin(), per[j].end()); }
        sh.cv.notify_all();
    }
    std::unique_lock<std::mutex> lk(call.mu);
    call.cv.wait(lk, [&] { return call.remaining == 0; });
}

void GPUWorkerPool::run(Shard& sh) {
    std::vector<Pending> batch;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(sh.mu);
            sh.cv.wait(lk, [&] { return sh.closing || !sh.queue.empty(); });
            if (sh.queue.empty() && sh.closing) return;
            // flush at batch_limit or batch_wait after the first queued item (peer_client.go:284-337)
            if (sh.queue.size() < batch_limit_ && !sh.closing)
                sh.cv.wait_for(lk, std::chrono::microseconds(batch_wait_us_), [&] { return sh.closing || sh.queue.size() >= batch_limit_; });
            const size_t take = std::min<size_t>(sh.queue.size(), batch_limit_);
            batch.assign(sh.queue.begin(), sh.queue.begin() + take);
            sh.queue.erase(sh.queue.begin(), sh.queue.begin() + take);
        }
        flush(sh, batch);
        batch.clear();
    }
}

void GPUWorkerPool::flush(Shard& sh, std::vector<Pending>& batch) {
    guber_engine_t* const engine_ = sh.engine;
    const uint32_t n = (uint32_t)batch.size();
    const int64_t now = NowMs();
    std::vector<uint8_t> keys; std::vector<uint32_t> off(n + 1), beh(n);
    std::vector<int64_t> hits(n), limit(n), duration(n), burst(n), created(n), gexp(n, 0), gdur(n, 0);
    std::vector<uint8_t> algo(n), owner(n), status(n), err(n);
    std::vector<int64_t> rlimit(n), rremaining(n), rreset(n);
    bool any_greg = false;
    for (uint32_t i = 0; i < n; ++i) {
        const RateLimitReq& r = *batch[i].req;
        const std::string k = r.HashKey();
        off[i] = (uint32_t)keys.size();
        keys.insert(keys.end(), k.begin(), k.end());
        hits[i] = r.hits; limit[i] = r.limit; duration[i] = r.duration; burst[i] = r.burst;
        created[i] = r.created_at ? r.created_at : now;
        algo[i] = (r.algorithm == 0 || r.algorithm == 1) ? (uint8_t)r.algorithm : 255;
        beh[i] = r.behavior; owner[i] = batch[i].st.is_owner ? 1 : 0;
        if (r.behavior & GUBER_BEHAVIOR_DURATION_IS_GREGORIAN) {      // interval.go:84-148 with clock.Now()
            any_greg = true;
            int64_t e = 0, d = 0;
            int rc = guber_gregorian_expiration(now * 1000000, r.duration, &e);
            if (rc == 0) rc = guber_gregorian_duration(now * 1000000, r.duration, &d);
            gexp[i] = e; gdur[i] = rc ? rc : d;
        }
    }
    off[n] = (uint32_t)keys.size();
    keys.resize(keys.size() + 16, 0);
    guber_batch_t b{}; guber_result_t res{};
    b.n = n; b.key_bytes = keys.data(); b.key_off = off.data(); b.hits = hits.data(); b.limit = limit.data();
    b.duration = duration.data(); b.burst = burst.data(); b.created_at = created.data(); b.algorithm = algo.data();
    b.behavior = beh.data(); b.is_owner = owner.data(); b.now_ms = now;
    if (any_greg) { b.greg_expire = gexp.data(); b.greg_duration = gdur.data(); }
    res.status = status.data(); res.limit = rlimit.data(); res.remaining = rremaining.data(); res.reset_time = rreset.data();
    res.err = err.data();
    // Config.Store (store.go:49-65): ask the store for keys that are not resident BEFORE the batch
    // (algorithms.go:45-51 `s.Get` on a cache miss, then `c.Add(item)`), evaluate, then issue the Remove / OnChange
    // calls the reference makes from inside the algorithms, in request order.
    std::vector<uint8_t> sflags; std::vector<guber_item_t> sitems;
    guber_store_events_t sev{nullptr, nullptr};
    auto store_req = [&](uint32_t i) {
        const RateLimitReq& r = *batch[i].req;
        guber_store_req_t q{};
        q.key = keys.data() + off[i]; q.key_len = off[i + 1] - off[i]; q.name_len = (uint32_t)r.name.size();
        q.hits = r.hits; q.limit = r.limit; q.duration = r.duration; q.burst = r.burst; q.created_at = created[i];
        q.algorithm = r.algorithm; q.behavior = r.behavior;
        return q;
    };
    int rc = GUBER_OK;
    if (has_store_) {
        std::vector<uint8_t> missing(n, 0);
        rc = guber_probe_missing(engine_, &b, missing.data());
        if (rc == GUBER_OK && store_.get) {
            std::vector<std::string> asked;
            for (uint32_t i = 0; i < n && rc == GUBER_OK; ++i) {
                if (!missing[i] || off[i + 1] == off[i]) continue;
                std::string k((const char*)keys.data() + off[i], off[i + 1] - off[i]);
                if (std::find(asked.begin(), asked.end(), k) != asked.end()) continue;
                asked.push_back(k);
                guber_item_t it{};
                const guber_store_req_t q = store_req(i);
                if (store_.get(store_.user, &q, &it)) {
                    it.key = (const uint8_t*)k.data(); it.key_len = (uint32_t)k.size();
                    rc = guber_add_items(engine_, &it, 1, nullptr);
                }
            }
        }
        sflags.assign(n, 0); sitems.resize(n);
        sev.flags = sflags.data(); sev.items = sitems.data();
    }
    if (rc == GUBER_OK) rc = has_store_ ? guber_eval_batch_store(engine_, &b, &res, &sev) : guber_eval_batch(engine_, &b, &res);
    if (rc == GUBER_OK && has_store_) {
        for (uint32_t i = 0; i < n; ++i) {
            if ((sflags[i] & GUBER_STORE_REMOVE) && store_.remove) store_.remove(store_.user, keys.data() + off[i], off[i + 1] - off[i]);
            if ((sflags[i] & GUBER_STORE_ONCHANGE) && store_.on_change) { const guber_store_req_t q = store_req(i); store_.on_change(store_.user, &q, &sitems[i]); }
        }
    }
    sh.flushed++;
    for (uint32_t i = 0; i < n; ++i) {
        RateLimitResp& o = *batch[i].resp;
        o = RateLimitResp{};
        if (rc != GUBER_OK) {
            o.error = std::string("gpu engine: ") + guber_strerror(rc);
        } else if (err[i] != 0) {
            char buf[256];
            if (err[i] == GUBER_ITEM_E_INVALID_ALGORITHM) snprintf(buf, sizeof buf, guber_item_strerror(err[i]), batch[i].req->algorithm);
            else snprintf(buf, sizeof buf, "%s", guber_item_strerror(err[i]));
            o.error = buf;                                           // nil response + error (workers.go:317-321)
        } else {
            o.status = status[i]; o.limit = rlimit[i]; o.remaining = rremaining[i]; o.reset_time = rreset[i];
        }
        Call* c = batch[i].call;
        bool last;
        { std::lock_guard<std::mutex> lk(c->mu); last = --c->remaining == 0; }
        if (last) c->cv.notify_all();
    }
}

int GPUWorkerPool::AddCacheItem(const guber_item_t& item) {
    if (shards_.empty()) return GUBER_E_INVALID_ARG;
    return guber_add_items(shards_[ShardOf(std::string((const char*)item.key, item.key_len))]->engine, &item, 1, nullptr);
}
int GPUWorkerPool::GetCacheItem(const std::string& key, guber_item_t* out, bool* found) {
    int f = 0;
    if (shards_.empty()) return GUBER_E_INVALID_ARG;
    const int rc = guber_get_item(shards_[ShardOf(key)]->engine, (const uint8_t*)key.data(), (uint32_t)key.size(), NowMs(), out, &f);
    *found = f != 0;
    return rc;
}
int GPUWorkerPool::Load(const guber_item_t* items, uint32_t n) {
    // workers.go:329-449: every item goes to the worker that owns its key; chunks bound the staging buffers
    std::vector<std::vector<guber_item_t>> per(shards_.size());
    for (uint32_t i = 0; i < n; ++i) per[ShardOf(std::string((const char*)items[i].key, items[i].key_len))].push_back(items[i]);
    for (size_t j = 0; j < per.size(); ++j)
        for (size_t lo = 0; lo < per[j].size(); lo += 65536) {
            const int rc = guber_add_items(shards_[j]->engine, per[j].data() + lo, (uint32_t)std::min<size_t>(65536, per[j].size() - lo), nullptr);
            if (rc != GUBER_OK) return rc;
        }
    return GUBER_OK;
}
int GPUWorkerPool::Store(const std::function<void(const guber_item_t&)>& save) {
    for (auto& sh : shards_) {                                                        // workers.go:451-534: every worker in turn
        uint64_t n = 0, arena = 0;
        int rc = guber_dump(sh->engine, nullptr, 0, nullptr, 0, &n, &arena);          // sizes first
        if (rc != GUBER_OK && rc != GUBER_E_NOMEM) return rc;
        std::vector<guber_item_t> items(n + 1024);
        std::vector<uint8_t> keys(arena + 64 * 1024);
        rc = guber_dump(sh->engine, items.data(), items.size(), keys.data(), keys.size(), &n, &arena);
        if (rc != GUBER_OK) return rc;
        for (uint64_t i = 0; i < n; ++i) save(items[i]);
    }
    return GUBER_OK;
}
int64_t GPUWorkerPool::Size() {
    int64_t n = 0;
    for (auto& sh : shards_) n += guber_size(sh->engine);
    return n;
}

bool V1Instance::GetRateLimits(std::vector<RateLimitReq>& reqs, std::vector<RateLimitResp>* resps, std::string* rpc_error) {
    if (reqs.size() > kMaxBatchSize) {                                // gubernator.go:189-193
        char buf[128];
        snprintf(buf, sizeof buf, "Requests.RateLimits list too large; max size is '%u'", kMaxBatchSize);
        *rpc_error = buf;
        return false;
    }
    const int64_t created_at = pool_->NowMs();                        // :195
    resps->assign(reqs.size(), RateLimitResp{});
    std::vector<const RateLimitReq*> send; std::vector<RateLimitReqState> st; std::vector<RateLimitResp*> out;
    for (size_t i = 0; i < reqs.size(); ++i) {
        RateLimitReq& r = reqs[i];
        if (r.unique_key.empty()) { (*resps)[i].error = "field 'unique_key' cannot be empty"; continue; }   // :208-212
        if (r.name.empty()) { (*resps)[i].error = "field 'namespace' cannot be empty"; continue; }          // :213-217
        if (r.created_at == 0) r.created_at = created_at;                                                    // :218-220
        send.push_back(&r); st.push_back(RateLimitReqState{true}); out.push_back(&(*resps)[i]);
    }
    pool_->GetRateLimitMany(send, st, out);
    for (size_t i = 0; i < reqs.size(); ++i) {
        RateLimitResp& o = (*resps)[i];
        if (!o.error.empty() && o.error.rfind("field '", 0) != 0) {
            // gubernator.go:250-255: errors of the local path are wrapped with the key
            o.error = "Error while apply rate limit for '" + reqs[i].HashKey() + "': " + o.error;
        }
    }
    return true;
}

}  // namespace gubernator

// ---- C entry points for bindings / tests ------------------------------------------------------------
using namespace gubernator;
struct guber_pool { GPUWorkerPool* pool; V1Instance* inst; };

extern "C" int guber_pool_create(const guber_config_t* cfg, uint32_t batch_limit, uint32_t batch_wait_us, guber_pool_t** out) {
    return guber_pool_create_sharded(cfg, 1, batch_limit, batch_wait_us, out);
}
extern "C" int guber_pool_create_sharded(const guber_config_t* cfg, uint32_t shards, uint32_t batch_limit, uint32_t batch_wait_us,
                                         guber_pool_t** out) {
    if (!cfg || !out) return GUBER_E_INVALID_ARG;
    GPUWorkerPool* p = new GPUWorkerPool(*cfg, batch_limit, batch_wait_us, shards);
    if (!p->ok()) { const int rc = p->create_error(); delete p; return rc; }
    *out = new guber_pool{p, new V1Instance(p)};
    return GUBER_OK;
}
extern "C" void guber_pool_destroy(guber_pool_t* p) {
    if (!p) return;
    p->pool->Close();
    delete p->inst; delete p->pool; delete p;
}
extern "C" uint32_t guber_pool_shard_of(guber_pool_t* p, const uint8_t* key, uint32_t key_len) {
    return p ? p->pool->ShardOf(std::string((const char*)key, key_len)) : 0;
}
extern "C" int guber_pool_load(guber_pool_t* p, const guber_item_t* items, uint32_t n) { return p ? p->pool->Load(items, n) : GUBER_E_INVALID_ARG; }
extern "C" int guber_pool_store(guber_pool_t* p, void (*save)(void* user, const guber_item_t* item), void* user) {
    if (!p || !save) return GUBER_E_INVALID_ARG;
    return p->pool->Store([&](const guber_item_t& it) { save(user, &it); });
}
extern "C" void guber_pool_set_store(guber_pool_t* p, const guber_store_callbacks_t* cb) { if (p) p->pool->SetStore(cb); }
extern "C" void guber_pool_set_clock(guber_pool_t* p, int64_t now_ms) { if (p) p->pool->SetClockMs(now_ms); }
extern "C" guber_engine_t* guber_pool_engine(guber_pool_t* p) { return p ? p->pool->engine() : nullptr; }
extern "C" uint64_t guber_pool_batches(guber_pool_t* p) { return p ? p->pool->batches_flushed() : 0; }

extern "C" int guber_pool_get_rate_limits(guber_pool_t* p, uint32_t n, const uint8_t* name_bytes, const uint32_t* name_off,
                                          const uint8_t* ukey_bytes, const uint32_t* ukey_off, const int64_t* hits,
                                          const int64_t* limit, const int64_t* duration, const int64_t* burst,
                                          const int64_t* created_at, const int32_t* algorithm, const uint32_t* behavior,
                                          guber_result_t* out, char* err_text, uint32_t err_stride) {
    if (!p || !out) return GUBER_E_INVALID_ARG;
    std::vector<RateLimitReq> reqs(n);
    for (uint32_t i = 0; i < n; ++i) {
        RateLimitReq& r = reqs[i];
        r.name.assign((const char*)name_bytes + name_off[i], name_off[i + 1] - name_off[i]);
        r.unique_key.assign((const char*)ukey_bytes + ukey_off[i], ukey_off[i + 1] - ukey_off[i]);
        r.hits = hits[i]; r.limit = limit[i]; r.duration = duration[i]; r.burst = burst ? burst[i] : 0;
        r.created_at = created_at ? created_at[i] : 0; r.algorithm = algorithm ? algorithm[i] : 0;
        r.behavior = behavior ? behavior[i] : 0;
    }
    std::vector<RateLimitResp> resps;
    std::string rpc_error;
    if (!p->inst->GetRateLimits(reqs, &resps, &rpc_error)) {
        if (err_text && err_stride) snprintf(err_text, err_stride, "%s", rpc_error.c_str());
        return GUBER_E_BATCH_TOO_LARGE;
    }
    for (uint32_t i = 0; i < n; ++i) {
        out->status[i] = (uint8_t)resps[i].status; out->limit[i] = resps[i].limit; out->remaining[i] = resps[i].remaining;
        out->reset_time[i] = resps[i].reset_time; out->err[i] = resps[i].error.empty() ? 0 : 1;
        if (err_text && err_stride) snprintf(err_text + (size_t)i * err_stride, err_stride, "%s", resps[i].error.c_str());
    }
    return GUBER_OK;
}
