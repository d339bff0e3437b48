// worker_pool.cpp — see worker_pool.h.
#include "worker_pool.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

namespace gubernator {

static int64_t mono_us() {
    using namespace std::chrono;
    return duration_cast<microseconds>(steady_clock::now().time_since_epoch()).count();
}

GPUWorkerPool::GPUWorkerPool(const guber_config_t& cfg, uint32_t batch_limit, uint32_t batch_wait_us, uint32_t shards,
                             const std::vector<int32_t>& devices)
    : batch_limit_(batch_limit ? batch_limit : 1000), batch_wait_us_(batch_wait_us ? batch_wait_us : 500) {
    if (shards == 0) shards = 1;
    std::vector<int32_t> devs = devices;
    if (devs.empty()) devs.push_back(cfg.device);
    n_devices_ = (uint32_t)devs.size(); shards_per_device_ = shards;
    ring_step_ = (1ull << 63) / shards;                              // workers.go:132 hashRingStep
    if (n_devices_ > 1) {                                            // the GPUs of the node are the peers of the ring
        std::vector<std::string> names; std::vector<const char*> ptrs;
        for (uint32_t i = 0; i < n_devices_; ++i) names.push_back("gpu" + std::to_string(i));
        for (auto& n : names) ptrs.push_back(n.c_str());
        create_rc_ = guber_ring_create(ptrs.data(), n_devices_, 512, 0, &ring_);
        if (create_rc_ != GUBER_OK) return;
    }
    guber_config_t c = cfg;
    if (c.max_batch < batch_limit_) c.max_batch = batch_limit_;
    const uint32_t total = n_devices_ * shards;
    if (total > 1) c.cache_size = c.cache_size / total + 1;          // workers.go:132 `CacheSize / Workers` per worker
    max_key_ = c.max_key_bytes ? c.max_key_bytes : 1024;
    // room for batch_limit keys of typical size; a batch whose keys do not fit is flushed early (never overrun)
    key_cap_ = (uint32_t)std::min<uint64_t>((uint64_t)batch_limit_ * std::min<uint32_t>(max_key_, 96u) + max_key_, 1u << 30);
    for (uint32_t d = 0; d < n_devices_ && create_rc_ == GUBER_OK; ++d) {
        for (uint32_t i = 0; i < shards; ++i) {
            std::unique_ptr<Shard> sh(new Shard());
            c.device = devs[d];
            sh->device = devs[d];
            create_rc_ = guber_engine_create(&c, &sh->engine);
            if (create_rc_ != GUBER_OK) { sh->engine = nullptr; break; }
            for (int k = 0; k < 2 && create_rc_ == GUBER_OK; ++k) create_rc_ = guber_stage_create(sh->engine, batch_limit_, key_cap_, &sh->stage[k]);
            shards_.push_back(std::move(sh));
            if (create_rc_ != GUBER_OK) break;
        }
    }
    if (create_rc_ != GUBER_OK) {
        for (auto& sh : shards_) { for (auto* st : sh->stage) guber_stage_destroy(st); guber_engine_destroy(sh->engine); }
        shards_.clear();
        return;
    }
    for (auto& sh : shards_) { Shard* p = sh.get(); p->thread = std::thread([this, p] { run(*p); }); }
}

GPUWorkerPool::~GPUWorkerPool() {
    Close();
    if (ring_) guber_ring_destroy(ring_);
}

void GPUWorkerPool::Close() {
    if (closed_.exchange(true)) return;
    for (auto& sh : shards_) {
        { std::lock_guard<std::mutex> lk(sh->mu); sh->closing = true; }
        sh->cv.notify_all();
    }
    for (auto& sh : shards_) {
        if (sh->thread.joinable()) sh->thread.join();
        for (auto*& st : sh->stage) { guber_stage_destroy(st); st = nullptr; }
        if (sh->engine) { guber_engine_destroy(sh->engine); sh->engine = nullptr; }
    }
}

uint32_t GPUWorkerPool::DeviceOf(const uint8_t* key, uint32_t len) const {
    if (n_devices_ <= 1 || !ring_) return 0;
    const uint32_t off[2] = {0, len};
    uint32_t owner = 0;
    guber_ring_route(ring_, key, off, 1, &owner);                                           // replicated_hash.go:104-119
    return owner < n_devices_ ? owner : 0;
}
uint32_t GPUWorkerPool::ShardOf(const uint8_t* key, uint32_t len) const {
    if (shards_.size() <= 1) return 0;
    uint32_t local = 0;
    if (shards_per_device_ > 1) {
        const uint64_t h63 = guber_xxhash64(key, len, 0) >> 1;                              // workers.go:153-155 ComputeHash63
        local = (uint32_t)std::min<uint64_t>(h63 / ring_step_, shards_per_device_ - 1);     // workers.go:180-184 getWorker
    }
    return DeviceOf(key, len) * shards_per_device_ + local;
}
uint64_t GPUWorkerPool::batches_flushed() const {
    uint64_t n = 0;
    for (auto& sh : shards_) n += sh->flushed.load();
    return n;
}
void GPUWorkerPool::Metrics(guber_pool_metrics_t* out) const {
    memset(out, 0, sizeof(*out));
    for (auto& sh : shards_) {
        out->batches += sh->flushed.load(); out->requests += sh->requests.load();
        { std::lock_guard<std::mutex> lk(sh->mu); out->queue_length += sh->queue.size(); }
        out->queue_length_max = std::max<uint64_t>(out->queue_length_max, sh->queue_max.load());
        out->send_duration_us_sum += sh->send_us_sum.load();
        out->send_duration_us_max = std::max<uint64_t>(out->send_duration_us_max, sh->send_us_max.load());
        out->batch_size_max = std::max<uint64_t>(out->batch_size_max, sh->batch_max.load());
        out->in_flight += sh->in_flight.load();
        out->key_too_long += sh->key_too_long.load(); out->flush_on_key_bytes += sh->flush_on_key_bytes.load();
    }
    out->shards = (uint32_t)shards_.size(); out->devices = n_devices_;
}

int64_t GPUWorkerPool::NowMs() const {
    const int64_t f = frozen_ms_.load();
    if (f) return f;
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
    if (closed_.load() || shards_.empty()) {
        for (auto* r : out) r->error = "worker pool is closed";
        return;
    }
    Call call;
    call.remaining = reqs.size();
    // every request goes to the queue of its key's shard (workers.go:261-291); requests of one key keep their order
    std::vector<std::vector<Pending>> per(shards_.size());
    std::string k;
    for (size_t i = 0; i < reqs.size(); ++i) {
        k.assign(reqs[i]->name); k.push_back('_'); k.append(reqs[i]->unique_key);               // HashKey, client.go:39-41
        per[ShardOf((const uint8_t*)k.data(), (uint32_t)k.size())].push_back({reqs[i], st[i], out[i], &call, (uint32_t)k.size()});
    }
    for (size_t j = 0; j < per.size(); ++j) {
        if (per[j].empty()) continue;
        Shard& sh = *shards_[j];
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.queue.insert(sh.queue.end(), per[j].begin(), per[j].end());
            const uint64_t q = sh.queue.size();
            if (q > sh.queue_max.load()) sh.queue_max.store(q);
        }
        sh.cv.notify_all();
    }
    std::unique_lock<std::mutex> lk(call.mu);
    call.cv.wait(lk, [&] { return call.remaining == 0; });
}

void GPUWorkerPool::answer(Pending& p, int rc, uint8_t err, uint8_t status, int64_t limit, int64_t remaining, int64_t reset_time) {
    RateLimitResp& o = *p.resp;
    o = RateLimitResp{};
    if (rc != GUBER_OK) {
        o.error = std::string("gpu engine: ") + guber_strerror(rc);
    } else if (err != 0) {
        char buf[256];
        if (err == GUBER_ITEM_E_INVALID_ALGORITHM) snprintf(buf, sizeof buf, guber_item_strerror(err), p.req->algorithm);
        else snprintf(buf, sizeof buf, "%s", guber_item_strerror(err));
        o.error = buf;                                               // nil response + error (workers.go:317-321)
    } else {
        o.status = status; o.limit = limit; o.remaining = remaining; o.reset_time = reset_time;
    }
    Call* c = p.call;
    // notify while holding the lock: the waiter owns the Call (it lives on its stack) and may destroy it as soon as it
    // can re-acquire the mutex and sees remaining == 0
    std::lock_guard<std::mutex> lk(c->mu);
    if (--c->remaining == 0) c->cv.notify_all();
}

void GPUWorkerPool::run(Shard& sh) {
    Flight fl[2];
    fl[0].stage = sh.stage[0]; fl[1].stage = sh.stage[1];
    for (auto& f : fl) f.batch.reserve(batch_limit_);
    int cur = 0;
    bool inflight = false;                      // fl[cur ^ 1] has been submitted and not completed yet
    for (;;) {
        Flight& f = fl[cur];
        {
            std::unique_lock<std::mutex> lk(sh.mu);
            if (inflight && (sh.queue.empty() || (sh.queue.size() < batch_limit_ && !sh.closing))) {
                // nothing (or not yet a full batch) to overlap with: deliver the batch in flight first
                lk.unlock();
                complete(sh, fl[cur ^ 1], GUBER_OK);
                inflight = false;
                continue;
            }
            sh.cv.wait(lk, [&] { return sh.closing || !sh.queue.empty(); });
            if (sh.queue.empty() && sh.closing) break;
            // flush at batch_limit or batch_wait after the first queued item (peer_client.go:284-337)
            if (sh.queue.size() < batch_limit_ && !sh.closing)
                sh.cv.wait_for(lk, std::chrono::microseconds(batch_wait_us_), [&] { return sh.closing || sh.queue.size() >= batch_limit_; });
            // take as many as fit: batch_limit items, and their keys into the stage's key buffer
            size_t take = 0; uint64_t kb = 0;
            while (take < sh.queue.size() && take < batch_limit_) {
                const uint32_t kl = sh.queue[take].key_len <= max_key_ ? sh.queue[take].key_len : 0;   // over-long keys are answered, not copied
                if (kb + kl > key_cap_) { sh.flush_on_key_bytes++; break; }
                kb += kl; take++;
            }
            f.batch.assign(sh.queue.begin(), sh.queue.begin() + take);
            sh.queue.erase(sh.queue.begin(), sh.queue.begin() + take);
        }
        f.t0_us = mono_us();
        const bool submitted = fill_and_submit(sh, f);
        if (inflight) { complete(sh, fl[cur ^ 1], GUBER_OK); inflight = false; }
        if (submitted) { inflight = true; cur ^= 1; }
    }
    if (inflight) complete(sh, fl[cur ^ 1], GUBER_OK);
}

// write the batch into the stage in place and submit it (asynchronous).  With a persistent Store configured the batch takes
// the synchronous path that makes the Store's calls (flush_with_store).
bool GPUWorkerPool::fill_and_submit(Shard& sh, Flight& f) {
    const uint32_t n = (uint32_t)f.batch.size();
    sh.requests += n;
    if (n > sh.batch_max.load()) sh.batch_max.store(n);
    if (has_store_.load()) { flush_with_store(sh, f.batch); sh.flushed++; f.batch.clear(); return false; }
    guber_batch_t* b = guber_stage_batch(f.stage);
    const int64_t now = NowMs();
    uint8_t* kp = (uint8_t*)b->key_bytes;
    uint32_t* off = (uint32_t*)b->key_off;
    int64_t *hits = (int64_t*)b->hits, *limit = (int64_t*)b->limit, *duration = (int64_t*)b->duration, *burst = (int64_t*)b->burst,
            *created = (int64_t*)b->created_at;
    uint8_t *algo = (uint8_t*)b->algorithm, *owner = (uint8_t*)b->is_owner;
    uint32_t* beh = (uint32_t*)b->behavior;
    uint32_t o = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const RateLimitReq& r = *f.batch[i].req;
        off[i] = o;
        if (f.batch[i].key_len <= max_key_) {                         // HashKey = name + "_" + unique_key, written in place
            memcpy(kp + o, r.name.data(), r.name.size()); o += (uint32_t)r.name.size();
            kp[o++] = '_';
            memcpy(kp + o, r.unique_key.data(), r.unique_key.size()); o += (uint32_t)r.unique_key.size();
        }                                                            // else: zero-length key -> answered below as key too long
        hits[i] = r.hits; limit[i] = r.limit; duration[i] = r.duration; burst[i] = r.burst;
        created[i] = r.created_at ? r.created_at : now;
        algo[i] = (r.algorithm == 0 || r.algorithm == 1) ? (uint8_t)r.algorithm : 255;
        beh[i] = r.behavior; owner[i] = f.batch[i].st.is_owner ? 1 : 0;
    }
    off[n] = o;
    b->n = n; b->now_ms = now;                                        // DURATION_IS_GREGORIAN: the kernels derive the interval from now_ms
    sh.in_flight++;
    const int rc = guber_stage_submit(f.stage);
    if (rc != GUBER_OK) { complete(sh, f, rc); return false; }
    return true;
}

void GPUWorkerPool::complete(Shard& sh, Flight& f, int rc) {
    if (rc == GUBER_OK) rc = guber_stage_wait(f.stage);
    const guber_result_t* r = guber_stage_result(f.stage);
    const uint32_t n = (uint32_t)f.batch.size();
    for (uint32_t i = 0; i < n; ++i) {
        Pending& p = f.batch[i];
        if (p.key_len > max_key_) { sh.key_too_long++; answer(p, GUBER_OK, GUBER_ITEM_E_KEY_TOO_LONG, 0, 0, 0, 0); continue; }
        answer(p, rc, rc == GUBER_OK ? r->err[i] : 0, r->status[i], r->limit[i], r->remaining[i], r->reset_time[i]);
    }
    const uint64_t us = (uint64_t)std::max<int64_t>(mono_us() - f.t0_us, 0);
    sh.send_us_sum += us;
    if (us > sh.send_us_max.load()) sh.send_us_max.store(us);
    sh.flushed++; sh.in_flight--;
    f.batch.clear();
}

// Config.Store (store.go:49-65) configured: ask the store for keys that are not resident BEFORE the batch
// (algorithms.go:45-51 `s.Get` on a cache miss, then `c.Add(item)`), evaluate, then issue the Remove / OnChange
// calls the reference makes from inside the algorithms, in request order.  Synchronous, host-pointer entry points.
void GPUWorkerPool::flush_with_store(Shard& sh, std::vector<Pending>& batch) {
    guber_engine_t* const engine_ = sh.engine;
    const uint32_t n = (uint32_t)batch.size();
    const int64_t now = NowMs();
    std::vector<uint8_t> keys; std::vector<uint32_t> off(n + 1), beh(n);
    std::vector<int64_t> hits(n), limit(n), duration(n), burst(n), created(n);
    std::vector<uint8_t> algo(n), owner(n), status(n), err(n);
    std::vector<int64_t> rlimit(n), rremaining(n), rreset(n);
    for (uint32_t i = 0; i < n; ++i) {
        const RateLimitReq& r = *batch[i].req;
        off[i] = (uint32_t)keys.size();
        if (batch[i].key_len <= max_key_) { const std::string k = r.HashKey(); keys.insert(keys.end(), k.begin(), k.end()); }
        hits[i] = r.hits; limit[i] = r.limit; duration[i] = r.duration; burst[i] = r.burst;
        created[i] = r.created_at ? r.created_at : now;
        algo[i] = (r.algorithm == 0 || r.algorithm == 1) ? (uint8_t)r.algorithm : 255;
        beh[i] = r.behavior; owner[i] = batch[i].st.is_owner ? 1 : 0;
    }
    off[n] = (uint32_t)keys.size();
    keys.resize(keys.size() + 16, 0);
    guber_batch_t b{}; guber_result_t res{};
    b.n = n; b.key_bytes = keys.data(); b.key_off = off.data(); b.hits = hits.data(); b.limit = limit.data();
    b.duration = duration.data(); b.burst = burst.data(); b.created_at = created.data(); b.algorithm = algo.data();
    b.behavior = beh.data(); b.is_owner = owner.data(); b.now_ms = now;
    res.status = status.data(); res.limit = rlimit.data(); res.remaining = rremaining.data(); res.reset_time = rreset.data();
    res.err = err.data();
    std::vector<uint8_t> sflags(n, 0); std::vector<guber_item_t> sitems(n);
    guber_store_events_t sev{sflags.data(), sitems.data()};
    auto store_req = [&](uint32_t i) {
        const RateLimitReq& r = *batch[i].req;
        guber_store_req_t q{};
        q.key = keys.data() + off[i]; q.key_len = off[i + 1] - off[i]; q.name_len = (uint32_t)r.name.size();
        q.hits = r.hits; q.limit = r.limit; q.duration = r.duration; q.burst = r.burst; q.created_at = created[i];
        q.algorithm = r.algorithm; q.behavior = r.behavior;
        return q;
    };
    std::vector<uint8_t> missing(n, 0);
    int rc = guber_probe_missing(engine_, &b, missing.data());
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
    if (rc == GUBER_OK) rc = guber_eval_batch_store(engine_, &b, &res, &sev);
    if (rc == GUBER_OK) {
        for (uint32_t i = 0; i < n; ++i) {
            if ((sflags[i] & GUBER_STORE_REMOVE) && store_.remove) store_.remove(store_.user, keys.data() + off[i], off[i + 1] - off[i]);
            if ((sflags[i] & GUBER_STORE_ONCHANGE) && store_.on_change) { const guber_store_req_t q = store_req(i); store_.on_change(store_.user, &q, &sitems[i]); }
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        if (batch[i].key_len > max_key_) { answer(batch[i], GUBER_OK, GUBER_ITEM_E_KEY_TOO_LONG, 0, 0, 0, 0); continue; }
        answer(batch[i], rc, rc == GUBER_OK ? err[i] : 0, status[i], rlimit[i], rremaining[i], rreset[i]);
    }
}

int GPUWorkerPool::AddCacheItem(const guber_item_t& item) {
    if (shards_.empty()) return GUBER_E_INVALID_ARG;
    return guber_add_items(shards_[ShardOf(item.key, item.key_len)]->engine, &item, 1, nullptr);
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
    for (uint32_t i = 0; i < n; ++i) per[ShardOf(items[i].key, items[i].key_len)].push_back(items[i]);
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
    return guber_pool_create_multi(cfg, nullptr, 0, shards, batch_limit, batch_wait_us, out);
}
extern "C" int guber_pool_create_multi(const guber_config_t* cfg, const int32_t* devices, uint32_t n_devices, uint32_t shards_per_device,
                                       uint32_t batch_limit, uint32_t batch_wait_us, guber_pool_t** out) {
    if (!cfg || !out || (n_devices && !devices)) return GUBER_E_INVALID_ARG;
    std::vector<int32_t> devs(devices, devices + n_devices);
    GPUWorkerPool* p = new GPUWorkerPool(*cfg, batch_limit, batch_wait_us, shards_per_device, devs);
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
    return p ? p->pool->ShardOf(key, key_len) : 0;
}
extern "C" int guber_pool_load(guber_pool_t* p, const guber_item_t* items, uint32_t n) { return p ? p->pool->Load(items, n) : GUBER_E_INVALID_ARG; }
extern "C" int guber_pool_store(guber_pool_t* p, void (*save)(void* user, const guber_item_t* item), void* user) {
    if (!p || !save) return GUBER_E_INVALID_ARG;
    return p->pool->Store([&](const guber_item_t& it) { save(user, &it); });
}
extern "C" void guber_pool_set_store(guber_pool_t* p, const guber_store_callbacks_t* cb) { if (p) p->pool->SetStore(cb); }
extern "C" void guber_pool_set_clock(guber_pool_t* p, int64_t now_ms) { if (p) p->pool->SetClockMs(now_ms); }
extern "C" guber_engine_t* guber_pool_engine(guber_pool_t* p) { return p ? p->pool->engine() : nullptr; }
extern "C" guber_engine_t* guber_pool_engine_at(guber_pool_t* p, uint32_t shard) { return p ? p->pool->engine(shard) : nullptr; }
extern "C" uint32_t guber_pool_shards(guber_pool_t* p) { return p ? p->pool->shards() : 0; }
extern "C" uint32_t guber_pool_device_of(guber_pool_t* p, const uint8_t* key, uint32_t key_len) { return p ? p->pool->DeviceOf(key, key_len) : 0; }
extern "C" int guber_pool_metrics(guber_pool_t* p, guber_pool_metrics_t* out) {
    if (!p || !out) return GUBER_E_INVALID_ARG;
    p->pool->Metrics(out);
    return GUBER_OK;
}
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
