// worker_pool.cpp — see worker_pool.h.
#include "worker_pool.h"

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstdio>
#include <cstring>

namespace gubernator {

// short timed waits against the system clock (pthread_cond_timedwait, which ThreadSanitizer understands; a clock step only
// makes one of these microsecond waits end early or late, and every waiter re-checks its condition)
static void wait_us(std::condition_variable& cv, std::unique_lock<std::mutex>& lk, int64_t us) {
    cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(us));
}

// Callers wait for their generation on a plain futex word: the batcher's announcement wakes exactly the threads sleeping on
// that stage, each of which re-reads the word and goes on — no mutex to queue up behind (a condition variable made every
// batch of small RPCs a convoy of all its callers).
static void futex_wait(std::atomic<uint32_t>* w, uint32_t seen) {
    syscall(SYS_futex, (uint32_t*)w, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
}
static void futex_wake_all(std::atomic<uint32_t>* w) { syscall(SYS_futex, (uint32_t*)w, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

static int64_t mono_us() {
    using namespace std::chrono;
    return duration_cast<microseconds>(steady_clock::now().time_since_epoch()).count();
}

GPUWorkerPool::GPUWorkerPool(const guber_config_t& cfg, uint32_t batch_limit, uint32_t batch_wait_us, uint32_t shards,
                             const std::vector<int32_t>& devices)
    : batch_limit_(batch_limit ? batch_limit : 1000), batch_wait_us_(batch_wait_us ? batch_wait_us : 500) {
    if (const char* v = getenv("GUBER_POOL_IDLE_US")) idle_us_ = (uint32_t)atoi(v);   // 0 = off (the reference's policy: limit or wait only)
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
            for (uint32_t k = 0; k < kStages && create_rc_ == GUBER_OK; ++k) {
                Stage& st = sh->st[k];
                create_rc_ = guber_stage_create(sh->engine, batch_limit_, key_cap_, &st.stage);
                if (create_rc_ != GUBER_OK) break;
                st.b = guber_stage_batch(st.stage); st.r = guber_stage_result(st.stage);
                st.name_len.assign(batch_limit_, 0);
            }
            shards_.push_back(std::move(sh));
            if (create_rc_ != GUBER_OK) break;
        }
    }
    if (create_rc_ != GUBER_OK) {
        for (auto& sh : shards_) { for (auto& st : sh->st) guber_stage_destroy(st.stage); guber_engine_destroy(sh->engine); }
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
        sh->cv_batcher.notify_all();
    }
    for (auto& sh : shards_) {
        if (sh->thread.joinable()) sh->thread.join();            // every reserved request has been evaluated and announced
        for (auto& st : sh->st) {
            // callers may still be copying their responses out of the stage
            while (st.gen.load() && st.consumed.load(std::memory_order_acquire) != st.n) std::this_thread::yield();
            guber_stage_destroy(st.stage); st.stage = nullptr;
        }
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
        for (auto& st : sh->st) { const uint64_t w = st.word.load(); if (!(w & kClosed)) out->queue_length += (uint32_t)w; }
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

// ---- the callers' side ---------------------------------------------------------------------------------------------------
struct GPUWorkerPool::Job {
    const std::vector<const RateLimitReq*>& reqs;
    const std::vector<RateLimitReqState>& st;
    std::vector<RateLimitResp*>& out;
    std::vector<uint32_t> key_len;                    // HashKey length per request
    std::vector<std::vector<uint32_t>> per;           // per shard: request indices in call order
    std::vector<Ticket> tickets;
};

static void answer_item(RateLimitResp& o, const RateLimitReq& req, int rc, uint8_t err, uint8_t status, int64_t limit, int64_t remaining,
                        int64_t reset_time) {
    o = RateLimitResp{};
    if (rc != GUBER_OK) {
        o.error = std::string("gpu engine: ") + guber_strerror(rc);
    } else if (err != 0) {
        char buf[256];
        if (err == GUBER_ITEM_E_INVALID_ALGORITHM) snprintf(buf, sizeof buf, guber_item_strerror(err), req.algorithm);
        else snprintf(buf, sizeof buf, "%s", guber_item_strerror(err));
        o.error = buf;                                               // nil response + error (workers.go:317-321)
    } else {
        o.status = status; o.limit = limit; o.remaining = remaining; o.reset_time = reset_time;
    }
}

void GPUWorkerPool::GetRateLimitMany(const std::vector<const RateLimitReq*>& reqs, const std::vector<RateLimitReqState>& st,
                                     std::vector<RateLimitResp*>& out) {
    if (reqs.empty()) return;
    if (closed_.load() || shards_.empty()) {
        for (auto* r : out) r->error = "worker pool is closed";
        return;
    }
    Job job{reqs, st, out, {}, {}, {}};
    job.key_len.resize(reqs.size());
    job.per.resize(shards_.size());
    // every request belongs to the shard of its key (workers.go:261-291); requests of one key keep their order
    std::string k;
    for (size_t i = 0; i < reqs.size(); ++i) {
        k.assign(reqs[i]->name); k.push_back('_'); k.append(reqs[i]->unique_key);               // HashKey, client.go:39-41
        job.key_len[i] = (uint32_t)k.size();
        const uint32_t j = ShardOf((const uint8_t*)k.data(), (uint32_t)k.size());
        if (k.size() > max_key_) {                                   // answered here, never reaches the device
            shards_[j]->key_too_long++;
            answer_item(*out[i], *reqs[i], GUBER_OK, GUBER_ITEM_E_KEY_TOO_LONG, 0, 0, 0, 0);
            continue;
        }
        job.per[j].push_back((uint32_t)i);
    }
    for (size_t j = 0; j < job.per.size(); ++j) {
        const std::vector<uint32_t>& list = job.per[j];
        uint32_t begin = 0;
        while (begin < list.size()) {
            Ticket t{};
            const uint32_t got = reserve(job, *shards_[j], list, begin, &t);
            if (!got) { fail_rest(job, list, begin, "worker pool is closed"); break; }
            write_requests(job, t, list);
            job.tickets.push_back(t);
            begin += got;
        }
    }
    for (auto& t : job.tickets)
        if (!t.consumed) try_consume(job, t, true);
}

void GPUWorkerPool::fail_rest(Job& job, const std::vector<uint32_t>& list, uint32_t begin, const char* why) {
    for (size_t q = begin; q < list.size(); ++q) { *job.out[list[q]] = RateLimitResp{}; job.out[list[q]]->error = why; }
}

// Reserve slots for as many of list[begin..] as the shard's open stage still takes: ONE compare-and-swap on the stage's
// reservation word (slots | key bytes << 32).  Returns the number reserved (>= 1), or 0 when the pool has been closed.
uint32_t GPUWorkerPool::reserve(Job& job, Shard& sh, const std::vector<uint32_t>& list, uint32_t begin, Ticket* out) {
    for (;;) {
        const uint32_t k = sh.open.load(std::memory_order_acquire);
        if (k == kStages + 1) return 0;
        if (k < kStages) {
            Stage& s = sh.st[k];
            uint64_t w = s.word.load(std::memory_order_acquire);
            while (!(w & kClosed)) {
                const uint32_t cnt = (uint32_t)w, kb = (uint32_t)(w >> 32);
                const uint32_t room = batch_limit_ - cnt;
                uint32_t take = 0; uint64_t bytes = 0;
                while (take < room && begin + take < list.size()) {
                    const uint32_t kl = job.key_len[list[begin + take]];
                    if ((uint64_t)kb + bytes + kl > key_cap_) break;
                    bytes += kl; ++take;
                }
                if (take == 0) {                                     // no slot or no key bytes left: flush it now, take the next stage
                    if (!s.flush_now.exchange(true)) {
                        if (room) sh.flush_on_key_bytes++;
                        { std::lock_guard<std::mutex> lk(sh.mu); }
                        sh.cv_batcher.notify_one();
                    }
                    break;
                }
                if (s.word.compare_exchange_weak(w, w + take + (bytes << 32), std::memory_order_acq_rel, std::memory_order_acquire)) {
                    *out = Ticket{&sh, &s, s.gen.load(std::memory_order_relaxed), cnt, take, begin, kb, &list, false};
                    const bool first = cnt == 0, full = cnt + take >= batch_limit_;
                    if (first) s.first_us.store(mono_us(), std::memory_order_release);
                    const uint64_t q = (uint64_t)cnt + take;
                    if (q > sh.queue_max.load(std::memory_order_relaxed)) sh.queue_max.store(q, std::memory_order_relaxed);
                    if (first || full) {
                        { std::lock_guard<std::mutex> lk(sh.mu); }
                        sh.cv_batcher.notify_one();
                    }
                    return take;
                }
            }
        }
        // no stage takes reservations right now: pick up responses that are ready (so that stages drain and the batcher can
        // rotate), then wait for the batcher to open the next one
        for (auto& t : job.tickets)
            if (!t.consumed) try_consume(job, t, false);
        std::unique_lock<std::mutex> lk(sh.mu);
        if (sh.open.load(std::memory_order_acquire) == k) wait_us(sh.cv_callers, lk, 50);
    }
}

// the caller writes its requests into the slots it reserved: HashKey = name + "_" + unique_key straight into the key buffer
void GPUWorkerPool::write_requests(Job& job, const Ticket& t, const std::vector<uint32_t>& list) {
    Stage& s = *t.st;
    const guber_batch_t* b = s.b;
    uint8_t* kp = (uint8_t*)b->key_bytes;
    uint32_t* off = (uint32_t*)b->key_off;
    int64_t *hits = (int64_t*)b->hits, *limit = (int64_t*)b->limit, *duration = (int64_t*)b->duration, *burst = (int64_t*)b->burst,
            *created = (int64_t*)b->created_at;
    uint8_t *algo = (uint8_t*)b->algorithm, *owner = (uint8_t*)b->is_owner;
    uint32_t* beh = (uint32_t*)b->behavior;
    int64_t now = 0;
    uint32_t o = t.key_base;
    for (uint32_t q = 0; q < t.count; ++q) {
        const uint32_t ri = list[t.list_begin + q], i = t.first_slot + q;
        const RateLimitReq& r = *job.reqs[ri];
        off[i] = o;
        memcpy(kp + o, r.name.data(), r.name.size()); o += (uint32_t)r.name.size();
        kp[o++] = '_';
        memcpy(kp + o, r.unique_key.data(), r.unique_key.size()); o += (uint32_t)r.unique_key.size();
        hits[i] = r.hits; limit[i] = r.limit; duration[i] = r.duration; burst[i] = r.burst;
        if (r.created_at) created[i] = r.created_at;
        else { if (!now) now = NowMs(); created[i] = now; }
        algo[i] = (r.algorithm == 0 || r.algorithm == 1) ? (uint8_t)r.algorithm : 255;
        beh[i] = r.behavior; owner[i] = job.st[ri].is_owner ? 1 : 0;
        s.name_len[i] = (uint16_t)std::min<size_t>(r.name.size(), 0xffff);
    }
    s.written.fetch_add(t.count, std::memory_order_release);
}

// the responses of a ticket, once its generation has been announced; returns false when not ready and !block
bool GPUWorkerPool::try_consume(Job& job, Ticket& t, bool block) {
    Stage& s = *t.st;
    // (a stage carries generation g + 1 only after every ticket of g has been consumed, so the word reads g - 1 or g here)
    for (uint32_t v; (v = s.done_gen.load(std::memory_order_acquire)) != (uint32_t)t.gen;) {
        if (!block) return false;
        futex_wait(&s.done_gen, v);
    }
    const int rc = s.rc;
    const guber_result_t* r = s.r;
    for (uint32_t q = 0; q < t.count; ++q) {
        const uint32_t ri = (*t.list)[t.list_begin + q], i = t.first_slot + q;
        answer_item(*job.out[ri], *job.reqs[ri], rc, rc == GUBER_OK ? r->err[i] : 0, r->status[i], r->limit[i], r->remaining[i], r->reset_time[i]);
    }
    t.consumed = true;
    s.consumed.fetch_add(t.count, std::memory_order_release);
    return true;
}

// ---- the batcher's side --------------------------------------------------------------------------------------------------
void GPUWorkerPool::open_stage(Shard& sh, uint32_t k) {
    Stage& s = sh.st[k];
    s.n = 0; s.rc = GUBER_OK;
    s.written.store(0); s.consumed.store(0); s.first_us.store(0); s.flush_now.store(false);
    s.gen.store(s.gen.load() + 1);
    s.word.store(0, std::memory_order_release);
    sh.open.store(k, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(sh.mu); }
    sh.cv_callers.notify_all();
}

void GPUWorkerPool::run(Shard& sh) {
    uint32_t cur = 0;
    int inflight = -1;                                               // stage submitted and not yet announced
    open_stage(sh, cur);
    for (;;) {
        Stage& s = sh.st[cur];
        bool due = false, closing = false;
        uint32_t seen_cnt = 0; int64_t seen_us = 0;                 // idle flush: when the reserved count last changed
        {
            // flush at batch_limit, at batch_wait after the first reservation (peer_client.go:284-337), or when a caller found
            // no room; with nothing due, deliver the batch in flight instead of sitting on it
            std::unique_lock<std::mutex> lk(sh.mu);
            for (;;) {
                closing = sh.closing;
                const uint32_t cnt = (uint32_t)s.word.load(std::memory_order_acquire);
                if (cnt >= batch_limit_ || (cnt && (closing || s.flush_now.load()))) { due = true; break; }
                if (cnt) {
                    const int64_t first = s.first_us.load(std::memory_order_acquire), now = mono_us();
                    if (first && now - first >= (int64_t)batch_wait_us_) { due = true; break; }
                    int64_t left = first ? (int64_t)batch_wait_us_ - (now - first) : (int64_t)batch_wait_us_;
                    if (idle_us_) {
                        // optional: nobody has reserved anything for idle_us and every reserved slot is written — the callers are
                        // all waiting for this batch, so holding it until batch_wait only adds latency
                        if (cnt != seen_cnt) { seen_cnt = cnt; seen_us = now; }
                        else if (now - seen_us >= (int64_t)idle_us_ && s.written.load(std::memory_order_acquire) == cnt) { due = true; break; }
                        left = std::min<int64_t>(left, (int64_t)idle_us_ - (now - seen_us));
                    }
                    if (inflight >= 0) break;
                    wait_us(sh.cv_batcher, lk, std::max<int64_t>(left, 1));
                    continue;
                }
                if (inflight >= 0 || closing) break;
                sh.cv_batcher.wait(lk);
            }
        }
        if (!due) {
            if (inflight >= 0) { complete(sh, sh.st[inflight]); inflight = -1; continue; }
            if (closing) {
                // nothing reserved, nothing in flight: stop taking reservations; a caller may have slipped one in meanwhile
                sh.open.store(kStages + 1, std::memory_order_release);
                const uint64_t w = s.word.fetch_or(kClosed, std::memory_order_acq_rel);
                if ((uint32_t)w == 0) break;
                s.n = (uint32_t)w;
                while (s.written.load(std::memory_order_acquire) != s.n) std::this_thread::yield();
                ((uint32_t*)s.b->key_off)[s.n] = (uint32_t)(w >> 32);
                submit(sh, s);
                complete(sh, s);
                break;
            }
            continue;
        }
        // the next stage takes the reservations from here on; it was announced two flushes ago and has been read out since
        const uint32_t next = (cur + 1) % kStages;
        Stage& nx = sh.st[next];
        if (inflight == (int)next) { complete(sh, nx); inflight = -1; }
        for (uint32_t spins = 0; nx.gen.load() && nx.consumed.load(std::memory_order_acquire) != nx.n; ++spins) {
            if (spins < 64) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        open_stage(sh, next);
        const uint64_t w = s.word.fetch_or(kClosed, std::memory_order_acq_rel);
        s.n = (uint32_t)w;
        while (s.written.load(std::memory_order_acquire) != s.n) std::this_thread::yield();   // callers still copying their requests in
        ((uint32_t*)s.b->key_off)[s.n] = (uint32_t)(w >> 32);
        submit(sh, s);
        if (inflight >= 0) complete(sh, sh.st[inflight]);
        inflight = (int)cur;
        cur = next;
    }
    if (inflight >= 0) complete(sh, sh.st[inflight]);
    sh.open.store(kStages + 1, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(sh.mu); }
    sh.cv_callers.notify_all();
}

// hand a sealed stage to the engine (asynchronous).  With a persistent Store configured the batch takes the synchronous
// path that makes the Store's calls.
void GPUWorkerPool::submit(Shard& sh, Stage& s) {
    s.t0_us = mono_us();
    sh.requests += s.n;
    if (s.n > sh.batch_max.load()) sh.batch_max.store(s.n);
    guber_batch_t* b = s.b;
    b->n = s.n; b->now_ms = NowMs();                                  // DURATION_IS_GREGORIAN: the kernels derive the interval from now_ms
    sh.in_flight++;
    if (has_store_.load()) { submit_with_store(sh, s); return; }
    s.rc = guber_stage_submit(s.stage);
    s.submitted = s.rc == GUBER_OK;
}

// wait for a submitted stage and announce its generation: the callers read their responses themselves
void GPUWorkerPool::complete(Shard& sh, Stage& s) {
    if (s.submitted) { s.rc = guber_stage_wait(s.stage); s.submitted = false; }
    const uint64_t us = (uint64_t)std::max<int64_t>(mono_us() - s.t0_us, 0);
    sh.send_us_sum += us;
    if (us > sh.send_us_max.load()) sh.send_us_max.store(us);
    sh.flushed++; sh.in_flight--;
    s.done_gen.store((uint32_t)s.gen.load(), std::memory_order_release);
    futex_wake_all(&s.done_gen);
}

// Config.Store (store.go:49-65) configured: ask the store for keys that are not resident BEFORE the batch
// (algorithms.go:45-51 `s.Get` on a cache miss, then `c.Add(item)`), evaluate, then issue the Remove / OnChange
// calls the reference makes from inside the algorithms, in request order.  Synchronous, host-pointer entry points over
// the stage's own arrays.
void GPUWorkerPool::submit_with_store(Shard& sh, Stage& s) {
    guber_engine_t* const engine_ = sh.engine;
    const uint32_t n = s.n;
    guber_batch_t b = *s.b;
    guber_result_t res = *s.r;
    const uint8_t* keys = b.key_bytes; const uint32_t* off = b.key_off;
    std::vector<uint8_t> sflags(n, 0); std::vector<guber_item_t> sitems(n);
    guber_store_events_t sev{sflags.data(), sitems.data()};
    auto store_req = [&](uint32_t i) {
        guber_store_req_t q{};
        q.key = keys + off[i]; q.key_len = off[i + 1] - off[i]; q.name_len = s.name_len[i];
        q.hits = b.hits[i]; q.limit = b.limit[i]; q.duration = b.duration[i]; q.burst = b.burst[i]; q.created_at = b.created_at[i];
        q.algorithm = b.algorithm[i] == 255 ? -1 : b.algorithm[i]; q.behavior = b.behavior[i];
        return q;
    };
    std::vector<uint8_t> missing(n, 0);
    int rc = guber_probe_missing(engine_, &b, missing.data());
    if (rc == GUBER_OK && store_.get) {
        std::vector<std::string> asked;
        for (uint32_t i = 0; i < n && rc == GUBER_OK; ++i) {
            if (!missing[i] || off[i + 1] == off[i]) continue;
            std::string k((const char*)keys + off[i], off[i + 1] - off[i]);
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
            if ((sflags[i] & GUBER_STORE_REMOVE) && store_.remove) store_.remove(store_.user, keys + off[i], off[i + 1] - off[i]);
            if ((sflags[i] & GUBER_STORE_ONCHANGE) && store_.on_change) { const guber_store_req_t q = store_req(i); store_.on_change(store_.user, &q, &sitems[i]); }
        }
    }
    s.rc = rc; s.submitted = false;
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
