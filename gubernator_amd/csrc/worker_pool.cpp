// worker_pool.cpp — see worker_pool.h.
#include "worker_pool.h"

#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "guber_algo.h"
#include "guber_host.h"
#include "guber_placement_impl.h"

// -DGUBER_POOL_TRACE (test builds only: tests/hostsim/engine_stub.cpp keeps the ring and prints it on a mismatch): who reserved,
// sealed, submitted, moved what, in which order.  Nothing in the product build.
#ifdef GUBER_POOL_TRACE
extern "C" void guber_pool_trace(const char* what, const void* obj, uint64_t a, uint64_t b, uint64_t c);
#define PT(what, obj, a, b, c) guber_pool_trace(what, (const void*)(obj), (uint64_t)(a), (uint64_t)(b), (uint64_t)(c))
#else
#define PT(what, obj, a, b, c) do {} while (0)
#endif
// -DGUBER_POOL_TEST_HOOKS (tests/hostsim/pool_test.cpp): the test is told where a caller is, so that it can make a placement pass
// happen exactly there (in the middle of a routing round) instead of waiting for the scheduler to arrange it
#ifdef GUBER_POOL_TEST_HOOKS
extern "C" void guber_pool_test_hook(int where, uint32_t i, uint32_t n);
#define HOOK(where, i, n) guber_pool_test_hook(where, i, n)
#else
#define HOOK(where, i, n) do {} while (0)
#endif

namespace gubernator {

// Callers wait for their generation — and the dispatcher for work — on plain futex words: an announcement wakes exactly the
// threads sleeping on that word, each of which re-reads it and goes on; no mutex to queue up behind.
static void futex_wait(std::atomic<uint32_t>* w, uint32_t seen, int64_t timeout_us = -1) {
    if (timeout_us < 0) { syscall(SYS_futex, (uint32_t*)w, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); return; }
    struct timespec ts; ts.tv_sec = timeout_us / 1000000; ts.tv_nsec = (timeout_us % 1000000) * 1000;
    syscall(SYS_futex, (uint32_t*)w, FUTEX_WAIT_PRIVATE, seen, &ts, nullptr, 0);
}
static void futex_wake_all(std::atomic<uint32_t>* w) { syscall(SYS_futex, (uint32_t*)w, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }
static void futex_wake_n(std::atomic<uint32_t>* w, int n) { syscall(SYS_futex, (uint32_t*)w, FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0); }

static int64_t mono_us() {
    using namespace std::chrono;
    return duration_cast<microseconds>(steady_clock::now().time_since_epoch()).count();
}
static inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}
static uint32_t env_u32(const char* name, uint32_t dflt) { const char* v = getenv(name); return v ? (uint32_t)strtoul(v, nullptr, 10) : dflt; }
// (the experiments' knobs: read by -DGUBER_LAB builds only, guber_host.h)
#define lab_u32(name, dflt) ([&]() -> uint32_t { const char* v_ = guber_lab_env(name); return v_ ? (uint32_t)strtoul(v_, nullptr, 10) : (uint32_t)(dflt); }())

GPUWorkerPool::GPUWorkerPool(const guber_config_t& cfg, uint32_t batch_limit, uint32_t batch_wait_us, uint32_t shards,
                             const std::vector<int32_t>& devices)
    : batch_limit_(batch_limit ? batch_limit : 1000), batch_wait_us_(batch_wait_us ? batch_wait_us : 500) {
    idle_us_ = lab_u32("GUBER_POOL_IDLE_US", 0);                     // optional extra trigger: nobody reserved for this long and all slots written
    depth_ = std::max(1u, std::min(env_u32("GUBER_POOL_DEPTH", 2), kStages - 2));   // batches of one shard on the GPU at a time
    eager_ = env_u32("GUBER_POOL_EAGER", 1) != 0;                    // 0 = the reference's peer batcher policy alone: limit or wait
    eager_min_ = lab_u32("GUBER_POOL_EAGER_MIN", 4096);
    direct_max_ = env_u32("GUBER_POOL_DIRECT_MAX", 4);               // RPCs of at most this many requests may be evaluated by their caller (0 = never)
    direct_callers_ = lab_u32("GUBER_POOL_DIRECT_CALLERS", 0);       // ... while at most this many calls are in progress (0 = half the shards, at least 2)
    one_pass_ = lab_u32("GUBER_POOL_ONE_PASS", 1) != 0;              // one shard on one device: reserve first, then touch every request once
    nt_stores_ = lab_u32("GUBER_POOL_NT_STORES", 1) != 0;            // the 8-byte request columns go into the stage with non-temporal stores
    spin_us_ = env_u32("GUBER_POOL_SPIN_US", 40);                    // how long a waiting caller looks before it sleeps
    {   // callers allowed in the CPU part of a call at a time: the CPUs this process may really use (a cgroup CPU quota counts),
        // minus one for the dispatcher.  More runnable callers than CPUs only get the whole group throttled.
        uint32_t cpus = std::max(1u, std::thread::hardware_concurrency());
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0}; unsigned long long period = 0;
            if (fscanf(f, "%31s %llu", q, &period) == 2 && period && strcmp(q, "max") != 0) {
                const unsigned long long quota = strtoull(q, nullptr, 10);
                if (quota) cpus = std::min<uint32_t>(cpus, (uint32_t)((quota + period - 1) / period));
            }
            fclose(f);
        }
        max_active_ = env_u32("GUBER_POOL_MAX_ACTIVE", cpus > 4 ? cpus - 2 - cpus / 8 : cpus);
        limit_active_ = max_active_ != 0;
        max_spinners_ = std::max(1u, std::min(cpus / 2, 16u));
        if (limit_active_) {
            turn_.reset(new std::atomic<uint32_t>[kTurns]);
            for (uint32_t k = 0; k < kTurns; ++k) turn_[k].store(0, std::memory_order_relaxed);
            granted_.store(max_active_, std::memory_order_relaxed);                 // the first max_active_ tickets walk in
        }
    }
    rebalance_ms_ = env_u32("GUBER_POOL_REBALANCE_MS", 250);         // 0 = placement stays the reference's worker rule
    if (shards == 0) shards = 1;
    if (shards > 1024) shards = 1024;
    std::vector<int32_t> devs = devices;
    if (devs.empty()) devs.push_back(cfg.device);
    has_global_ = (cfg.flags & GUBER_FLAG_GLOBAL) != 0;
    if (direct_callers_ == 0) direct_callers_ = std::max<uint32_t>(2u, (uint32_t)devs.size() * shards / 2);
    n_devices_ = (uint32_t)devs.size(); plain_per_device_ = shards; shards_per_device_ = shards + (has_global_ ? 1 : 0);
    if (n_devices_ > 1 || has_global_) {                             // the GPUs of the node are the peers of the ring
        std::vector<std::string> names; std::vector<const char*> ptrs;
        for (uint32_t i = 0; i < n_devices_; ++i) names.push_back("gpu" + std::to_string(i));
        for (auto& n : names) ptrs.push_back(n.c_str());
        create_rc_ = guber_ring_create(ptrs.data(), n_devices_, 512, 0, &ring_);
        if (create_rc_ != GUBER_OK) return;
    }
    // Several shards per device: the callers of a device share ONE set of stages (its front) and tag every request with its
    // shard and its place in that shard's share; the copy kernel that brings a batch to HBM hands the requests to the shards
    // (guber_stage_submit_routed).  What a caller pays per RPC — one compare-and-swap, contiguous writes, one sleep — then
    // does not grow with the number of shards.  GUBER_POOL_ROUTED=0: every shard has stages of its own and the callers sort
    // their requests by shard (the arrangement this replaced; also what a build without the routed entry point would do).
    routed_ = lab_u32("GUBER_POOL_ROUTED", 1) != 0 && shards_per_device_ >= 2 && shards_per_device_ <= kMaxEngines;
    stage_cap_ = routed_ ? (uint32_t)std::min<uint64_t>((uint64_t)batch_limit_ * shards, 65536) : batch_limit_;
    if (routed_ && batch_limit_ > 65536) { routed_ = false; stage_cap_ = batch_limit_; }     // (a share may be the whole stage: the two-launch pipeline takes 65 536)
    // a second generation behind the one in flight: per-shard stages pay a set of launches per group of four shards, so it waits
    // until it is worth them; a front stage pays ONE launch for a handful of requests (k_small_routed) and four beyond
    // GUBER_POOL_DEVROUTE=1: the device also ROUTES (XXH64 of the HashKey, slot table, hot-key list, rank in the shard's share:
    // guber_stage_route) and a caller's per-request work is writing the request — 62 instead of 130 cycles per request.  Off by
    // default: the shares' sizes have to come back to the host before the batch can be enqueued, and that round trip (two launches,
    // a flag over PCIe, then the submission: +45 us on a 2 000-request batch, +70..90 us on a 20 000-request one on MI355X) costs a
    // closed loop of callers more than their CPUs gain (64 x 1000-item RPCs, 8 shards: 218 instead of 236 M decisions/s).  It pays
    // where the callers' CPUs are the scarce resource and latency is not.
    dev_route_ = routed_ && lab_u32("GUBER_POOL_DEVROUTE", 0) != 0;
    if (!guber_lab_env("GUBER_POOL_EAGER_MIN")) eager_min_ = routed_ ? 16 : 4096;
    guber_config_t c = cfg;
    if (c.max_batch < stage_cap_) c.max_batch = stage_cap_;
    const uint32_t total = n_devices_ * shards;
    const uint64_t per_shard = cfg.cache_size / total;   // workers.go:132 `CacheSize / Workers` per worker (0 -> the engine takes NewLRUCache's default of 50 000, lrucache.go:62)
    max_key_ = c.max_key_bytes ? c.max_key_bytes : 1024;
    // room for batch_limit keys of typical size; a batch whose keys do not fit is flushed early (never overrun)
    key_cap_ = (uint32_t)std::min<uint64_t>((uint64_t)stage_cap_ * std::min<uint32_t>(max_key_, 96u) + max_key_, (1u << 24) - 1);
    // the shards of a device are spread over a few streams; shards that share one share their launches (guber_stages_submit)
    uint32_t n_streams = lab_u32("GUBER_POOL_STREAMS", 0);
    if (n_streams == 0) n_streams = (shards + 3) / 4;                // a fused launch carries the batches of up to four shards
    n_streams = std::max(1u, std::min(n_streams, shards));
    if (routed_) n_streams = 1;                                      // (the shares of a front stage travel in one pair of launches)
    for (uint32_t d = 0; d < n_devices_ && create_rc_ == GUBER_OK; ++d) {
        std::unique_ptr<Device> dev(new Device());
        dev->index = d; dev->ordinal = devs[d]; dev->n_plain = shards;
        if (shards > 1) {
            create_rc_ = guber_placement_create(shards, 0, &dev->place);
            if (create_rc_ != GUBER_OK) { devs_.push_back(std::move(dev)); break; }
        }
        std::vector<void*> stream_of(n_streams, nullptr);
        for (uint32_t i = 0; i < shards_per_device_; ++i) {
            std::unique_ptr<Shard> sh(new Shard());
            const bool global = i >= shards;
            const uint32_t sidx = global ? n_streams - 1 : (uint32_t)((uint64_t)i * n_streams / shards);
            c.device = devs[d];
            c.flags = global ? cfg.flags : (cfg.flags & ~(uint32_t)GUBER_FLAG_GLOBAL);
            c.cache_size = global ? std::max<uint64_t>(cfg.cache_size / n_devices_, 1) : per_shard;
            c.stream = stream_of[sidx];
            sh->device = devs[d]; sh->global = global; sh->dev = dev.get();
            create_rc_ = guber_engine_create(&c, &sh->engine);
            if (create_rc_ != GUBER_OK) { sh->engine = nullptr; break; }
            if (!stream_of[sidx]) { stream_of[sidx] = guber_engine_stream(sh->engine); sh->owns_stream = true; }
            if (!routed_) create_rc_ = create_stages(*sh);
            dev->shards.push_back(sh.get());
            shards_.push_back(std::move(sh));
            if (create_rc_ != GUBER_OK) break;
        }
        if (routed_ && create_rc_ == GUBER_OK) {
            dev->front.reset(new Shard());
            Shard& f = *dev->front;
            f.front = true; f.engine = dev->shards[0]->engine; f.device = devs[d]; f.dev = dev.get();
            create_rc_ = create_stages(f);
            for (auto& st : f.st) if (st.stage) st.dest = guber_stage_dest(st.stage);
            dev->staging.push_back(&f);
        } else {
            dev->staging = dev->shards;
        }
        devs_.push_back(std::move(dev));
    }
    if (create_rc_ != GUBER_OK) {
        destroy_engines();
        shards_.clear();
        return;
    }
    for (auto& d : devs_) for (Shard* sh : d->staging) staging_.push_back(sh);
    for (auto& d : devs_) { Device* p = d.get(); p->thread = std::thread([this, p] { run(*p); }); }
}

int GPUWorkerPool::create_stages(Shard& sh) {
    for (uint32_t k = 0; k < kStages; ++k) {
        Stage& st = sh.st[k];
        const int rc = guber_stage_create(sh.engine, stage_cap_, key_cap_, &st.stage);
        if (rc != GUBER_OK) return rc;
        st.b = guber_stage_batch(st.stage); st.r = guber_stage_result(st.stage);
        st.shard = &sh;
        st.name_len.assign(stage_cap_, 0);
    }
    return GUBER_OK;
}

void GPUWorkerPool::destroy_engines() {
    for (auto& sh : shards_) for (auto& st : sh->st) if (st.stage) { guber_stage_destroy(st.stage); st.stage = nullptr; }
    for (auto& d : devs_) if (d->front) for (auto& st : d->front->st) if (st.stage) { guber_stage_destroy(st.stage); st.stage = nullptr; }
    for (int pass = 0; pass < 2; ++pass)                             // the engines that own a stream go last
        for (auto& sh : shards_)
            if (sh->engine && (pass == 1 || !sh->owns_stream)) { guber_engine_destroy(sh->engine); sh->engine = nullptr; }
    // (the Shard objects and the placements stay until the destructor: a caller that raced Close() may still look at them)
}

GPUWorkerPool::~GPUWorkerPool() {
    Close();
    for (auto& d : devs_) if (d->place) { guber_placement_destroy(d->place); d->place = nullptr; }
    if (ring_) guber_ring_destroy(ring_);
}

// at most max_active_ callers are in the CPU part of a call at a time; the others wait their turn asleep (a counting semaphore:
// one sleeper is woken per slot that frees up — no herd)
void GPUWorkerPool::enter() {
    if (!limit_active_) return;
    const uint64_t t = next_ticket_.fetch_add(1, std::memory_order_seq_cst);
    std::atomic<uint32_t>* w = &turn_[t % kTurns];            // the word this ticket sleeps on: it changes whenever a ticket that maps to it is granted
    for (uint32_t spins = 0;; ++spins) {
        const uint32_t v = w->load(std::memory_order_seq_cst);
        if (t < granted_.load(std::memory_order_seq_cst)) return;
        if (spins < 64) { cpu_relax(); continue; }
        futex_wait(w, v, 2000);                               // (woken by the grant — the word has changed by then, or changes before the sleep starts; the bound is a belt)
    }
}
void GPUWorkerPool::leave() {
    if (!limit_active_) return;
    const uint64_t t = granted_.fetch_add(1, std::memory_order_seq_cst);       // the ticket that may enter now
    std::atomic<uint32_t>* w = &turn_[t % kTurns];
    w->store((uint32_t)(t + 1), std::memory_order_seq_cst);
    if (t < next_ticket_.load(std::memory_order_seq_cst)) futex_wake_all(w);    // (its holder is waiting or about to look; tickets kTurns apart share the word and look again)
}

void GPUWorkerPool::wake(Device& d) {
    d.wake.fetch_add(1, std::memory_order_seq_cst);
    if (d.sleeping.load(std::memory_order_seq_cst)) futex_wake_all(&d.wake);
}

void GPUWorkerPool::Close() {
    if (closed_.exchange(true)) return;
    for (auto& d : devs_) { d->closing.store(true); wake(*d); }
    for (auto& d : devs_) if (d->thread.joinable()) d->thread.join();   // every reserved request has been evaluated and announced
    {
        std::lock_guard<std::mutex> lk(comm_mu_);
        if (comm_) { guber_comm_destroy(comm_); comm_ = nullptr; }
    }
    for (Shard* sh : staging_)
        for (auto& st : sh->st)                                      // callers may still be copying their responses out of the stage
            while (st.gen.load() && st.state != Stage::kFree && st.state != Stage::kOpen && st.consumed.load(std::memory_order_acquire) != st.n) std::this_thread::yield();
    destroy_engines();
}

uint32_t GPUWorkerPool::DeviceOf(const uint8_t* key, uint32_t len) const {
    if (n_devices_ <= 1 || !ring_) return 0;
    const uint32_t off[2] = {0, len};
    uint32_t owner = 0;
    guber_ring_route(ring_, key, off, 1, &owner);                                           // replicated_hash.go:104-119
    return owner < n_devices_ ? owner : 0;
}
uint32_t GPUWorkerPool::route(const Device& d, uint64_t h, uint32_t behavior) const {
    if (has_global_ && (behavior & 2u)) return d.n_plain;                                   // Behavior_GLOBAL: the device's GLOBAL engine
    return d.place ? guber_placement_shard_inl(d.place, h) : 0;                             // workers.go:180-184 getWorker, generalised
}
uint32_t GPUWorkerPool::ShardOf(const uint8_t* key, uint32_t len, uint32_t behavior) const {
    if (shards_.size() <= 1) return 0;
    const uint32_t dv = DeviceOf(key, len);
    return dv * shards_per_device_ + route(*devs_[dv], guber_xxhash64(key, len, 0), behavior);   // workers.go:153-155 ComputeHash63
}
uint64_t GPUWorkerPool::batches_flushed() const {
    uint64_t n = 0;
    for (auto& sh : shards_) n += sh->flushed.load();
    for (auto& d : devs_) if (d->front) n += d->front->flushed.load();
    return n;
}
void GPUWorkerPool::Metrics(guber_pool_metrics_t* out) const {
    memset(out, 0, sizeof(*out));
    std::vector<const Shard*> counted;                               // the shards (their direct batches, key errors) and the devices' fronts (the staged batches)
    for (auto& sh : shards_) counted.push_back(sh.get());
    for (auto& d : devs_) if (d->front) counted.push_back(d->front.get());
    for (const Shard* sh : counted) {
        out->batches += sh->flushed.load(); out->requests += sh->requests.load();
        for (auto& st : sh->st) { const uint64_t w = st.word.load(); if (!(w & kClosed)) out->queue_length += word_count(w); }
        out->queue_length_max = std::max<uint64_t>(out->queue_length_max, sh->queue_max.load());
        out->send_duration_us_sum += sh->send_us_sum.load();
        out->send_duration_us_max = std::max<uint64_t>(out->send_duration_us_max, sh->send_us_max.load());
        out->batch_size_max = std::max<uint64_t>(out->batch_size_max, sh->batch_max.load());
        out->in_flight += sh->in_flight.load();
        out->key_too_long += sh->key_too_long.load(); out->flush_on_key_bytes += sh->flush_on_key_bytes.load();
    }
    for (const Shard* sh : counted) out->direct_batches += sh->direct.load();
    for (auto& d : devs_) { out->rebalances += d->rebalances.load(); out->keys_moved += d->moves.load(); out->submit_us_sum += d->submit_us.load(); out->submits += d->submits.load(); }
    out->shards = (uint32_t)shards_.size(); out->devices = n_devices_;
    if (guber_lab_env("GUBER_POOL_DEBUG") && d_dbg_[3].load())
        fprintf(stderr, "[pool] per batch: wait for writers %.1f us, submit %.1f us, on the GPU until seen %.1f us (%llu batches); dispatcher loops %llu, polls %llu\n",
                (double)d_dbg_[0] / d_dbg_[3], (double)d_dbg_[1] / d_dbg_[3], (double)d_dbg_[2] / d_dbg_[3], (unsigned long long)d_dbg_[3].load(),
                (unsigned long long)d_dbg_[4].load(), (unsigned long long)d_dbg_[5].load());
}

int64_t GPUWorkerPool::NowMs() const {
    const int64_t f = frozen_ms_.load(std::memory_order_relaxed);
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
// One call = a list of requests from some source (C++ objects, or the structure-of-arrays a binding hands over) whose answers go
// to some sink.  The work per request is a few dozen nanoseconds and all of it is the caller's: HashKey bytes, XXH64, device and
// shard, a share of one compare-and-swap, eight stores into the stage, four loads out of it.
struct ReqRef {
    const uint8_t *name, *ukey; uint32_t name_len, ukey_len;
    int64_t hits, limit, duration, burst, created_at; int32_t algorithm; uint32_t behavior; bool is_owner;
};
// per-thread scratch: nothing is allocated per call once a thread has served a few RPCs
struct GPUWorkerPool::Scratch {
    std::vector<uint8_t> keys;                        // HashKey bytes of every request, back to back
    std::vector<uint32_t> koff, klen;
    std::vector<uint64_t> hash;
    std::vector<uint16_t> dev;
    std::vector<uint8_t> eng;                         // a request's shard inside its device (routed pools: what the front stage's dest column carries)
    std::vector<uint32_t> shard, order, todo, next, count, vers;
    std::vector<Ticket2> tickets;
    std::vector<int64_t> col;                         // a ticket's 8-byte columns, gathered before they are streamed into the stage
    std::vector<uint8_t> dkeys;                       // key bytes of a batch the caller evaluates itself
    uint32_t observe_tick = 0;
};

#ifdef GUBER_POOL_PHASES
static std::atomic<uint64_t> g_ph[6];
static inline uint64_t tsc() { return __builtin_ia32_rdtsc(); }
struct PhasePrinter { ~PhasePrinter() { uint64_t t = 0; for (int k = 0; k < 5; ++k) t += g_ph[k]; if (g_ph[5]) fprintf(stderr, "[phases] per request cycles: build %.1f route+sort %.1f reserve+write %.1f leave %.1f wait+consume %.1f (n=%llu)\n", (double)g_ph[0] / g_ph[5], (double)g_ph[1] / g_ph[5], (double)g_ph[2] / g_ph[5], (double)g_ph[3] / g_ph[5], (double)g_ph[4] / g_ph[5], (unsigned long long)g_ph[5].load()); } } g_phase_printer;
#define PH(k) do { const uint64_t _t = tsc(); g_ph[k] += _t - ph_t; ph_t = _t; } while (0)
#else
#define PH(k) do {} while (0)
#endif
template <class Src, class Sink>
struct GPUWorkerPool::Call {
    GPUWorkerPool& P; const Src& src; Sink& sink; Scratch& S;

    void run() {
        const uint32_t n = src.size();
        if (n == 0) return;
        if (P.closed_.load() || P.shards_.empty()) { for (uint32_t i = 0; i < n; ++i) sink.closed(i); return; }
        struct InCall { std::atomic<uint32_t>& c; InCall(std::atomic<uint32_t>& x) : c(x) { c.fetch_add(1, std::memory_order_relaxed); } ~InCall() { c.fetch_sub(1, std::memory_order_relaxed); } } in_call{P.in_calls_};
        P.enter();
#ifdef GUBER_POOL_PHASES
        uint64_t ph_t = tsc(); g_ph[5] += n;
#endif
        S.koff.resize(n + 1); S.klen.resize(n); S.hash.resize(n); S.dev.resize(n); S.shard.resize(n); S.eng.resize(n); S.todo.resize(n);
        S.order.clear(); S.tickets.clear();
        if (P.one_pass_ && one_pass(n)) {
            PH(2);
            P.leave();
            PH(3);
            for (auto& t : S.tickets)
                if (!t.consumed) consume(t, true);
            PH(4);
            return;
        }
        // HashKey = name + "_" + unique_key (client.go:39-41), its XXH64 (workers.go:153-155), its device (replicated_hash.go:104-119)
        S.keys.resize(src.key_bytes_total() + 16);
        uint8_t* const kb = S.keys.data();
        uint32_t o = 0, nt = 0;
        const bool multi = P.n_devices_ > 1;
        const bool need_route = !P.dev_route_ || n <= P.direct_max_;   // (a handful of requests may be evaluated by their caller: it must know their shard)
        ReqRef r;
        for (uint32_t i = 0; i < n; ++i) {
            src.key(i, r);
            S.koff[i] = o;
            const uint32_t len = r.name_len + 1 + r.ukey_len;
            S.klen[i] = len;
            uint8_t* k = kb + o;
            memcpy(k, r.name, r.name_len); k[r.name_len] = '_'; memcpy(k + r.name_len + 1, r.ukey, r.ukey_len);
            o += len;
            if (r.name_len == 0 || r.ukey_len == 0) { if (src.front_end_checks()) continue; }   // (answered by the front end: empty field)
            const uint32_t dv = multi ? P.DeviceOf(k, len) : 0;
            S.dev[i] = (uint16_t)dv;
            if (len > P.max_key_) {                                  // answered here, never reaches the device
                P.shards_[(size_t)dv * P.shards_per_device_]->key_too_long++;
                sink.item_error(i, GUBER_ITEM_E_KEY_TOO_LONG, src.algorithm(i));
                continue;
            }
            S.todo[nt++] = i;
            Device& d = *P.devs_[dv];
            const bool sample = d.place && (++S.observe_tick & 63u) == 0 && P.rebalance_ms_ && !(P.has_global_ && (src.behavior(i) & 2u));
            if (!need_route && !sample) continue;                    // (the device routes: no hash on the host)
            const uint64_t h = guber::xxhash64(k, len, 0);
            S.hash[i] = h;
            if (sample) guber_placement_observe(d.place, h, 64);     // every 64th request feeds the placement's view of the traffic
        }
        S.koff[n] = o;
        S.todo.resize(nt);
        PH(0);
        // Every request belongs to the shard of its key (workers.go:261-291); requests of one key keep their order.  A round
        // routes what is left with the devices' current placement versions; a reservation refused as stale (the placement
        // changed between routing and reserving) sends the rest of that list into the next round.
        const uint32_t n_shards = (uint32_t)P.staging_.size();       // (routed pools: one per device — the requests stay in arrival order)
        while (!S.todo.empty()) {
            S.vers.assign(P.n_devices_, 0xffffffffu);
            S.count.assign(n_shards + 1, 0);
            [[maybe_unused]] uint32_t routed_so_far = 0;
            for (uint32_t i : S.todo) {
                const uint32_t dv = S.dev[i];
                Device& d = *P.devs_[dv];
                if (S.vers[dv] == 0xffffffffu) S.vers[dv] = d.ver.load(std::memory_order_acquire) & 0x7fu;
                const uint32_t e = need_route ? P.route(d, S.hash[i], src.behavior(i)) : 0;
                const uint32_t j = P.routed_ ? dv : dv * P.shards_per_device_ + e;
                S.shard[i] = j; S.eng[i] = (uint8_t)e;
                S.count[j + 1]++;
                HOOK(1, routed_so_far++, (uint32_t)S.todo.size());
            }
            for (uint32_t j = 0; j < n_shards; ++j) S.count[j + 1] += S.count[j];
            const uint32_t base = (uint32_t)S.order.size();
            S.order.resize(base + S.todo.size());
            if (n_shards == 1) std::copy(S.todo.begin(), S.todo.end(), S.order.begin() + base);   // (one target: arrival order as it is)
            else {
                std::vector<uint32_t>& fill = S.next;                // (scratch: running positions)
                fill.assign(S.count.begin(), S.count.end() - 1);
                for (uint32_t i : S.todo) S.order[base + fill[S.shard[i]]++] = i;
            }
            S.next.clear();
            PH(1);
            // a handful of requests and nobody else at their shard: the caller evaluates them itself, now (the reference's worker
            // takes a request the moment it arrives) — one launch, no hand-off to the dispatcher and back
            // — when the pool is lightly loaded: with many callers at once, requests that share a launch serve more of them per microsecond
            const bool direct_ok = base == 0 && need_route && P.eager_ && S.todo.size() <= P.direct_max_ && !P.has_store_.load(std::memory_order_relaxed) &&
                                   P.in_calls_.load(std::memory_order_relaxed) <= P.direct_callers_;
            bool closed = false;
            for (uint32_t j = 0; j < n_shards; ++j) {
                uint32_t pos = base + S.count[j], left = S.count[j + 1] - S.count[j];
                if (!left) continue;
                Shard& sh = *P.staging_[j];
                if (direct_ok) {
                    Shard* table = &sh;                              // routed pools: the handful of requests must belong to one shard
                    if (P.routed_) {
                        const uint32_t e0 = S.eng[S.order[pos]];
                        table = sh.dev->shards[e0];
                        for (uint32_t q = 1; q < left; ++q) if (S.eng[S.order[pos + q]] != e0) { table = nullptr; break; }
                    }
                    if (table && direct(*table, sh, S.vers[sh.dev->index], pos, left)) continue;
                }
                while (left) {
                    if (closed) { for (uint32_t q = 0; q < left; ++q) sink.closed(S.order[pos + q]); break; }
                    Ticket2 t{};
                    const int got = reserve(sh, S.vers[sh.dev->index], pos, left, &t);
                    if (got == 0) { closed = true; continue; }
                    if (got < 0) { S.next.insert(S.next.end(), S.order.begin() + pos, S.order.begin() + pos + left); break; }
                    write(t);
                    S.tickets.push_back(t);
                    pos += (uint32_t)got; left -= (uint32_t)got;
                }
            }
            PH(2);
            S.todo.swap(S.next);
            // What was refused goes into the next round IN THE CALL'S ORDER.  The lists above are per shard, and a round that was
            // routing while a placement pass published (the version it holds is the old one: everything it reserves from here on is
            // refused) may have put the earlier requests of a moving key on the old shard's list and the later ones on the new
            // shard's; taken list by list, the later ones would come first in the next round and be evaluated first.  (Found under
            // ThreadSanitizer's timing with stages per shard, GUBER_POOL_ROUTED=0: the answers of one RPC's hot key came out
            // permuted, the totals right.  With one front stage per device a device has ONE list, in arrival order.)
            if (S.todo.size() > 1 && n_shards > 1) std::sort(S.todo.begin(), S.todo.end());
            if (closed) { for (uint32_t i : S.todo) sink.closed(i); S.todo.clear(); }
        }
        P.leave();                                                   // (waiting for the answers needs no CPU)
        PH(3);
        for (auto& t : S.tickets)
            if (!t.consumed) consume(t, true);
        PH(4);
    }

    // ONE shard on ONE device (the reference's Workers = 1): nothing about a request has to be known before its slot is
    // reserved — no device to pick, no shard, so no hash on the host at all — and the call reserves first, lengths only, then
    // touches every request ONCE: HashKey bytes straight into the stage's key buffer, the request columns.  With several shards
    // the hashing and the placement lookups would sit between the reservation and the `written` count, i.e. inside the time the
    // dispatcher waits for a sealed stage's writers (measured on the host-only stub: 8 shards 37 instead of 44 M/s; one shard 80
    // instead of 61), so those pools keep hashing before they reserve.  Reserving first also means that an overloaded pool queues
    // its requests INSIDE the stages: with more outstanding than two stages hold, callers end up asleep on a full stage while
    // holding one of the few CPU slots (256 callers: 124 instead of 240 M/s on the GPU box) — then the general path runs, whose
    // callers arrive at the stage with their work done.  false = not applicable.
    bool one_pass(uint32_t n) {
        if (P.staging_.size() != 1 || n <= P.direct_max_) return false;
        Shard& sh = *P.staging_[0];
        Device& d = *sh.dev;
        if ((d.place || sh.front) && !P.dev_route_) return false;    // (several shards: only when the device routes)
        if ((uint64_t)P.in_calls_.load(std::memory_order_relaxed) * n > 2ull * P.stage_cap_) return false;
        ReqRef r;
        uint32_t nt = 0;
        S.order.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            src.key(i, r);
            const uint32_t len = r.name_len + 1 + r.ukey_len;
            S.klen[i] = len;
            if (r.name_len == 0 || r.ukey_len == 0) { if (src.front_end_checks()) continue; }   // (answered by the front end: empty field)
            if (len > P.max_key_) {                                  // answered here, never reaches the device
                P.shards_[0]->key_too_long++;
                sink.item_error(i, GUBER_ITEM_E_KEY_TOO_LONG, src.algorithm(i));
                continue;
            }
            S.order[nt++] = i;
        }
        S.order.resize(nt);
        uint32_t pos = 0, left = nt;
        while (left) {
            const uint32_t ver = d.ver.load(std::memory_order_acquire) & 0x7fu;
            Ticket2 t{};
            const int got = reserve(sh, ver, pos, left, &t);
            if (got == 0) { for (uint32_t q = 0; q < left; ++q) sink.closed(S.order[pos + q]); break; }
            if (got < 0) continue;                                   // (no placement, no versions: cannot happen)
            write_one_pass(t);
            S.tickets.push_back(t);
            pos += (uint32_t)got; left -= (uint32_t)got;
        }
        return true;
    }
    void write_one_pass(const Ticket2& t) {
        Stage& s = *t.st;
        const guber_batch_t* b = s.b;
        uint8_t* kp = (uint8_t*)b->key_bytes;
        uint32_t* off = (uint32_t*)b->key_off + t.first_slot;
        uint8_t *algo = (uint8_t*)b->algorithm + t.first_slot, *owner = (uint8_t*)b->is_owner + t.first_slot;
        uint32_t* beh = (uint32_t*)b->behavior + t.first_slot;
        uint16_t* nlen = s.name_len.data() + t.first_slot;
        int64_t now = 0;
        uint32_t o = t.key_base;
        const uint32_t n = t.count;
        const uint32_t* list = S.order.data() + t.list_begin;
        if (S.col.size() < (size_t)5 * n) S.col.resize((size_t)5 * n);
        int64_t *c_hits = S.col.data(), *c_limit = c_hits + n, *c_dur = c_limit + n, *c_burst = c_dur + n, *c_created = c_burst + n;
        guber_placement_t* const place = P.rebalance_ms_ ? s.shard->dev->place : nullptr;
        ReqRef r;
        for (uint32_t q = 0; q < n; ++q) {
            src.get(list[q], r);
            uint8_t* k = kp + o;                                     // HashKey = name + "_" + unique_key (client.go:39-41), exactly its bytes
            memcpy(k, r.name, r.name_len); k[r.name_len] = '_'; memcpy(k + r.name_len + 1, r.ukey, r.ukey_len);
            off[q] = o; o += r.name_len + 1 + r.ukey_len;
            if (place && (++S.observe_tick & 63u) == 0 && !(P.has_global_ && (r.behavior & 2u)))
                guber_placement_observe(place, guber::xxhash64(k, r.name_len + 1 + r.ukey_len, 0), 64);   // every 64th request feeds the placement's view of the traffic
            c_hits[q] = r.hits; c_limit[q] = r.limit; c_dur[q] = r.duration; c_burst[q] = r.burst;
            if (r.created_at) c_created[q] = r.created_at;
            else { if (!now) now = P.NowMs(); c_created[q] = now; }
            algo[q] = (r.algorithm == 0 || r.algorithm == 1) ? (uint8_t)r.algorithm : 255;
            beh[q] = r.behavior; owner[q] = r.is_owner ? 1 : 0;
            nlen[q] = (uint16_t)std::min<uint32_t>(r.name_len, 0xffff);
        }
        stream64((int64_t*)b->hits + t.first_slot, c_hits, n); stream64((int64_t*)b->limit + t.first_slot, c_limit, n);
        stream64((int64_t*)b->duration + t.first_slot, c_dur, n); stream64((int64_t*)b->burst + t.first_slot, c_burst, n);
        stream64((int64_t*)b->created_at + t.first_slot, c_created, n);
#if defined(__x86_64__)
        __builtin_ia32_sfence();
#endif
        s.written.fetch_add(n, std::memory_order_release);
    }

    // Evaluate order[pos .. pos + n) of one shard on the caller's thread (guber_eval_batch: the engine's one-launch path, polled).
    // Only when nobody else does the same at this shard and the placement is still the one the caller routed with; a move of a hot
    // key waits for it (place_mu).  Requests of one key from different callers have no order among each other; a caller's own
    // earlier calls have been answered.  false = not taken: the staged path applies.
    bool direct(Shard& sh, Shard& staging, uint32_t ver, uint32_t pos, uint32_t n) {
        constexpr uint32_t kMax = 16;
        if (n > kMax || sh.direct_busy.exchange(true, std::memory_order_acquire)) return false;
        struct Release { std::atomic<bool>& f; ~Release() { f.store(false, std::memory_order_release); } } release{sh.direct_busy};
        Device& d = *sh.dev;
        std::shared_lock<std::shared_mutex> lk(d.place_mu);
        if ((d.ver.load(std::memory_order_acquire) & 0x7fu) != ver || staging.open.load(std::memory_order_acquire) == kOpenDead) return false;
        const uint32_t* list = S.order.data() + pos;
        uint32_t off[kMax + 1], beh[kMax]; int64_t hits[kMax], limit[kMax], duration[kMax], burst[kMax], created[kMax], ol[kMax], orem[kMax], ors[kMax];
        uint8_t algo[kMax], owner[kMax], ost[kMax], oerr[kMax];
        uint32_t bytes = 0;
        for (uint32_t q = 0; q < n; ++q) bytes += S.klen[list[q]];
        if (S.dkeys.size() < (size_t)bytes + 16) S.dkeys.resize((size_t)bytes + 16);
        const int64_t now = P.NowMs();
        ReqRef r;
        uint32_t o = 0;
        for (uint32_t q = 0; q < n; ++q) {
            const uint32_t ri = list[q];
            src.get(ri, r);
            off[q] = o; memcpy(S.dkeys.data() + o, S.keys.data() + S.koff[ri], S.klen[ri]); o += S.klen[ri];
            hits[q] = r.hits; limit[q] = r.limit; duration[q] = r.duration; burst[q] = r.burst; created[q] = r.created_at ? r.created_at : now;
            algo[q] = (r.algorithm == 0 || r.algorithm == 1) ? (uint8_t)r.algorithm : 255; beh[q] = r.behavior; owner[q] = r.is_owner ? 1 : 0;
        }
        off[n] = o; memset(S.dkeys.data() + o, 0, 16);
        guber_batch_t b{}; guber_result_t res{};
        b.n = n; b.key_bytes = S.dkeys.data(); b.key_off = off; b.hits = hits; b.limit = limit; b.duration = duration; b.burst = burst; b.created_at = created;
        b.algorithm = algo; b.behavior = beh; b.is_owner = owner; b.now_ms = now;
        res.status = ost; res.limit = ol; res.remaining = orem; res.reset_time = ors; res.err = oerr;
        const int rc = guber_eval_batch(sh.engine, &b, &res);
        for (uint32_t q = 0; q < n; ++q) {
            const uint32_t ri = list[q];
            if (rc != GUBER_OK) sink.engine_error(ri, rc);
            else if (oerr[q] == 0) sink.ok(ri, ost[q], ol[q], orem[q], ors[q]);
            else sink.item_error(ri, oerr[q], src.algorithm(ri));
        }
        sh.requests += n; sh.flushed++; sh.direct++;
        return true;
    }

    // Reserve slots for as many of order[pos .. pos + count) as the shard's open stage still takes: ONE compare-and-swap on the
    // stage's reservation word (slots | key bytes << 32 | placement version << 56).  Returns the number reserved (>= 1), 0 when
    // the pool has been closed, -1 when the device's placement is no longer the one the caller routed with.
    int reserve(Shard& sh, uint32_t ver, uint32_t pos, uint32_t count, Ticket2* out) {
        Device& d = *sh.dev;
        for (;;) {
            const uint32_t seq = sh.open_seq.load(std::memory_order_acquire);
            const uint32_t k = sh.open.load(std::memory_order_acquire);
            if (k == kOpenDead) return 0;
            if ((d.ver.load(std::memory_order_acquire) & 0x7fu) != ver) { PT("refused", &sh, ver, d.ver.load(), count); return -1; }
            if (k < kStages) {
                Stage& s = sh.st[k];
                uint64_t w = s.word.load(std::memory_order_acquire);
                while (!(w & kClosed)) {
                    if (word_ver(w) != ver) { PT("refused_word", &s, ver, word_ver(w), count); return -1; }
                    const uint32_t cnt = word_count(w), kbytes = word_bytes(w);
                    const uint32_t room = P.stage_cap_ - cnt;
                    uint32_t take = std::min(room, count); uint64_t bytes = 0;
                    const uint32_t* list = S.order.data() + pos;
                    for (uint32_t q = 0; q < take; ++q) bytes += S.klen[list[q]];
                    if ((uint64_t)kbytes + bytes > P.key_cap_) {          // rare: the stage's key buffer is the limit
                        take = 0; bytes = 0;
                        while (take < room && take < count) {
                            const uint32_t kl = S.klen[list[take]];
                            if ((uint64_t)kbytes + bytes + kl > P.key_cap_) break;
                            bytes += kl; ++take;
                        }
                    }
                    if (take == 0) {                                 // no slot or no key bytes left: flush it now, take the next stage
                        if (!s.flush_now.exchange(true)) {
                            if (room) sh.flush_on_key_bytes++;
                            P.wake(d);
                        }
                        break;
                    }
                    if (s.word.compare_exchange_weak(w, w + take + (bytes << 32), std::memory_order_acq_rel, std::memory_order_acquire)) {
                        *out = Ticket2{&s, s.gen.load(std::memory_order_relaxed), cnt, take, pos, kbytes, false};
                        PT("reserve", &s, out->gen, (uint64_t)cnt << 32 | take, (uint64_t)ver << 32 | S.order[pos]);
                        const bool first = cnt == 0, full = cnt + take >= P.stage_cap_;
                        const int64_t now = mono_us();
                        if (first) s.first_us.store(now, std::memory_order_release);
                        s.last_us.store(now, std::memory_order_release);
                        const uint64_t q = (uint64_t)cnt + take;
                        if (q > sh.queue_max.load(std::memory_order_relaxed)) sh.queue_max.store(q, std::memory_order_relaxed);
                        if (first || full) P.wake(d);
                        return (int)take;
                    }
                }
            }
            // no stage takes reservations right now: pick up responses that are ready (so that stages drain and the dispatcher
            // can rotate), then wait for it to open the next one
            for (auto& t : S.tickets)
                if (!t.consumed) consume(t, false);
            sh.open_waiters.fetch_add(1, std::memory_order_seq_cst);
            if (sh.open_seq.load(std::memory_order_seq_cst) == seq) futex_wait(&sh.open_seq, seq, 200);
            sh.open_waiters.fetch_sub(1, std::memory_order_seq_cst);
        }
    }

    // the caller writes its requests into the slots it reserved: the HashKey bytes straight into the stage's key buffer
    // A ticket's share of a stage is a short run in each of ten columns, and the lines were last touched by other cores or read by
    // the GPU: ordinary stores would first fetch every line for ownership.  The 8-byte columns (40 of a request's 54 bytes) are
    // therefore gathered into per-thread scratch and streamed out column by column with non-temporal stores — no ownership
    // fetch, the data is next read over PCIe anyway; the small columns and the key bytes go through the cache.
    void stream64(int64_t* dst, const int64_t* from, uint32_t n) {
#if defined(__x86_64__)
        if (P.nt_stores_) { for (uint32_t q = 0; q < n; ++q) __builtin_ia32_movnti64((long long*)dst + q, (long long)from[q]); return; }
#endif
        memcpy(dst, from, (size_t)n * 8);
    }
    void write(const Ticket2& t) {
        Stage& s = *t.st;
        const guber_batch_t* b = s.b;
        uint8_t* kp = (uint8_t*)b->key_bytes;
        uint32_t* off = (uint32_t*)b->key_off + t.first_slot;
        uint8_t *algo = (uint8_t*)b->algorithm + t.first_slot, *owner = (uint8_t*)b->is_owner + t.first_slot;
        uint32_t* beh = (uint32_t*)b->behavior + t.first_slot;
        uint16_t* nlen = s.name_len.data() + t.first_slot;
        int64_t now = 0;
        uint32_t o = t.key_base;
        const uint32_t n = t.count;
        const uint32_t* list = S.order.data() + t.list_begin;
        const uint8_t* kb = S.keys.data();
        if (S.col.size() < (size_t)5 * n) S.col.resize((size_t)5 * n);
        int64_t *c_hits = S.col.data(), *c_limit = c_hits + n, *c_dur = c_limit + n, *c_burst = c_dur + n, *c_created = c_burst + n;
        ReqRef r;
        for (uint32_t q = 0; q < n; ++q) {
            const uint32_t ri = list[q];
            src.get(ri, r);
            off[q] = o;
            {   // exactly the key's bytes: the next slot's key may belong to another caller, who may have written it already
                const uint8_t* from = kb + S.koff[ri]; uint8_t* to = kp + o; const uint32_t len = S.klen[ri];
                uint32_t w = 0;
                for (; w + 8 <= len; w += 8) { uint64_t v; memcpy(&v, from + w, 8); memcpy(to + w, &v, 8); }
                for (; w < len; ++w) to[w] = from[w];
                o += len;
            }
            c_hits[q] = r.hits; c_limit[q] = r.limit; c_dur[q] = r.duration; c_burst[q] = r.burst;
            if (r.created_at) c_created[q] = r.created_at;
            else { if (!now) now = P.NowMs(); c_created[q] = now; }
            algo[q] = (r.algorithm == 0 || r.algorithm == 1) ? (uint8_t)r.algorithm : 255;
            beh[q] = r.behavior; owner[q] = r.is_owner ? 1 : 0;
            nlen[q] = (uint16_t)std::min<uint32_t>(S.klen[ri] - 1 - r.ukey_len, 0xffff);
        }
        if (s.dest && !P.dev_route_) {                               // a front stage whose callers route: every request's shard and its place in the shard's share
            uint32_t cnt[kMaxEngines] = {0}, at[kMaxEngines];
            for (uint32_t q = 0; q < n; ++q) cnt[S.eng[list[q]] & (kMaxEngines - 1)]++;
            for (uint32_t e = 0; e < kMaxEngines; ++e) at[e] = cnt[e] ? s.eng_n[e].fetch_add(cnt[e], std::memory_order_relaxed) : 0;
            uint32_t* dest = s.dest + t.first_slot;
            for (uint32_t q = 0; q < n; ++q) { const uint32_t e = S.eng[list[q]] & (kMaxEngines - 1); dest[q] = e << 24 | at[e]++; }
        }
        stream64((int64_t*)b->hits + t.first_slot, c_hits, n); stream64((int64_t*)b->limit + t.first_slot, c_limit, n);
        stream64((int64_t*)b->duration + t.first_slot, c_dur, n); stream64((int64_t*)b->burst + t.first_slot, c_burst, n);
        stream64((int64_t*)b->created_at + t.first_slot, c_created, n);
#if defined(__x86_64__)
        __builtin_ia32_sfence();                                      // the streamed stores are globally visible before the count says so
#endif
        s.written.fetch_add(n, std::memory_order_release);
    }

    // the responses of a ticket, once its generation has been announced; returns false when not ready and !block
    bool consume(Ticket2& t, bool block) {
        Stage& s = *t.st;
        // (a stage carries generation g + 1 only after every ticket of g has been consumed, so the word reads g - 1 or g here)
        // A caller of a small batch looks for a moment (the answer is a few microseconds away), everybody else sleeps on the
        // word at once: a host's cores belong to the callers that still have requests to write.  The dispatcher wakes TWO
        // sleepers per announcement and every woken caller wakes two more — the wake-ups fan out as a tree instead of being
        // one thread's serial work (hundreds of callers may wait for one generation).
        uint32_t v = s.done_gen.load(std::memory_order_acquire);
        if (v != (uint32_t)t.gen) {
            if (!block) return false;
            bool spin = P.spin_us_ && word_count(s.word.load(std::memory_order_relaxed)) <= 256;
            if (spin && P.spinners_.fetch_add(1, std::memory_order_relaxed) >= P.max_spinners_) { P.spinners_.fetch_sub(1, std::memory_order_relaxed); spin = false; }
            if (spin) {
                const int64_t t0 = mono_us();
                for (uint32_t spins = 0; (v = s.done_gen.load(std::memory_order_acquire)) != (uint32_t)t.gen; ++spins) {
                    if ((spins & 63u) == 63u && mono_us() - t0 >= (int64_t)P.spin_us_) break;
                    cpu_relax();
                }
                P.spinners_.fetch_sub(1, std::memory_order_relaxed);
            }
            if (v != (uint32_t)t.gen) {
                s.sleepers.fetch_add(1, std::memory_order_seq_cst);
                while ((v = s.done_gen.load(std::memory_order_seq_cst)) != (uint32_t)t.gen) futex_wait(&s.done_gen, v);
                if (s.sleepers.fetch_sub(1, std::memory_order_seq_cst) > 1) futex_wake_n(&s.done_gen, 2);
            }
        }
        const int rc = s.rc;
        const guber_result_t* r = s.r;
        const uint32_t* list = S.order.data() + t.list_begin;
        if (rc != GUBER_OK) {
            for (uint32_t q = 0; q < t.count; ++q) sink.engine_error(list[q], rc);
        } else {
            for (uint32_t q = 0; q < t.count; ++q) {
                const uint32_t ri = list[q], i = t.first_slot + q;
                if (r->err[i] == 0) sink.ok(ri, r->status[i], r->limit[i], r->remaining[i], r->reset_time[i]);
                else sink.item_error(ri, r->err[i], src.algorithm(ri));
            }
        }
        t.consumed = true;
        s.consumed.fetch_add(t.count, std::memory_order_release);
        return true;
    }
};

static void item_error_text(char* buf, size_t cap, uint8_t err, int32_t algorithm) {
    if (err == GUBER_ITEM_E_INVALID_ALGORITHM) snprintf(buf, cap, guber_item_strerror(err), algorithm);
    else snprintf(buf, cap, "%s", guber_item_strerror(err));
}

// ---- source / sink of the C++ interface: RateLimitReq / RateLimitResp objects
namespace {
struct VecSrc {
    const std::vector<const RateLimitReq*>& reqs; const std::vector<RateLimitReqState>& st;
    uint32_t size() const { return (uint32_t)reqs.size(); }
    bool front_end_checks() const { return false; }                  // (V1Instance::GetRateLimits has made them)
    size_t key_bytes_total() const { size_t t = 0; for (auto* q : reqs) t += q->name.size() + 1 + q->unique_key.size(); return t; }
    void key(uint32_t i, ReqRef& r) const {
        const RateLimitReq& q = *reqs[i];
        r.name = (const uint8_t*)q.name.data(); r.name_len = (uint32_t)q.name.size();
        r.ukey = (const uint8_t*)q.unique_key.data(); r.ukey_len = (uint32_t)q.unique_key.size();
    }
    uint32_t behavior(uint32_t i) const { return reqs[i]->behavior; }
    int32_t algorithm(uint32_t i) const { return reqs[i]->algorithm; }
    void get(uint32_t i, ReqRef& r) const {
        const RateLimitReq& q = *reqs[i];
        r.name = (const uint8_t*)q.name.data(); r.name_len = (uint32_t)q.name.size();
        r.ukey = (const uint8_t*)q.unique_key.data(); r.ukey_len = (uint32_t)q.unique_key.size();
        r.hits = q.hits; r.limit = q.limit; r.duration = q.duration; r.burst = q.burst; r.created_at = q.created_at;
        r.algorithm = q.algorithm; r.behavior = q.behavior; r.is_owner = st[i].is_owner;
    }
};
struct VecSink {
    std::vector<RateLimitResp*>& out;
    void ok(uint32_t i, uint8_t status, int64_t limit, int64_t remaining, int64_t reset_time) {
        RateLimitResp& o = *out[i];
        o.status = status; o.limit = limit; o.remaining = remaining; o.reset_time = reset_time;
        if (!o.error.empty()) o.error.clear();
    }
    void item_error(uint32_t i, uint8_t err, int32_t algorithm) {
        char buf[256];
        item_error_text(buf, sizeof buf, err, algorithm);
        *out[i] = RateLimitResp{}; out[i]->error = buf;              // nil response + error (workers.go:317-321)
    }
    void engine_error(uint32_t i, int rc) { *out[i] = RateLimitResp{}; out[i]->error = std::string("gpu engine: ") + guber_strerror(rc); }
    void closed(uint32_t i) { *out[i] = RateLimitResp{}; out[i]->error = "worker pool is closed"; }
};
}  // namespace

void GPUWorkerPool::GetRateLimitMany(const std::vector<const RateLimitReq*>& reqs, const std::vector<RateLimitReqState>& st,
                                     std::vector<RateLimitResp*>& out) {
    static thread_local Scratch tls;
    VecSrc src{reqs, st}; VecSink sink{out};
    Call<VecSrc, VecSink>{*this, src, sink, tls}.run();
}

// ---- source / sink of the C interface (what a binding hands over: structure-of-arrays in, structure-of-arrays out) with the
// V1Instance.GetRateLimits front end folded in (gubernator.go:183-306): empty-field errors, CreatedAt default, error wrapping
namespace {
struct SoaSrc {
    uint32_t n; const uint8_t* name_bytes; const uint32_t* name_off; const uint8_t* ukey_bytes; const uint32_t* ukey_off;
    const int64_t *hits, *limit, *duration, *burst, *created_at; const int32_t* algo; const uint32_t* beh; const uint8_t* owner;
    int64_t created_default;
    uint32_t size() const { return n; }
    bool front_end_checks() const { return true; }                   // empty fields have been answered: skip them
    size_t key_bytes_total() const { return (size_t)(name_off[n] - name_off[0]) + (ukey_off[n] - ukey_off[0]) + n; }
    void key(uint32_t i, ReqRef& r) const {
        r.name = name_bytes + name_off[i]; r.name_len = name_off[i + 1] - name_off[i];
        r.ukey = ukey_bytes + ukey_off[i]; r.ukey_len = ukey_off[i + 1] - ukey_off[i];
    }
    uint32_t behavior(uint32_t i) const { return beh ? beh[i] : 0; }
    int32_t algorithm(uint32_t i) const { return algo ? algo[i] : 0; }
    void get(uint32_t i, ReqRef& r) const {
        r.name = name_bytes + name_off[i]; r.name_len = name_off[i + 1] - name_off[i];
        r.ukey = ukey_bytes + ukey_off[i]; r.ukey_len = ukey_off[i + 1] - ukey_off[i];
        r.hits = hits[i]; r.limit = limit[i]; r.duration = duration[i]; r.burst = burst ? burst[i] : 0;
        r.created_at = created_at && created_at[i] ? created_at[i] : created_default;                    // gubernator.go:218-220
        r.algorithm = algorithm(i); r.behavior = behavior(i); r.is_owner = owner ? owner[i] != 0 : true;
    }
};
struct SoaSink {
    const SoaSrc& src; guber_result_t* out; char* err_text; uint32_t stride;
    void fail(uint32_t i, const char* msg, bool wrap) {
        out->status[i] = 0; out->limit[i] = 0; out->remaining[i] = 0; out->reset_time[i] = 0; out->err[i] = 1;
        if (!err_text || !stride) return;
        char* dst = err_text + (size_t)i * stride;
        if (!wrap) { snprintf(dst, stride, "%s", msg); return; }
        // gubernator.go:250-255: errors of the local path are wrapped with the key
        ReqRef r; src.get(i, r);
        snprintf(dst, stride, "Error while apply rate limit for '%.*s_%.*s': %s", (int)r.name_len, (const char*)r.name, (int)r.ukey_len, (const char*)r.ukey, msg);
    }
    void ok(uint32_t i, uint8_t status, int64_t limit, int64_t remaining, int64_t reset_time) {
        out->status[i] = status; out->limit[i] = limit; out->remaining[i] = remaining; out->reset_time[i] = reset_time; out->err[i] = 0;
    }
    void item_error(uint32_t i, uint8_t err, int32_t algorithm) { char buf[256]; item_error_text(buf, sizeof buf, err, algorithm); fail(i, buf, true); }
    void engine_error(uint32_t i, int rc) { char buf[128]; snprintf(buf, sizeof buf, "gpu engine: %s", guber_strerror(rc)); fail(i, buf, true); }
    void closed(uint32_t i) { fail(i, "worker pool is closed", true); }
};
}  // namespace

int GPUWorkerPool::GetRateLimitsSoA(uint32_t n, const uint8_t* name_bytes, const uint32_t* name_off, const uint8_t* ukey_bytes,
                                    const uint32_t* ukey_off, const int64_t* hits, const int64_t* limit, const int64_t* duration,
                                    const int64_t* burst, const int64_t* created_at, const int32_t* algorithm, const uint32_t* behavior,
                                    guber_result_t* out, char* err_text, uint32_t err_stride, const uint8_t* is_owner) {
    if (n > kMaxBatchSize) {                                          // gubernator.go:189-193
        if (err_text && err_stride) snprintf(err_text, err_stride, "Requests.RateLimits list too large; max size is '%u'", kMaxBatchSize);
        return GUBER_E_BATCH_TOO_LARGE;
    }
    static thread_local Scratch tls;
    SoaSrc src{n, name_bytes, name_off, ukey_bytes, ukey_off, hits, limit, duration, burst, created_at, algorithm, behavior, is_owner, NowMs()};
    SoaSink sink{src, out, err_text, err_stride};
    if (err_text && err_stride) for (uint32_t i = 0; i < n; ++i) err_text[(size_t)i * err_stride] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (ukey_off[i + 1] == ukey_off[i]) sink.fail(i, "field 'unique_key' cannot be empty", false);       // gubernator.go:208-212
        else if (name_off[i + 1] == name_off[i]) sink.fail(i, "field 'namespace' cannot be empty", false);   // :213-217
    }
    Call<SoaSrc, SoaSink>{*this, src, sink, tls}.run();
    return GUBER_OK;
}

// ---- the dispatcher's side -----------------------------------------------------------------------------------------------
void GPUWorkerPool::open_stage(Shard& sh, uint32_t k, uint32_t ver) {
    Stage& s = sh.st[k];
    s.n = 0; s.rc = GUBER_OK; s.submitted = false;
    s.written.store(0); s.consumed.store(0); s.first_us.store(0); s.last_us.store(0); s.flush_now.store(false);
    if (sh.front) for (auto& c : s.eng_n) c.store(0, std::memory_order_relaxed);
    s.gen.store(s.gen.load() + 1);
    s.state = Stage::kOpen;
    PT("open", &s, s.gen.load(), ver, k);
    s.word.store((uint64_t)(ver & 0x7fu) << 56, std::memory_order_release);
    sh.cur = k;
    sh.open.store(k, std::memory_order_release);
    sh.open_seq.fetch_add(1, std::memory_order_seq_cst);
    if (sh.open_waiters.load(std::memory_order_seq_cst) > 0) futex_wake_all(&sh.open_seq);
}

// a stage whose responses have all been picked up is free again
int GPUWorkerPool::find_free(Shard& sh) {
    for (uint32_t k = 0; k < kStages; ++k) {
        Stage& s = sh.st[k];
        if (s.state == Stage::kDraining && s.consumed.load(std::memory_order_acquire) == s.n) s.state = Stage::kFree;
        if (s.state == Stage::kFree) return (int)k;
    }
    return -1;
}

// close the shard's open stage to reservations; with requests in it, it joins `due`
void GPUWorkerPool::seal(Shard& sh, Stage& s, std::vector<Stage*>& due) {
    const uint64_t w = s.word.fetch_or(kClosed, std::memory_order_acq_rel);
    s.n = word_count(w);
    PT("seal", &s, s.gen.load(), s.n, word_ver(w));
    if (s.n == 0) { s.state = Stage::kFree; return; }
    ((uint32_t*)s.b->key_off)[s.n] = word_bytes(w);
    s.state = Stage::kSealed;
    due.push_back(&s);
}

// Flush policy of one shard: at batch_limit, at batch_wait after the first reservation (peer_client.go:284-337), when a caller
// found no room, and — by default — as soon as nobody has reserved anything for idle_us and every reserved slot is written
// (the callers are all waiting: holding the batch only adds latency).  *deadline = when to look again at the latest.
void GPUWorkerPool::seal_if_due(Shard& sh, int64_t now, bool force, bool eager_ok, std::vector<Stage*>& due, int64_t* deadline) {
    Device& d = *sh.dev;
    if (sh.open.load(std::memory_order_relaxed) == kOpenNone) {      // no stage was free when the last one was sealed
        const int k = find_free(sh);
        if (k >= 0) open_stage(sh, (uint32_t)k, d.ver.load(std::memory_order_relaxed));
        else *deadline = std::min(*deadline, now + 20);
        return;
    }
    Stage& s = sh.st[sh.cur];
    if (s.state != Stage::kOpen) return;
    const uint32_t cnt = word_count(s.word.load(std::memory_order_acquire));
    if (cnt == 0) return;
    bool is_due = cnt >= stage_cap_ || force || s.flush_now.load(std::memory_order_relaxed);
    if (!is_due && eager_ok) is_due = true;                          // (the device has room: see run())
    if (!is_due) {
        const int64_t first = s.first_us.load(std::memory_order_acquire), last = s.last_us.load(std::memory_order_acquire);
        if (!first) { *deadline = std::min(*deadline, now + 5); return; }       // the first reserver is about to stamp it
        if (now - first >= (int64_t)batch_wait_us_) is_due = true;
        else {
            *deadline = std::min(*deadline, first + (int64_t)batch_wait_us_);
            if (idle_us_) {
                if (now - last >= (int64_t)idle_us_) {
                    if (s.written.load(std::memory_order_acquire) == cnt && word_count(s.word.load(std::memory_order_acquire)) == cnt) is_due = true;
                    else *deadline = std::min(*deadline, now + 2);
                } else *deadline = std::min(*deadline, last + (int64_t)idle_us_);
            }
        }
    }
    if (!is_due) return;
    // the next free stage takes the reservations from here on
    const int k = find_free(sh);
    const uint32_t cur = sh.cur;
    if (k >= 0) open_stage(sh, (uint32_t)k, d.ver.load(std::memory_order_relaxed));
    else sh.open.store(kOpenNone, std::memory_order_release);
    seal(sh, sh.st[cur], due);
}

// hand the sealed stages to the engines: ONE submission for all of them (fused launches), or — with a persistent Store
// configured — the synchronous path that makes the Store's calls
void GPUWorkerPool::submit_due(Device& d, std::vector<Stage*>& due, std::vector<Stage*>& inflight) {
    if (due.empty()) return;
    const int64_t t0 = mono_us();
    const int64_t now_ms = NowMs();
    for (Stage* s : due) {
        for (uint32_t spins = 0; s->written.load(std::memory_order_acquire) != s->n; ++spins) {   // callers still copying their requests in
            if ((spins & 15u) == 15u && !inflight.empty()) (void)poll(inflight);                  // (the batches on the GPU are announced meanwhile)
            if (spins < 2000) cpu_relax(); else std::this_thread::yield();
        }
        s->t_written_us = mono_us();
        Shard& sh = *s->shard;
        s->t0_us = t0;
        sh.requests += s->n;
        if (s->n > sh.batch_max.load(std::memory_order_relaxed)) sh.batch_max.store(s->n, std::memory_order_relaxed);
        s->b->n = s->n; s->b->now_ms = now_ms;                       // DURATION_IS_GREGORIAN: the kernels derive the interval from now_ms
        sh.in_flight++;
    }
    if (has_store_.load()) {
        for (Stage* s : due) { submit_with_store(*s->shard, *s); announce(*s->shard, *s); }
        due.clear();
        return;
    }
    const uint32_t gen = d.gen_seq++ & 7u;
    if (routed_) {                                                   // the device's front stage: the GPU hands the requests to the shards
        const uint32_t ne = (uint32_t)d.shards.size();
        for (Stage* s : due) {
            int rc;
            const int64_t ts = mono_us();
            if (dev_route_ && s->n <= 256 && d.routing_q.empty()) {  // a handful of requests: routed here (one launch then, no round trip for the sizes)
                // (only while no earlier generation is still waiting for its shares' sizes: that one's batch is enqueued when they are
                // back, and a generation submitted here meanwhile would be evaluated BEFORE it — the second half of an RPC ahead of the first)
                route_on_host(d, *s);
                uint32_t counts[kMaxEngines];
                for (uint32_t j = 0; j < ne; ++j) counts[j] = s->eng_n[j].load(std::memory_order_relaxed);
                rc = submit_routed_now(d, *s, counts) ? GUBER_OK : s->rc;
            } else if (dev_route_) {
                // ... after deciding which shard each belongs to: two small launches now, guber_stage_submit_routed as soon as the
                // shares' sizes are back (poll) — the rule travels with the first stage after a placement change
                static const uint16_t one_shard[1] = {0};
                guber_route_rule_t rule{};
                if (d.rule_dirty) {
                    if (d.place) (void)guber_placement_export(d.place, &rule);
                    else { rule.n_shards = 1; rule.per = 1; rule.step = 1ull << 63; rule.table = one_shard; }
                    rule.global_engine = has_global_ ? (int32_t)d.n_plain : -1;
                }
                rc = guber_stage_route(s->stage, d.rule_dirty ? &rule : nullptr, ne);
                if (rc == GUBER_OK) { d.rule_dirty = false; s->routing = true; d.routing_q.push_back(s); }
            } else {
                uint32_t counts[kMaxEngines];
                for (uint32_t j = 0; j < ne; ++j) counts[j] = s->eng_n[j].load(std::memory_order_acquire);
                rc = submit_routed_now(d, *s, counts) ? GUBER_OK : s->rc;
            }
            d.submit_us.fetch_add((uint64_t)(mono_us() - ts), std::memory_order_relaxed); d.submits.fetch_add(1, std::memory_order_relaxed);
            if (rc == GUBER_OK) {
                s->state = Stage::kInFlight; s->shard->stages_in_flight++; s->t_submitted_us = mono_us(); inflight.push_back(s);
                s->dev_gen = gen;
                if (d.gen_left[gen]++ == 0) d.gens_in_flight++;
            } else { s->rc = rc; announce(*s->shard, *s); }
        }
        due.clear();
        return;
    }
    guber_stage_t* arr[64];
    for (size_t lo = 0; lo < due.size(); lo += 64) {
        const uint32_t m = (uint32_t)std::min<size_t>(64, due.size() - lo);
        for (uint32_t q = 0; q < m; ++q) arr[q] = due[lo + q]->stage;
        uint32_t done = 0;
        const int64_t ts = mono_us();
        for (uint32_t q = 0; q < m; ++q) PT("submit", due[lo + q], due[lo + q]->gen.load(), due[lo + q]->n, q);
        const int rc = guber_stages_submit(arr, m, GUBER_STAGES_NO_AGGREGATES, &done);
        d.submit_us.fetch_add((uint64_t)(mono_us() - ts), std::memory_order_relaxed); d.submits.fetch_add(1, std::memory_order_relaxed);
        for (uint32_t q = 0; q < m; ++q) {
            Stage* s = due[lo + q];
            if (q < done) {
                s->submitted = true; s->state = Stage::kInFlight; s->shard->stages_in_flight++; s->t_submitted_us = mono_us(); inflight.push_back(s);
                s->dev_gen = gen;
                if (d.gen_left[gen]++ == 0) d.gens_in_flight++;
            }
            else { (void)guber_stage_wait(s->stage); s->rc = rc != GUBER_OK ? rc : GUBER_E_HIP; announce(*s->shard, *s); }   // (it may have been enqueued after all)
        }
    }
    due.clear();
}

// the shares of a front stage go to the engines (their sizes known: from the callers' ranks, or from guber_stage_route)
bool GPUWorkerPool::submit_routed_now(Device& d, Stage& s, const uint32_t* counts) {
    guber_engine_t* eng[kMaxEngines];
    const uint32_t ne = (uint32_t)d.shards.size();
    for (uint32_t j = 0; j < ne; ++j) eng[j] = d.shards[j]->engine;
    PT("submit_routed", &s, s.gen.load(), s.n, 0);
    const int rc = guber_stage_submit_routed(s.stage, eng, ne, counts);
    s.submitted = rc == GUBER_OK;
    if (rc != GUBER_OK) s.rc = rc;
    return rc == GUBER_OK;
}
// a stage leaves the device: bookkeeping + announcement
void GPUWorkerPool::finish(Stage& s) {
    s.submitted = false; s.routing = false; s.shard->stages_in_flight--;
    { Device& d = *s.shard->dev; if (--d.gen_left[s.dev_gen] == 0) d.gens_in_flight--; }
    { const int64_t t = mono_us(); d_dbg_[0] += (uint64_t)(s.t_written_us - s.t0_us); d_dbg_[1] += (uint64_t)(s.t_submitted_us - s.t_written_us); d_dbg_[2] += (uint64_t)(t - s.t_submitted_us); d_dbg_[3]++; }
    announce(*s.shard, s);
}
// look at the stages in flight; announce the ones whose responses are there.  Returns whether any finished.
bool GPUWorkerPool::poll(std::vector<Stage*>& inflight) {
    bool any = false;
    // generations whose shards the device is deciding (guber_stage_route): the sizes are back -> the batch itself goes.  Strictly in
    // the order they were sealed: the batches are evaluated in the order they are enqueued HERE, and the halves of an RPC that spans
    // two generations must be evaluated in that order (round 4 looked at them in the order of `inflight`, which swap-removes, and let a
    // small generation that is routed on the host overtake: the answers of such an RPC came out permuted — found under ThreadSanitizer's
    // timing)
    if (!inflight.empty()) {
        Device& d = *inflight[0]->shard->dev;
        while (!d.routing_q.empty()) {
            Stage* s = d.routing_q.front();
            uint32_t counts[kMaxEngines] = {0};
            const int r = guber_stage_route_poll(s->stage, counts);
            if (r == 0) break;
            d.routing_q.pop_front();
            s->routing = false;
            any = true;
            if (r > 0 && submit_routed_now(d, *s, counts)) continue;
            if (r < 0) s->rc = r;
            finish(*s);
            for (size_t q = 0; q < inflight.size(); ++q) if (inflight[q] == s) { inflight[q] = inflight.back(); inflight.pop_back(); break; }
        }
    }
    for (size_t q = 0; q < inflight.size();) {
        Stage* s = inflight[q];
        if (s->routing) { ++q; continue; }                           // (its turn comes above)
        const int r = guber_stage_poll(s->stage);
        if (r == 0) { ++q; continue; }
        s->rc = r < 0 ? r : guber_stage_wait(s->stage);              // (already complete: resolves the rare internal retry)
        finish(*s);
        inflight[q] = inflight.back(); inflight.pop_back();
        any = true;
    }
    return any;
}

// announce a stage's generation: the callers read their responses themselves
void GPUWorkerPool::announce(Shard& sh, Stage& s) {
    const uint64_t us = (uint64_t)std::max<int64_t>(mono_us() - s.t0_us, 0);
    sh.send_us_sum += us;
    if (us > sh.send_us_max.load(std::memory_order_relaxed)) sh.send_us_max.store(us, std::memory_order_relaxed);
    sh.flushed++; sh.in_flight--;
    PT("announce", &s, s.gen.load(), s.n, s.rc);
    s.state = Stage::kDraining;
    s.done_gen.store((uint32_t)s.gen.load(), std::memory_order_seq_cst);
    if (s.sleepers.load(std::memory_order_seq_cst) > 0) futex_wake_n(&s.done_gen, 2);
}

void GPUWorkerPool::drain(std::vector<Stage*>& inflight) {
    for (uint32_t spins = 0; !inflight.empty(); ++spins) {
        if (!poll(inflight)) { if (spins < 2000) cpu_relax(); else std::this_thread::yield(); }
    }
}

// Placement pass: the keys that turned out to carry a large share of the device's traffic get a shard of their own choice.
// A resident key moves at a batch boundary — every open stage is sealed, the batches in flight drain, the bucket changes
// tables on the device, the new exception list is published and the stages reopen under the next placement version — so that
// every request of the key routed with the old placement has been evaluated before the first one routed with the new.
void GPUWorkerPool::rebalance(Device& d, std::vector<Stage*>& inflight) {
    if (!d.place || d.n_plain < 2) return;
    guber_placement_move_t moves[64];
    uint32_t nm = 0;
    if (guber_placement_plan(d.place, 0.125, moves, 64, &nm) != GUBER_OK) return;   // (nothing published: the next pass plans afresh)
    d.rebalances++;
    if (nm == 0) { (void)guber_placement_commit(d.place); return; }  // (keys pinned where they are change nothing a caller can see)
    std::vector<Stage*> due;
    PT("rebalance", &d, nm, d.ver.load(), inflight.size());
    for (Shard* sh : d.staging) {
        if (sh->open.load(std::memory_order_relaxed) == kOpenNone) continue;
        sh->open.store(kOpenNone, std::memory_order_release);
        Stage& s = sh->st[sh->cur];
        if (s.state == Stage::kOpen) seal(*sh, s, due);
    }
    submit_due(d, due, inflight);
    drain(inflight);
    {
        std::unique_lock<std::shared_mutex> lk(d.place_mu);
        for (uint32_t q = 0; q < nm; ++q) {
            if (moves[q].from >= d.n_plain || moves[q].to >= d.n_plain) continue;
            uint32_t moved = 0;
            const int mrc = guber_move_items_by_hash(d.shards[moves[q].from]->engine, d.shards[moves[q].to]->engine, &moves[q].key_hash, 1, &moved);
            PT("move", moves[q].key_hash, moves[q].from, moves[q].to, moved);
            // a bucket that did not arrive (an error, a full or colliding destination) went back to its table: its key must keep
            // following its slot, or its requests would start a fresh bucket elsewhere.  (moved == 0 with no error also means
            // "the hash named no live bucket" — nothing to lose, the new placement may stand.)
            if (mrc != GUBER_OK) (void)guber_placement_cancel(d.place, moves[q].key_hash);
            else d.moves += moved;
        }
        (void)guber_placement_commit(d.place);
        d.rule_dirty = true;
        d.ver.fetch_add(1, std::memory_order_acq_rel);
    }
    const uint32_t ver = d.ver.load();
    PT("rebalanced", &d, ver, 0, 0);
    for (Shard* sh : d.staging) {
        const int k = find_free(*sh);
        if (k >= 0) open_stage(*sh, (uint32_t)k, ver);               // (none free: seal_if_due opens one as soon as its callers have read it out)
        else { sh->open_seq.fetch_add(1, std::memory_order_release); futex_wake_all(&sh->open_seq); }
    }
}

void GPUWorkerPool::RebalanceNow() {
    for (auto& d : devs_) { d->rebalance_now.store(true); wake(*d); }
}

void GPUWorkerPool::run(Device& d) {
    std::vector<Stage*> inflight, due;
    for (Shard* sh : d.staging) open_stage(*sh, 0, 0);
    int64_t next_rebalance = rebalance_ms_ ? mono_us() + (int64_t)rebalance_ms_ * 1000 : INT64_MAX;
    uint32_t idle_spins = 0;
    int64_t last_active = 0;
    for (;;) {
        const uint32_t seen = d.wake.load(std::memory_order_seq_cst);
        const bool closing = d.closing.load(std::memory_order_acquire);
        d_dbg_[4]++; d_dbg_[5] += inflight.size();
        const int64_t now = mono_us();
        bool progressed = !inflight.empty() && poll(inflight);
        int64_t deadline = INT64_MAX;
        // The reference's workers take a request the moment it arrives (workers.go:261-291): what is waiting goes — the batches of
        // ALL the device's shards together, so that they share launches — as soon as the device has nothing in flight; it then
        // collects what arrives while that runs, so batches grow with the load by themselves.  A second generation follows
        // behind the first once it is worth its launches.
        bool eager_ok = false;
        if (eager_ && d.gens_in_flight < depth_) {
            if (d.gens_in_flight == 0) eager_ok = true;
            else {
                uint64_t pending = 0;
                for (Shard* sh : d.staging) if (sh->st[sh->cur].state == Stage::kOpen) pending += word_count(sh->st[sh->cur].word.load(std::memory_order_relaxed));
                eager_ok = pending >= eager_min_;
            }
        }
        for (Shard* sh : d.staging) seal_if_due(*sh, now, closing, eager_ok, due, &deadline);
        if (!due.empty()) { submit_due(d, due, inflight); progressed = true; }
        if (closing && inflight.empty()) {
            // stop taking reservations; a caller may have slipped one in meanwhile: those are still evaluated
            for (Shard* sh : d.staging) {
                const uint32_t was = sh->open.exchange(kOpenDead, std::memory_order_acq_rel);
                if (was < kStages && sh->st[was].state == Stage::kOpen) seal(*sh, sh->st[was], due);
                sh->open_seq.fetch_add(1, std::memory_order_release);
                futex_wake_all(&sh->open_seq);
            }
            submit_due(d, due, inflight);
            drain(inflight);
            break;
        }
        if (!closing && (now >= next_rebalance || d.rebalance_now.exchange(false))) {
            rebalance(d, inflight);
            if (rebalance_ms_) next_rebalance = mono_us() + (int64_t)rebalance_ms_ * 1000;
            progressed = true;
        }
        if (progressed) { idle_spins = 0; last_active = now; continue; }
        if (!inflight.empty()) {                                     // something is on the GPU: keep looking (completion latency matters)
            if (++idle_spins < 4000) cpu_relax(); else std::this_thread::yield();
            last_active = now;
            continue;
        }
        int64_t wait = deadline == INT64_MAX ? 2000 : std::max<int64_t>(deadline - now, 1);
        if (next_rebalance != INT64_MAX) wait = std::min(wait, std::max<int64_t>(next_rebalance - now, 1));
        if (wait <= 3 || now - last_active < (int64_t)spin_us_ * 4) { cpu_relax(); continue; }   // requests tend to come in trains: stay awake a little
        d.sleeping.store(true, std::memory_order_seq_cst);
        if (d.wake.load(std::memory_order_seq_cst) == seen) futex_wait(&d.wake, seen, wait);
        d.sleeping.store(false, std::memory_order_seq_cst);
    }
}

// Config.Store (store.go:49-65) configured: ask the store for keys that are not resident BEFORE the batch
// (algorithms.go:45-51 `s.Get` on a cache miss, then `c.Add(item)`), evaluate, then issue the Remove / OnChange
// calls the reference makes from inside the algorithms, in request order.  Synchronous, host-pointer entry points over
// the stage's own arrays.
void GPUWorkerPool::submit_with_store(Shard& sh, Stage& s) {
    s.submitted = false;
    if (!sh.front) {
        guber_batch_t B = *s.b; B.n = s.n;
        s.rc = store_eval(sh.engine, B, *s.r, s.name_len.data());
        return;
    }
    // a device's front stage: every shard's share is gathered in rank order (the order the device would apply it in), goes
    // through the Store sequence on that shard's engine, and the answers return to the slots the callers wrote
    Device& d = *sh.dev;
    const guber_batch_t& B = *s.b;
    int rc = GUBER_OK;
    if (dev_route_) route_on_host(d, s);                             // (this path is synchronous and on the host anyway)
    for (uint32_t e = 0; e < d.shards.size() && rc == GUBER_OK; ++e) {
        const uint32_t ne = s.eng_n[e].load(std::memory_order_acquire);
        if (!ne) continue;
        std::vector<uint32_t> at(ne, 0);
        for (uint32_t i = 0; i < s.n; ++i) if ((s.dest[i] >> 24) == e) at[s.dest[i] & 0xffffffu] = i;
        std::vector<uint8_t> keys; std::vector<uint32_t> off(ne + 1, 0), beh(ne); std::vector<uint16_t> nlen(ne);
        std::vector<int64_t> hits(ne), limit(ne), duration(ne), burst(ne), created(ne), rl(ne), rr(ne), rs(ne);
        std::vector<uint8_t> algo(ne), owner(ne), status(ne), err(ne);
        for (uint32_t r = 0; r < ne; ++r) {
            const uint32_t i = at[r];
            keys.insert(keys.end(), B.key_bytes + B.key_off[i], B.key_bytes + (i + 1 < s.n ? B.key_off[i + 1] : B.key_off[s.n]));
            off[r + 1] = (uint32_t)keys.size();
            hits[r] = B.hits[i]; limit[r] = B.limit[i]; duration[r] = B.duration[i]; burst[r] = B.burst[i]; created[r] = B.created_at[i];
            algo[r] = B.algorithm[i]; beh[r] = B.behavior[i]; owner[r] = B.is_owner[i]; nlen[r] = s.name_len[i];
        }
        keys.resize(keys.size() + 16, 0);
        guber_batch_t b{}; guber_result_t res{};
        b.n = ne; b.key_bytes = keys.data(); b.key_off = off.data(); b.hits = hits.data(); b.limit = limit.data(); b.duration = duration.data();
        b.burst = burst.data(); b.created_at = created.data(); b.algorithm = algo.data(); b.behavior = beh.data(); b.is_owner = owner.data(); b.now_ms = B.now_ms;
        res.status = status.data(); res.limit = rl.data(); res.remaining = rr.data(); res.reset_time = rs.data(); res.err = err.data();
        rc = store_eval(d.shards[e]->engine, b, res, nlen.data());
        for (uint32_t r = 0; r < ne && rc == GUBER_OK; ++r) {
            const uint32_t i = at[r];
            s.r->status[i] = status[r]; s.r->limit[i] = rl[r]; s.r->remaining[i] = rr[r]; s.r->reset_time[i] = rs[r]; s.r->err[i] = err[r];
        }
    }
    s.rc = rc;
}

// what guber_stage_route computes, on the host: every request's shard and its rank in the shard's share, in arrival order
void GPUWorkerPool::route_on_host(Device& d, Stage& s) {
    const guber_batch_t& B = *s.b;
    uint32_t cnt[kMaxEngines] = {0};
    for (uint32_t i = 0; i < s.n; ++i) {
        const uint32_t off = B.key_off[i], len = (i + 1 < s.n ? B.key_off[i + 1] : B.key_off[s.n]) - off;
        const uint32_t e = route(d, guber::xxhash64(B.key_bytes + off, len, 0), B.behavior[i]) & (kMaxEngines - 1);
        s.dest[i] = e << 24 | cnt[e]++;
    }
    for (uint32_t e = 0; e < kMaxEngines; ++e) s.eng_n[e].store(cnt[e], std::memory_order_relaxed);
}

int GPUWorkerPool::store_eval(guber_engine_t* engine_, const guber_batch_t& B, guber_result_t& R, const uint16_t* name_len) {
    const uint32_t n_all = B.n;
    const uint8_t* keys = B.key_bytes; const uint32_t* off = B.key_off;
    // The store is asked on EVERY cache miss (algorithms.go:45-51), also on the miss a request causes for a later request of
    // the same key in the same batch: RESET_REMAINING removes the item from cache and store (:78-90), and the key's next
    // request misses again.  The batch is therefore cut in front of any request whose key an earlier request of the cut has
    // reset: each piece is a batch of its own (residency probe, Store.Get, evaluation, callbacks), which changes nothing else.
    std::vector<uint32_t> cuts{0};
    {
        bool any_reset = false;
        for (uint32_t i = 0; i < n_all && !any_reset; ++i) any_reset = (B.behavior[i] & 8u) != 0;      // Behavior_RESET_REMAINING
        if (any_reset) {
            std::vector<uint32_t> reset;                             // requests of the current piece that carry the bit
            for (uint32_t i = 0; i < n_all; ++i) {
                const uint32_t len = off[i + 1] - off[i];
                for (uint32_t q : reset)
                    if (off[q + 1] - off[q] == len && memcmp(keys + off[q], keys + off[i], len) == 0) { cuts.push_back(i); reset.clear(); break; }
                if (B.behavior[i] & 8u) reset.push_back(i);
            }
        }
        cuts.push_back(n_all);
    }
    int rc = GUBER_OK;
    for (size_t piece = 0; piece + 1 < cuts.size() && rc == GUBER_OK; ++piece) {
        const uint32_t lo = cuts[piece], n = cuts[piece + 1] - lo;
        if (n == 0) continue;
        guber_batch_t b = B;
        guber_result_t res = R;
        b.n = n; b.key_off = B.key_off + lo; b.hits = B.hits + lo; b.limit = B.limit + lo; b.duration = B.duration + lo;
        if (B.burst) b.burst = B.burst + lo;
        if (B.created_at) b.created_at = B.created_at + lo;
        if (B.algorithm) b.algorithm = B.algorithm + lo;
        if (B.behavior) b.behavior = B.behavior + lo;
        if (B.is_owner) b.is_owner = B.is_owner + lo;
        res.status += lo; res.limit += lo; res.remaining += lo; res.reset_time += lo; res.err += lo;
        const uint32_t* poff = b.key_off;
        std::vector<uint8_t> sflags(n, 0); std::vector<guber_item_t> sitems(n);
        guber_store_events_t sev{sflags.data(), sitems.data()};
        auto store_req = [&](uint32_t i) {
            guber_store_req_t q{};
            q.key = keys + poff[i]; q.key_len = poff[i + 1] - poff[i]; q.name_len = name_len[lo + i];
            q.hits = b.hits[i]; q.limit = b.limit[i]; q.duration = b.duration[i]; q.burst = b.burst[i]; q.created_at = b.created_at[i];
            q.algorithm = b.algorithm[i] == 255 ? -1 : b.algorithm[i]; q.behavior = b.behavior[i];
            return q;
        };
        std::vector<uint8_t> missing(n, 0);
        rc = guber_probe_missing(engine_, &b, missing.data());
        if (rc == GUBER_OK && store_.get) {
            std::vector<std::string> asked;
            for (uint32_t i = 0; i < n && rc == GUBER_OK; ++i) {
                if (!missing[i] || poff[i + 1] == poff[i]) continue;
                std::string k((const char*)keys + poff[i], poff[i + 1] - poff[i]);
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
                if ((sflags[i] & GUBER_STORE_REMOVE) && store_.remove) store_.remove(store_.user, keys + poff[i], poff[i + 1] - poff[i]);
                if ((sflags[i] & GUBER_STORE_ONCHANGE) && store_.on_change) { const guber_store_req_t q = store_req(i); store_.on_change(store_.user, &q, &sitems[i]); }
            }
        }
    }
    return rc;
}

// ---- cache operations from other threads: they follow the placement and exclude a move in progress -----------------------
// A key lives in ONE of two places on its device: the GLOBAL engine (requests with Behavior_GLOBAL go there: route()) or the plain
// shard its hash selects.  The reference has one cache per worker, so its AddCacheItem / GetCacheItem need no such distinction;
// here the caller says which (behavior: the RateLimitReq behaviour the item belongs to — UpdatePeerGlobals, gubernator.go:425-459,
// installs GLOBAL state), or leaves it open (behavior < 0): then the item goes where the key already is, the GLOBAL engine first.
// "does this engine hold the key?" without the side effects of a cache access: guber_probe_missing only reads (k_probe_missing) — no
// hit / miss counted, no move to the front of the recency order.  (The reference has ONE cache per worker, so its AddCacheItem /
// GetCacheItem never have to ask where a key lives; a probe that counted would skew a GLOBAL pool's metrics and LRU order: ADVICE r04.)
static int peek_resident(guber_engine_t* e, const uint8_t* key, uint32_t key_len, int64_t now_ms, bool* resident) {
    const uint32_t off[2] = {0, key_len};
    guber_batch_t b{};
    b.n = 1; b.key_bytes = key; b.key_off = off; b.now_ms = now_ms;
    uint8_t missing = 1;
    const int rc = guber_probe_missing(e, &b, &missing);
    *resident = rc == GUBER_OK && missing == 0;
    return rc;
}
int GPUWorkerPool::AddCacheItem(const guber_item_t& item, int behavior) {
    if (shards_.empty()) return GUBER_E_INVALID_ARG;
    const uint32_t dv = DeviceOf(item.key, item.key_len);
    Device& d = *devs_[dv];
    std::shared_lock<std::shared_mutex> lk(d.place_mu);
    if (has_global_) {
        guber_engine_t* ge = d.shards[d.n_plain]->engine;
        bool global = behavior >= 0 && (behavior & 2);
        if (behavior < 0) {
            const int rc = peek_resident(ge, item.key, item.key_len, NowMs(), &global);
            if (rc != GUBER_OK) return rc;
        }
        if (global) return guber_add_items(ge, &item, 1, nullptr);
    }
    return guber_add_items(shards_[ShardOf(item.key, item.key_len)]->engine, &item, 1, nullptr);
}
int GPUWorkerPool::GetCacheItem(const std::string& key, guber_item_t* out, bool* found) {
    int f = 0;
    if (shards_.empty()) return GUBER_E_INVALID_ARG;
    Device& d = *devs_[DeviceOf((const uint8_t*)key.data(), (uint32_t)key.size())];
    std::shared_lock<std::shared_mutex> lk(d.place_mu);
    if (has_global_) {                                                                 // the GLOBAL engine first (see AddCacheItem): a look without side effects,
        bool there = false;                                                            // then ONE counted access where the key lives, as the reference's one cache has
        const int rc = peek_resident(d.shards[d.n_plain]->engine, (const uint8_t*)key.data(), (uint32_t)key.size(), NowMs(), &there);
        if (rc != GUBER_OK) return rc;
        if (there) {
            const int rc2 = guber_get_item(d.shards[d.n_plain]->engine, (const uint8_t*)key.data(), (uint32_t)key.size(), NowMs(), out, &f);
            if (rc2 != GUBER_OK || f) { *found = f != 0; return rc2; }
            // (gone between the look and the access — evicted, or expired by now: the key's plain shard is asked like any other key's.
            //  An expired item of the GLOBAL engine that nobody asks for by a GLOBAL request keeps its place until it reaches the back
            //  of the list, as every expired item does: lrucache.go:111-128 reaps on access, :138-149 at the back)
        }
    }
    const int rc = guber_get_item(shards_[ShardOf(key)]->engine, (const uint8_t*)key.data(), (uint32_t)key.size(), NowMs(), out, &f);
    *found = f != 0;
    return rc;
}
int GPUWorkerPool::Load(const guber_item_t* items, uint32_t n, const uint8_t* global_hint) {
    // workers.go:329-449: every item goes to the worker that owns its key; chunks bound the staging buffers
    std::vector<std::shared_lock<std::shared_mutex>> locks;
    for (auto& d : devs_) locks.emplace_back(d->place_mu);
    // (a restored item carries no behaviour — CacheItem has none, cache.go:29-41: it goes to its key's plain shard unless the caller
    // says it belongs to GLOBAL requests, guber_pool_load_hinted)
    std::vector<std::vector<guber_item_t>> per(shards_.size());
    for (uint32_t i = 0; i < n; ++i) per[ShardOf(items[i].key, items[i].key_len, (global_hint && global_hint[i]) ? 2u : 0u)].push_back(items[i]);
    for (size_t j = 0; j < per.size(); ++j)
        for (size_t lo = 0; lo < per[j].size(); lo += 65536) {
            const int rc = guber_add_items(shards_[j]->engine, per[j].data() + lo, (uint32_t)std::min<size_t>(65536, per[j].size() - lo), nullptr);
            if (rc != GUBER_OK) return rc;
        }
    return GUBER_OK;
}
int GPUWorkerPool::Store(const std::function<void(const guber_item_t&)>& save) {
    std::vector<std::shared_lock<std::shared_mutex>> locks;
    for (auto& d : devs_) locks.emplace_back(d->place_mu);
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
    std::vector<std::shared_lock<std::shared_mutex>> locks;         // (a bucket on its way between two tables is in neither for a moment)
    for (auto& d : devs_) locks.emplace_back(d->place_mu);
    int64_t n = 0;
    for (auto& sh : shards_) n += guber_size(sh->engine);
    return n;
}

// one GlobalSyncWait tick (global.go:91-283) over the devices' GLOBAL engines: pending hits travel to the owners, the owners
// apply them and broadcast their state to the other devices' replicas — natively (guber_global_sync).  A daemon that is one
// rank of a multi-node ring builds its own communicator over GlobalEngine(device) instead (guber_comm_create_rank).
int GPUWorkerPool::GlobalSync(guber_global_sync_stats_t* stats) {
    if (!has_global_ || shards_.empty()) return GUBER_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(comm_mu_);
    if (closed_.load()) return GUBER_E_INVALID_ARG;
    if (!comm_) {
        std::vector<guber_engine_t*> eng;
        for (auto& d : devs_) eng.push_back(d->shards[d->n_plain]->engine);
        const int rc = guber_comm_create_local(eng.data(), (uint32_t)eng.size(), ring_, 0, &comm_);
        if (rc != GUBER_OK) return rc;
    }
    return guber_global_sync(comm_, NowMs(), stats);
}
guber_engine_t* GPUWorkerPool::GlobalEngine(uint32_t device) {
    if (!has_global_ || device >= devs_.size()) return nullptr;
    return devs_[device]->shards[devs_[device]->n_plain]->engine;
}

bool V1Instance::GetRateLimits(std::vector<RateLimitReq>& reqs, std::vector<RateLimitResp>* resps, std::string* rpc_error) {
    if (reqs.size() > kMaxBatchSize) {                                // gubernator.go:189-193
        char buf[128];
        snprintf(buf, sizeof buf, "Requests.RateLimits list too large; max size is '%u'", kMaxBatchSize);
        *rpc_error = buf;
        return false;
    }
    const int64_t created_at = pool_->NowMs();                        // :195
    resps->resize(reqs.size());
    static thread_local std::vector<const RateLimitReq*> send; static thread_local std::vector<RateLimitReqState> st; static thread_local std::vector<RateLimitResp*> out;
    send.clear(); st.clear(); out.clear();
    for (size_t i = 0; i < reqs.size(); ++i) {
        RateLimitReq& r = reqs[i];
        RateLimitResp& o = (*resps)[i];
        if (r.unique_key.empty()) { o = RateLimitResp{}; o.error = "field 'unique_key' cannot be empty"; continue; }   // :208-212
        if (r.name.empty()) { o = RateLimitResp{}; o.error = "field 'namespace' cannot be empty"; continue; }          // :213-217
        if (r.created_at == 0) r.created_at = created_at;                                                    // :218-220
        send.push_back(&r); st.push_back(RateLimitReqState{true}); out.push_back(&o);
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
    if (!p->ok()) { const int rc = p->create_error(); delete p; return rc ? rc : GUBER_E_INVALID_ARG; }
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
extern "C" int guber_pool_load_hinted(guber_pool_t* p, const guber_item_t* items, uint32_t n, const uint8_t* global_hint) {
    return p ? p->pool->Load(items, n, global_hint) : GUBER_E_INVALID_ARG;
}
extern "C" int guber_pool_store(guber_pool_t* p, void (*save)(void* user, const guber_item_t* item), void* user) {
    if (!p || !save) return GUBER_E_INVALID_ARG;
    return p->pool->Store([&](const guber_item_t& it) { save(user, &it); });
}
extern "C" void guber_pool_set_store(guber_pool_t* p, const guber_store_callbacks_t* cb) { if (p) p->pool->SetStore(cb); }
extern "C" void guber_pool_set_clock(guber_pool_t* p, int64_t now_ms) { if (p) p->pool->SetClockMs(now_ms); }
extern "C" guber_engine_t* guber_pool_engine(guber_pool_t* p) { return p ? p->pool->engine() : nullptr; }
extern "C" guber_engine_t* guber_pool_engine_at(guber_pool_t* p, uint32_t shard) { return p ? p->pool->engine(shard) : nullptr; }
extern "C" guber_engine_t* guber_pool_global_engine(guber_pool_t* p, uint32_t device) { return p ? p->pool->GlobalEngine(device) : nullptr; }
extern "C" int guber_pool_global_sync(guber_pool_t* p, guber_global_sync_stats_t* stats) { return p ? p->pool->GlobalSync(stats) : GUBER_E_INVALID_ARG; }
extern "C" void guber_pool_rebalance(guber_pool_t* p) { if (p) p->pool->RebalanceNow(); }
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
    if (!p || !out || (n && (!name_bytes || !name_off || !ukey_bytes || !ukey_off || !hits || !limit || !duration))) return GUBER_E_INVALID_ARG;
    return p->pool->GetRateLimitsSoA(n, name_bytes, name_off, ukey_bytes, ukey_off, hits, limit, duration, burst, created_at, algorithm, behavior,
                                     out, err_text, err_stride, nullptr);
}
// the same with RateLimitReqState.IsOwner per request (workers.go:261 GetRateLimit(ctx, req, reqState); NULL = every request owned):
// what V1Instance hands over for GLOBAL requests it answers from its replica (gubernator.go:395-421)
extern "C" int guber_pool_get_rate_limits_owner(guber_pool_t* p, uint32_t n, const uint8_t* name_bytes, const uint32_t* name_off,
                                                const uint8_t* ukey_bytes, const uint32_t* ukey_off, const int64_t* hits,
                                                const int64_t* limit, const int64_t* duration, const int64_t* burst,
                                                const int64_t* created_at, const int32_t* algorithm, const uint32_t* behavior,
                                                const uint8_t* is_owner, guber_result_t* out, char* err_text, uint32_t err_stride) {
    if (!p || !out || (n && (!name_bytes || !name_off || !ukey_bytes || !ukey_off || !hits || !limit || !duration))) return GUBER_E_INVALID_ARG;
    return p->pool->GetRateLimitsSoA(n, name_bytes, name_off, ukey_bytes, ukey_off, hits, limit, duration, burst, created_at, algorithm, behavior,
                                     out, err_text, err_stride, is_owner);
}
extern "C" int guber_pool_add_item(guber_pool_t* p, const guber_item_t* item) { return p && item ? p->pool->AddCacheItem(*item) : GUBER_E_INVALID_ARG; }
extern "C" int guber_pool_add_item_for(guber_pool_t* p, const guber_item_t* item, uint32_t behavior) {
    return p && item ? p->pool->AddCacheItem(*item, (int)(behavior & 0x7fffffffu)) : GUBER_E_INVALID_ARG;
}
extern "C" int guber_pool_get_item(guber_pool_t* p, const uint8_t* key, uint32_t key_len, guber_item_t* out, int* found) {
    if (!p || !key || !out || !found) return GUBER_E_INVALID_ARG;
    bool f = false;
    const int rc = p->pool->GetCacheItem(std::string((const char*)key, key_len), out, &f);
    *found = f ? 1 : 0;
    return rc;
}
extern "C" int64_t guber_pool_size(guber_pool_t* p) { return p ? p->pool->Size() : -1; }
