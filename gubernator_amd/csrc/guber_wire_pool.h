// guber_wire_pool.h — guber_wire_pool_*: the payload stage.  Caller threads (the gRPC handlers of a daemon) hand over the SERIALIZED
// GetRateLimitsReq / GetPeerRateLimitsReq of their RPC and get the serialized response back; per RPC the host does one compare-and-swap
// (a place in the open stage) and two memcpys (the payload into pinned memory, the response out of it).  Everything the reference does per
// REQUEST on the CPU — unmarshalling into heap objects (generated code of gubernator.proto:137-182), validation and the CreatedAt default
// (gubernator.go:189-220), HashKey (client.go:39-41), the worker's choice by XXH64 (workers.go:180-184, :261-289), the evaluation
// (algorithms.go), the answers' order (gubernator.proto:51-54), marshalling the response (gubernator.proto:184-203) — happens on the device:
// k_wire_* (decode) -> guber_front (k_fr_*: routing, the engines' fused pipelines, the answers in arrival order) -> k_wire_enc (every RPC's
// GetRateLimitsResp bytes, written in place into host memory over PCIe; an RPC with an item error: its raw answers, for the host transcoder).
// The batching shape is the reference's own (peer_client.go:284-337: a queue that is sent when it is full or BatchWait after its first
// entry), turned around: RPCs are the entries, a stage is the queue.
//
// Part of guber_engine.hip's translation unit (it uses the decoder's and the front's internals).
//
//   callers                     one word per stage: closed | generation | RPCs | items (upper bound) | bytes / 16 — a CAS reserves all three
//   two threads of the pool's   neither ever blocks on the GPU (events and host words are polled).  INTAKE seals the open stage (full, BatchWait after
//                               its first payload, or at once while fewer than two stages are queued for the decode), opens the next free one,
//                               enqueues the decodes, collects them and enqueues the front's routing; FRONT enqueues the evaluation when the
//                               shares' sizes are in host memory, announces the stage when its answers are, frees it when its callers have left:
//                               SEALED (payloads still being copied in) -> DECODING (event) -> DECODED -> ROUTING (host words) ->
//                               EVALUATING (event) -> ANSWERED (callers encoding their slices) -> FREE.
//                               (One thread did all of it at first: 110 us of its own time per stage — 28 decode, 18 routing, 64 evaluation
//                               enqueue — against a 150 us stage cycle: profiles/r06_wire_pool.txt.)
//   streams                     the decodes' copies and kernels on TWO streams of the pool's own (consecutive stages alternate: one stage's copy
//                               and latency-bound kernels run beside the other's), the front's routing stream, the engines' stream(s) — the HIP
//                               runtime has four hardware queues, so the engines of a payload stage had better share ONE stream
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <sched.h>
#include <sys/resource.h>
#include <climits>

#ifndef GUBER_WPL_WALK_MAX
#define GUBER_WPL_WALK_MAX 8192       // payloads shorter than this have their records counted; longer ones are bounded by the cap
#endif
namespace {
constexpr uint64_t WPL_CLOSED = 1ull << 63;
constexpr uint32_t WPL_MAX_RPCS = 4095, WPL_MAX_ITEMS = (1u << 20) - 1, WPL_MAX_B16 = (1u << 20) - 1, WPL_NONE = 15, WPL_MAX_STAGES = 12;
inline uint32_t wpl_gen(uint64_t w) { return (uint32_t)(w >> 52) & 0x7ffu; }
inline uint32_t wpl_rpcs(uint64_t w) { return (uint32_t)(w >> 40) & 0xfffu; }
inline uint32_t wpl_items(uint64_t w) { return (uint32_t)(w >> 20) & 0xfffffu; }
inline uint32_t wpl_b16(uint64_t w) { return (uint32_t)w & 0xfffffu; }
inline uint64_t wpl_word(uint32_t gen, uint32_t rpcs, uint32_t items, uint32_t b16) {
    return ((uint64_t)(gen & 0x7ffu) << 52) | ((uint64_t)rpcs << 40) | ((uint64_t)items << 20) | b16;
}
inline void wpl_futex_wait(std::atomic<uint32_t>* w, uint32_t seen, int64_t timeout_us) {
    struct timespec ts; ts.tv_sec = timeout_us / 1000000; ts.tv_nsec = (timeout_us % 1000000) * 1000;
    syscall(SYS_futex, (uint32_t*)w, FUTEX_WAIT_PRIVATE, seen, timeout_us < 0 ? nullptr : &ts, nullptr, 0);
}
inline void wpl_futex_wake(std::atomic<uint32_t>* w, int n) { syscall(SYS_futex, (uint32_t*)w, FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0); }
inline int64_t wpl_mono_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void wpl_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}
// how many RateLimitReq records a payload can hold at most: the top-level chain walked for short payloads (exact when well-formed), the cap for
// long ones (an RPC with more than the cap is turned away whole and takes no place: gubernator.go:189-193).  Counting the records of EVERY
// payload (a tag, a length and a skip per record: ~3 us for 1000) would keep stages full when long RPCs hold few, long requests, and was
// measured on full ones: -3 ... -4 % at every caller count (profiles/r06_wire_pool.txt) — the bound of a long payload stays the cap.
inline uint32_t wpl_item_bound(const uint8_t* p, size_t len, uint32_t cap) {
    if (len >= GUBER_WPL_WALK_MAX) return (uint32_t)std::min<size_t>(cap, len / 2);
    const uint8_t* end = p + len;
    uint32_t n = 0;
    while (p < end) {
        uint64_t tag = 0; int sh = 0;
        while (p < end && sh < 64) { const uint8_t b = *p++; tag |= (uint64_t)(b & 0x7f) << sh; sh += 7; if (!(b & 0x80)) break; }
        const uint32_t wt = (uint32_t)(tag & 7);
        if (wt == 0) { while (p < end && (*p++ & 0x80)) {} }
        else if (wt == 1) p += 8;
        else if (wt == 5) p += 4;
        else if (wt == 2) {
            uint64_t L = 0; sh = 0;
            while (p < end && sh < 64) { const uint8_t b = *p++; L |= (uint64_t)(b & 0x7f) << sh; sh += 7; if (!(b & 0x80)) break; }
            if ((tag >> 3) == 1) ++n;
            if (L > (uint64_t)(end - p)) break;
            p += L;
        } else break;                                                  // (groups, reserved wire types: the decoders turn the payload away)
    }
    return std::min(n, cap);
}
// the CPUs this process may really use: the affinity mask, capped by the cgroup's quota (cpu.max) — a container on a 256-thread host with a
// quota of 16 has 16
inline uint32_t wpl_usable_cpus() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0}; unsigned long long per = 0;
        if (fscanf(f, "%31s %llu", q, &per) == 2 && per && strcmp(q, "max") != 0) n = std::min<uint32_t>(n, (uint32_t)std::max(1ull, (strtoull(q, nullptr, 10) + per - 1) / per));
        fclose(f);
    }
    return n;
}
inline size_t wpl_put_varint(uint8_t* p, uint64_t v) { size_t n = 0; while (v >= 0x80) { p[n++] = (uint8_t)(v | 0x80); v >>= 7; } p[n++] = (uint8_t)v; return n; }
}  // namespace

struct guber_wire_pool {
    enum State : int { FREE = 0, OPEN, SEALED, DECODING, DECODED, ROUTING, EVALUATING, ANSWERED };
    static constexpr uint32_t RING = 32;
    struct Stage {
        guber_wire_dev* dec = nullptr;
        uint8_t* buf = nullptr;
        PinBuf<uint32_t> meta; PinBuf<uint8_t> owner;                     // offs[R] | lens[R]; is_owner[R] — written by the callers at their index
        std::vector<int32_t> status; std::vector<uint32_t> first, count;  // per RPC, after the decode
        CohBuf<uint8_t> res; guber_result_t r{};                          // the raw answers in arrival order (host memory the device writes in place): of the RPCs the device does not encode
        CohBuf<uint8_t> enc; CohBuf<uint32_t> enc_len;                    // every RPC's GetRateLimitsResp bytes (k_wire_enc): at wire_enc_off(first, idx), enc_len[idx] of them
        alignas(64) std::atomic<uint64_t> word{WPL_CLOSED};
        alignas(64) std::atomic<uint32_t> filled{0}; std::atomic<int64_t> t_first_us{0};
        alignas(64) std::atomic<uint32_t> done_gen{0xffffffffu}; std::atomic<uint32_t> sleepers{0};
        alignas(64) std::atomic<uint32_t> readers{0};
        // the pool's two threads' (a stage belongs to the intake thread until its routing is enqueued, then to the front thread)
        std::atomic<int> state{FREE}; uint32_t gen = 0, n_rpc = 0, n_items = 0; int rc = 0; int64_t now_ms = 0, t_seal_us = 0, t_decoded_us = 0;
    };
    int device = 0;
    std::vector<guber_engine*> eng;
    guber_front* front = nullptr;
    hipStream_t ws[2] = {nullptr, nullptr};            // the decodes' streams: consecutive stages alternate, so one stage's copy and latency-bound kernels run beside the other's
    uint32_t next_open = 0;
    uint32_t n_stages = 0, max_items = 0, max_b16 = 0, max_rpcs = 0, wait_us = 0, max_per_rpc = 0, item_cap = 0, spin_us = 0, decodes = 2;
    // the direct path of a lone one-request RPC (wpl_direct): the placement's rule as the HOST applies it
    struct HostRule { uint32_t n_shards = 1, per = 1, ex_cells = 0, ex_n = 0; int32_t global_engine = -1; uint64_t step = 0, inv_step = 0, inv_sub = 0;
                      std::vector<uint16_t> table, ex_shard; std::vector<uint64_t> ex_hash; } hrule;
    // callers inside the pool up to which a one-request RPC is evaluated by its caller (0: never).  Measured (profiles/r06_wire_direct.txt, 8 tables, p50 of a
    // 1-item RPC): 1 / 2 / 4 / 8 callers 12 / 15 / 23 / 38 us against 86 / 105 / 139 / 212 us through the stages, 0.13 against 0.04 M/s; from 16 callers on the
    // launches of many threads queue up in the runtime and on the engines' locks (p99 0.8 ms with 16, 2 ms with 32) while the stages' shared launches keep
    // p99 under 0.45 ms and overtake in throughput near 48 callers — so: up to 8
    uint32_t direct_max = 8;
    bool host_encode = false;                           // laboratory build only: the callers write the responses' varints themselves (round 6's first form, for A/B runs)
    std::unique_ptr<Stage[]> stages;
    alignas(64) std::atomic<uint32_t> open_word{WPL_NONE};               // (sequence << 4) | stage index (15: none): futex word of callers waiting for a stage
    std::atomic<uint32_t> open_waiters{0};
    alignas(64) std::atomic<uint32_t> inside{0}; uint32_t cpus = 1;          // callers inside the call: more of them than CPUs to look with -> they sleep at once
    alignas(64) std::atomic<uint32_t> wake{0}; std::atomic<uint32_t> intake_sleeping{0};
    alignas(64) std::atomic<uint32_t> wake2{0}; std::atomic<uint32_t> front_sleeping{0};
    uint32_t ring[RING] = {0}; alignas(64) std::atomic<uint32_t> ring_tail{0};
    alignas(64) std::atomic<uint64_t> evals_started{0}; std::atomic<uint64_t> handed{0}, freed{0}; std::atomic<bool> intake_done{false};
    std::atomic<bool> stop{false}, closed{false};
    std::atomic<int64_t> clock_ms{0};
    std::thread intake, fronter;
    uint32_t gen_counter = 0, open_seq = 0;
    // statistics (relaxed atomics: read by anybody)
    std::atomic<uint64_t> st_rpcs{0}, st_items{0}, st_stages{0}, st_full{0}, st_wait{0}, st_eager{0}, st_open_waits{0}, st_decode_us{0}, st_eval_us{0}, st_fill_us{0}, st_host_decode_ns{0}, st_host_route_ns{0}, st_host_eval_ns{0};
};

static int64_t wpl_now_ms(guber_wire_pool* p) {
    const int64_t c = p->clock_ms.load(std::memory_order_relaxed);
    if (c) return c;
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

static void wpl_open_next(guber_wire_pool* p) {
    for (uint32_t q = 0; q < p->n_stages; ++q) {
        const uint32_t k = (p->next_open + q) % p->n_stages;             // (round robin: consecutive stages sit on different decode streams)
        guber_wire_pool::Stage& s = p->stages[k];
        if (s.state.load(std::memory_order_acquire) != guber_wire_pool::FREE) continue;
        p->next_open = k + 1;
        s.state.store(guber_wire_pool::OPEN, std::memory_order_relaxed);
        s.gen = ++p->gen_counter & 0x7ffu;
        if (s.gen == s.done_gen.load(std::memory_order_relaxed)) s.gen = ++p->gen_counter & 0x7ffu;   // (2 048 openings later: the word its last callers waited for)
        s.filled.store(0, std::memory_order_relaxed); s.t_first_us.store(0, std::memory_order_relaxed);
        s.rc = 0;
        s.word.store(wpl_word(s.gen, 0, 0, 0), std::memory_order_release);
        p->open_word.store((++p->open_seq << 4) | k, std::memory_order_seq_cst);
        // (as many sleepers as the stage is likely to hold, not all of them: with several hundred callers the rest would find it full, go
        //  back to sleep and come back every timeout — measured: the wake-ups alone starve the pool's threads, 512 callers 10 M/s.  The next
        //  opening wakes the next lot; the sleepers' timeout is the belt)
        if (p->open_waiters.load(std::memory_order_seq_cst)) wpl_futex_wake(&p->open_word, (int)std::max(8u, p->max_items / std::max(1u, p->item_cap)));
        return;
    }
}

static void wpl_publish(guber_wire_pool* p, guber_wire_pool::Stage& s, int rc) {
    s.rc = rc;
    p->st_stages.fetch_add(1, std::memory_order_relaxed);
    s.readers.store(s.n_rpc, std::memory_order_relaxed);
    s.state.store(guber_wire_pool::ANSWERED, std::memory_order_relaxed);
    s.done_gen.store(s.gen, std::memory_order_seq_cst);
    if (s.sleepers.load(std::memory_order_seq_cst)) wpl_futex_wake(&s.done_gen, 2);   // (the woken wake two more each: guber_wire_pool_get_rate_limits)
}

static inline uint64_t wpl_ns_since(std::chrono::steady_clock::time_point t0) {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}
static void wpl_kick(guber_wire_pool* p) {
    p->wake.fetch_add(1, std::memory_order_seq_cst);
    if (p->intake_sleeping.load(std::memory_order_seq_cst)) wpl_futex_wake(&p->wake, 1);
}
static void wpl_kick_front(guber_wire_pool* p) {
    p->wake2.fetch_add(1, std::memory_order_seq_cst);
    if (p->front_sleeping.load(std::memory_order_seq_cst)) wpl_futex_wake(&p->wake2, 1);
}
static void wpl_hand_over(guber_wire_pool* p, uint32_t k) {               // intake -> front (single producer, single consumer; a stage is in the ring at most once)
    const uint32_t t = p->ring_tail.load(std::memory_order_relaxed);
    p->ring[t % guber_wire_pool::RING] = k;
    p->ring_tail.store(t + 1, std::memory_order_release);
    wpl_kick_front(p);
}
// every stage waits for these threads; among several hundred caller threads they should not wait their turn like one of them
// (best effort: needs CAP_SYS_NICE or root, and is not needed while the callers are fewer than the CPUs)
static void wpl_thread_setup(guber_wire_pool* p) {
    (void)hipSetDevice(p->device);
    (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), -10);
}

// The INTAKE thread: seals the open stage when it is due and opens the next, enqueues the decodes (copy + kernels, two stages deep), collects
// them and enqueues the front's routing; then the stage is the front thread's.  Never blocked on the GPU.
static void wpl_intake(guber_wire_pool* p) {
    using WP = guber_wire_pool;
    wpl_thread_setup(p);
    std::vector<uint32_t> decq;                                          // sealed stages on their way through the decode, oldest first
    uint64_t routes_done = 0;
    uint32_t idle_spins = 0;
    for (;;) {
        bool progress = false;
        const bool stopping = p->stop.load(std::memory_order_acquire);
        // ---- the open stage: seal it when it is due
        const uint32_t ow = p->open_word.load(std::memory_order_relaxed), oi = ow & 15u;
        auto seal = [&](WP::Stage& s, uint32_t idx, int why, int64_t now) {
            const uint64_t fin = s.word.fetch_or(WPL_CLOSED, std::memory_order_acq_rel);
            s.n_rpc = wpl_rpcs(fin); s.t_seal_us = now;
            s.state.store(WP::SEALED, std::memory_order_relaxed);
            decq.push_back(idx);
            p->open_word.store((++p->open_seq << 4) | WPL_NONE, std::memory_order_seq_cst);
            (why == 1 ? p->st_full : why == 2 ? p->st_wait : p->st_eager).fetch_add(1, std::memory_order_relaxed);
            const int64_t t1 = s.t_first_us.load(std::memory_order_relaxed);
            if (t1) p->st_fill_us.fetch_add((uint64_t)(now - t1), std::memory_order_relaxed);
        };
        if (oi != WPL_NONE) {
            WP::Stage& s = p->stages[oi];
            const uint64_t w = s.word.load(std::memory_order_acquire);
            if (wpl_rpcs(w)) {
                // (the decodes' streams are kept fed: a stage leaves as soon as fewer than `decodes` stages are waiting for / in their decode — the
                //  next one's copy and kernels then queue up behind the current one's instead of after a round trip through this thread)
                uint32_t decoding = 0;
                for (uint32_t k : decq) { const int st = p->stages[k].state.load(std::memory_order_relaxed); decoding += st == WP::SEALED || st == WP::DECODING; }
                const int64_t t1 = s.t_first_us.load(std::memory_order_relaxed);
                const int64_t now = wpl_mono_us();
                int why = 0;
                if (w & WPL_CLOSED) why = 1;
                else if (stopping || (t1 && now - t1 >= (int64_t)p->wait_us)) why = 2;
                else if (decoding < p->decodes) why = 3;
                if (why) { seal(s, oi, why, now); progress = true; }
            }
        }
        if ((p->open_word.load(std::memory_order_relaxed) & 15u) == WPL_NONE && !p->closed.load(std::memory_order_relaxed)) wpl_open_next(p);
        // ---- the sealed stages, oldest first; a stage never overtakes the one before it
        int prev = WP::ANSWERED;
        for (size_t q = 0; q < decq.size(); ++q) {
            WP::Stage& s = p->stages[decq[q]];
            int st = s.state.load(std::memory_order_relaxed);
            if (st == WP::SEALED && prev > WP::SEALED && s.filled.load(std::memory_order_acquire) == s.n_rpc) {
                s.now_ms = wpl_now_ms(p);
                const auto h0 = std::chrono::steady_clock::now();
                const int rc = guber_wire_dev_decode_staged_async(s.dec, s.meta.p, s.meta.p + p->max_rpcs, s.n_rpc, s.owner.p, p->max_per_rpc, s.now_ms);
                p->st_host_decode_ns.fetch_add(wpl_ns_since(h0), std::memory_order_relaxed);
                if (rc) { s.rc = rc; st = WP::DECODED; } else st = WP::DECODING;
                s.state.store(st, std::memory_order_relaxed);
                progress = true;
            }
            if (st == WP::DECODING && q == 0) {
                uint32_t n = 0;
                const int rc = guber_wire_dev_decode_collect(s.dec, 0, s.status.data(), s.first.data(), s.count.data(), &n);
                if (rc != GUBER_PENDING) {
                    s.n_items = n; s.rc = rc; s.t_decoded_us = wpl_mono_us();
                    p->st_decode_us.fetch_add((uint64_t)(s.t_decoded_us - s.t_seal_us), std::memory_order_relaxed);
                    st = WP::DECODED; s.state.store(st, std::memory_order_relaxed);
                    progress = true;
                }
            }
            // (the front routes ONE generation ahead of its evaluations: the next routing goes out when the front thread has enqueued the
            //  evaluation of the last one)
            if (st == WP::DECODED && q == 0 && (s.rc || routes_done == p->evals_started.load(std::memory_order_acquire))) {
                if (!s.rc) {
                    const auto h0 = std::chrono::steady_clock::now();
                    s.rc = guber_wire_dev_route_front_async(s.dec, p->front);
                    p->st_host_route_ns.fetch_add(wpl_ns_since(h0), std::memory_order_relaxed);
                }
                if (s.rc) wpl_publish(p, s, s.rc);                      // (the callers are told; the front thread frees the stage when they have left)
                else { ++routes_done; s.state.store(WP::ROUTING, std::memory_order_relaxed); }
                p->handed.fetch_add(1, std::memory_order_relaxed);
                wpl_hand_over(p, decq[q]);
                decq.erase(decq.begin());
                progress = true;
                break;                                                  // (the indices moved: next round)
            }
            prev = st;
        }
        if (stopping && decq.empty() && p->handed.load(std::memory_order_relaxed) == p->freed.load(std::memory_order_acquire)) {
            const uint32_t oi2 = p->open_word.load(std::memory_order_relaxed) & 15u;
            if (oi2 == WPL_NONE || wpl_rpcs(p->stages[oi2].word.load(std::memory_order_acquire)) == 0) {
                // nobody is inside: close the door (a caller that comes now finds the word closed and then the pool)
                bool slipped_in = false;
                if (oi2 != WPL_NONE) {
                    WP::Stage& s = p->stages[oi2];
                    uint64_t fin = s.word.load(std::memory_order_acquire);
                    if (wpl_rpcs(fin) || !s.word.compare_exchange_strong(fin, fin | WPL_CLOSED, std::memory_order_acq_rel)) slipped_in = true;
                }
                if (slipped_in) continue;                               // (somebody came in meanwhile: served like everybody else, next round)
                p->closed.store(true, std::memory_order_seq_cst);
                p->open_word.store((++p->open_seq << 4) | WPL_NONE, std::memory_order_seq_cst);
                wpl_futex_wake(&p->open_word, INT_MAX);
                p->intake_done.store(true, std::memory_order_seq_cst);
                wpl_kick_front(p);
                return;
            }
        }
        if (progress) { idle_spins = 0; continue; }
        // nothing moved: look again at once while a stage is on its way or payloads are arriving (never a yield: with more callers than CPUs
        // the thread would come back milliseconds later, and every stage waits for it); sleep when the pool is empty
        const uint32_t oi3 = p->open_word.load(std::memory_order_relaxed) & 15u;
        const bool quiet = oi3 != WPL_NONE && wpl_rpcs(p->stages[oi3].word.load(std::memory_order_relaxed)) == 0 && decq.empty();
        if (!quiet || ++idle_spins < 2000) { wpl_relax(); continue; }
        const uint32_t seen = p->wake.load(std::memory_order_seq_cst);
        p->intake_sleeping.store(1, std::memory_order_seq_cst);
        const uint32_t oi4 = p->open_word.load(std::memory_order_seq_cst) & 15u;
        if (!p->stop.load(std::memory_order_seq_cst) && oi4 != WPL_NONE && wpl_rpcs(p->stages[oi4].word.load(std::memory_order_seq_cst)) == 0)
            wpl_futex_wait(&p->wake, seen, 1000);
        p->intake_sleeping.store(0, std::memory_order_seq_cst);
        idle_spins = 0;
    }
}

// The FRONT thread: when a routed stage's shares' sizes are in host memory it enqueues the evaluation (the engines' fused launches and the
// answers' last hop, which writes them in place into the stage's host arrays), announces the stage when the GPU has got there, and frees it
// when its callers have taken their answers.  Never blocked on the GPU.
static void wpl_front(guber_wire_pool* p) {
    using WP = guber_wire_pool;
    wpl_thread_setup(p);
    std::vector<uint32_t> evq;
    uint32_t head = 0, idle_spins = 0;
    for (;;) {
        bool progress = false;
        for (uint32_t t = p->ring_tail.load(std::memory_order_acquire); head != t; ++head) { evq.push_back(p->ring[head % WP::RING]); progress = true; }
        for (size_t q = 0; q < evq.size();) {
            WP::Stage& s = p->stages[evq[q]];
            int st = s.state.load(std::memory_order_relaxed);
            if (st == WP::ROUTING && guber_wire_dev_route_ready(s.dec, p->front) != GUBER_PENDING) {
                const auto h0 = std::chrono::steady_clock::now();
                const int rc = p->host_encode ? guber_wire_dev_eval_front_async(s.dec, p->front, &s.r)
                                              : wire_dev_eval_front_enc_async(s.dec, p->front, s.enc.p, s.enc_len.p, &s.r);
                p->st_host_eval_ns.fetch_add(wpl_ns_since(h0), std::memory_order_relaxed);
                p->evals_started.fetch_add(1, std::memory_order_release);       // (also after a failure: the routing was consumed or given up)
                if (rc) { wpl_publish(p, s, rc); st = WP::ANSWERED; }
                else { st = WP::EVALUATING; s.state.store(st, std::memory_order_relaxed); }
                progress = true;
            }
            if (st == WP::EVALUATING) {
                const int rc = guber_wire_dev_eval_collect(s.dec, 0);
                if (rc != GUBER_PENDING) {
                    p->st_eval_us.fetch_add((uint64_t)(wpl_mono_us() - s.t_decoded_us), std::memory_order_relaxed);
                    p->st_items.fetch_add(s.n_items, std::memory_order_relaxed); p->st_rpcs.fetch_add(s.n_rpc, std::memory_order_relaxed);
                    wpl_publish(p, s, rc);
                    st = WP::ANSWERED;
                    progress = true;
                }
            }
            if (st == WP::ANSWERED && s.readers.load(std::memory_order_acquire) == 0) {
                s.state.store(WP::FREE, std::memory_order_release);
                p->freed.fetch_add(1, std::memory_order_release);
                evq.erase(evq.begin() + q);
                progress = true;
                continue;
            }
            ++q;
        }
        if (progress) { idle_spins = 0; continue; }
        if (!evq.empty() || ++idle_spins < 2000) { wpl_relax(); continue; }
        if (p->intake_done.load(std::memory_order_seq_cst) && head == p->ring_tail.load(std::memory_order_seq_cst)) return;
        const uint32_t seen = p->wake2.load(std::memory_order_seq_cst);
        p->front_sleeping.store(1, std::memory_order_seq_cst);
        if (head == p->ring_tail.load(std::memory_order_seq_cst) && !p->intake_done.load(std::memory_order_seq_cst)) wpl_futex_wait(&p->wake2, seen, 1000);
        p->front_sleeping.store(0, std::memory_order_seq_cst);
        idle_spins = 0;
    }
}

extern "C" void guber_wire_pool_destroy(guber_wire_pool_t* p) {
    if (!p) return;
    if (p->intake.joinable()) {
        p->stop.store(true, std::memory_order_seq_cst);
        wpl_kick(p);
        p->intake.join();
    }
    if (p->fronter.joinable()) { wpl_kick_front(p); p->fronter.join(); }
    (void)hipSetDevice(p->device);
    if (p->stages) for (uint32_t k = 0; k < p->n_stages; ++k) {
        guber_wire_pool::Stage& s = p->stages[k];
        if (s.dec) guber_wire_dev_destroy(s.dec);
        s.meta.release(); s.owner.release(); s.res.release(); s.enc.release(); s.enc_len.release();
    }
    if (p->front) guber_front_destroy(p->front);
    for (hipStream_t st : p->ws) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    delete p;
}

extern "C" int guber_wire_pool_create(guber_engine_t* const* engines, uint32_t n_engines, const guber_route_rule_t* rule, const guber_wire_pool_config_t* cfg,
                                      guber_wire_pool_t** out) {
    if (!engines || !n_engines || !out) return fail(GUBER_E_INVALID_ARG, "null argument");
    *out = nullptr;
    guber_wire_pool_config_t c{};
    if (cfg) c = *cfg;
    // (stages no larger than what the front evaluates as ONE pair of launches for all tables, FRONT_ONE_PAIR_MAX, and enough of them in rotation
    //  for a few hundred callers: measured against six stages of 131 072 items — 192 callers 258-363 -> 398-413 M/s, 256 callers 301-311 ->
    //  375-382, 384 callers 247-259 -> 370-373: profiles/r06_wire_pool.txt)
    if (!c.stages) c.stages = 12;
    if (!c.max_items) c.max_items = 49152;
    if (!c.max_payload_bytes) c.max_payload_bytes = 8u << 20;
    if (!c.max_rpcs) c.max_rpcs = 1024;
    if (!c.batch_wait_us) c.batch_wait_us = 500;                       // config.go:131 BatchWait
    if (!c.max_per_rpc) c.max_per_rpc = 1000;                          // gubernator.go:40
    if (!c.spin_us) c.spin_us = 150;
    if (!c.decodes_queued) c.decodes_queued = 2;
    if (c.stages < 2 || c.stages > WPL_MAX_STAGES || c.max_items > WPL_MAX_ITEMS || c.max_rpcs > WPL_MAX_RPCS || c.max_payload_bytes / 16 > WPL_MAX_B16 || c.max_payload_bytes < 64)
        return fail(GUBER_E_INVALID_ARG, "guber_wire_pool: 2 .. 12 stages of at most 1 048 575 items, 4 095 RPCs and 16 MiB of payload bytes");
    std::unique_ptr<guber_wire_pool, void (*)(guber_wire_pool*)> p(new guber_wire_pool(), [](guber_wire_pool* q) { guber_wire_pool_destroy(q); });
    for (uint32_t j = 0; j < n_engines; ++j) { if (!engines[j]) return fail(GUBER_E_INVALID_ARG, "null engine"); p->eng.push_back(engines[j]); }
    p->device = engines[0]->device;
    p->n_stages = c.stages; p->max_items = c.max_items; p->max_b16 = c.max_payload_bytes / 16; p->max_rpcs = c.max_rpcs; p->wait_us = c.batch_wait_us;
    p->max_per_rpc = c.max_per_rpc == 0xffffffffu ? 0 : c.max_per_rpc; p->spin_us = c.spin_us; p->decodes = c.decodes_queued;
    p->cpus = wpl_usable_cpus();
    if (rule && n_engines > 1) {
        if (rule->n_shards == 0 || rule->per == 0 || !rule->table || (rule->ex_cells & (rule->ex_cells - 1)) || (rule->ex_n && (!rule->ex_hash || !rule->ex_shard || rule->ex_n >= rule->ex_cells)))
            return fail(GUBER_E_INVALID_ARG, "malformed route rule");
        guber_wire_pool::HostRule& H = p->hrule;
        H.n_shards = rule->n_shards; H.per = rule->per; H.ex_cells = rule->ex_cells; H.ex_n = rule->ex_n; H.global_engine = rule->global_engine;
        H.step = rule->step; H.inv_step = rule->inv_step; H.inv_sub = rule->inv_sub;
        H.table.assign(rule->table, rule->table + (size_t)rule->n_shards * rule->per);
        if (rule->ex_n) { H.ex_hash.assign(rule->ex_hash, rule->ex_hash + rule->ex_cells); H.ex_shard.assign(rule->ex_shard, rule->ex_shard + rule->ex_cells); }
    }
    p->item_cap = std::min<uint32_t>(std::min<uint32_t>(c.max_items, 4096u), p->max_per_rpc ? p->max_per_rpc : 4096u);   // (an RPC holds at most min(max_items, 4096) items: guber_wire_dev_create)
    {
        const int rc = guber_front_create(engines, n_engines, rule, c.max_items, 4, &p->front);
        if (rc) return rc;
    }
    if (hipSetDevice(p->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
#ifdef GUBER_LAB
    {   // (laboratory knob: the front's routing on the engines' one stream, which frees a hardware queue for the second decode stream:
        //  profiles/r06_wire_pool_hw_queues.txt — faster with 256 callers, slower with fewer)
        if (const char* dv = guber_lab_env("GUBER_WIRE_DIRECT")) p->direct_max = (uint32_t)atoi(dv);
        const char* he = guber_lab_env("GUBER_WIRE_HOST_ENCODE");
        p->host_encode = he && atoi(he) != 0;
        const char* v = guber_lab_env("GUBER_WIRE_ROUTE_ON_ENGINES");
        bool one = true;
        for (uint32_t j = 1; j < n_engines; ++j) one = one && engines[j]->stream == engines[0]->stream;
        if (v && atoi(v) != 0 && one) { const int rc = front_route_on(p->front, engines[0]->stream); if (rc) return rc; }
    }
#endif
    HIPCHK(hipStreamCreateWithFlags(&p->ws[0], hipStreamNonBlocking));
    // (both decode streams land on ONE of the runtime's hardware queues, ~75 % busy at 400 M/s; a second one at another priority or GPU_MAX_HW_QUEUES > 4
    //  gives it a queue of its own — a FIFTH queue, and the rate falls to a quarter: profiles/r06_wire_pool_hw_queues.txt)
    if (c.decodes_queued > 1) HIPCHK(hipStreamCreateWithFlags(&p->ws[1], hipStreamNonBlocking));
    p->stages.reset(new guber_wire_pool::Stage[c.stages]);
    const size_t M = c.max_items, R = c.max_rpcs;
    for (uint32_t k = 0; k < c.stages; ++k) {
        guber_wire_pool::Stage& s = p->stages[k];
        int rc = guber_wire_dev_create(engines[k % n_engines], c.max_items, c.max_payload_bytes, c.max_rpcs, &s.dec);
        if (rc) return rc;
        rc = guber_wire_dev_set_stream(s.dec, p->ws[1] ? p->ws[k & 1] : p->ws[0]);
        if (rc) return rc;
        size_t cap = 0;
        rc = guber_wire_dev_buffer(s.dec, &s.buf, &cap);
        if (rc) return rc;
        const size_t col8 = (M + 63) & ~(size_t)63;
        if (s.meta.ensure(2 * R) || s.owner.ensure(R) || s.res.ensure(2 * col8 + 3 * 8 * M + 64) || s.enc.ensure(guber::wire_enc_bytes(c.max_items, c.max_rpcs)) || s.enc_len.ensure(R))
            return GUBER_E_NOMEM;
        s.status.resize(R); s.first.resize(R); s.count.resize(R);
        uint8_t* q = s.res.p;
        s.r = guber_result_t{};
        s.r.status = q; q += col8; s.r.err = q; q += col8;
        s.r.limit = (int64_t*)q; q += 8 * M; s.r.remaining = (int64_t*)q; q += 8 * M; s.r.reset_time = (int64_t*)q;
    }
    wpl_open_next(p.get());
    p->intake = std::thread(wpl_intake, p.get());
    p->fronter = std::thread(wpl_front, p.get());
    *out = p.release();
    return GUBER_OK;
}

extern "C" int guber_wire_pool_set_clock(guber_wire_pool_t* p, int64_t now_ms) {
    if (!p) return fail(GUBER_E_INVALID_ARG, "null argument");
    p->clock_ms.store(now_ms, std::memory_order_relaxed);
    return GUBER_OK;
}

// ---- an RPC of at most WPL_DIRECT_ITEMS (four) requests while the pool is nearly idle (the reference's BenchmarkServer shape, benchmark_test.go:63-84; a lightly loaded daemon's usual call):
// hardly anybody to share a stage with, and a stage's way through the GPU is a dozen launches (~90 us).  While at most direct_max calls are inside the pool the
// caller evaluates it itself: the host
// transcoder (wire.cpp) parses the payload into a batch of the thread's own (pinned), the placement's rule picks the table exactly as k_fr_count does — XXH64 of
// the HashKey, individually placed keys first, then slot -> shard; Behavior_GLOBAL to the GLOBAL engine —, guber_eval_batch takes the engine's one-launch
// path in place (k_small: ~12 us), the host transcoder writes the response.  Same bytes (tests/test_gpu_wire_pool.py), no stage, no pool thread involved.
// With more than direct_max callers inside, everybody goes through the stages again and shares launches.
static uint32_t wpl_route_host(const guber_wire_pool::HostRule& R, uint64_t h) {   // = route_engine (guber_kernels_route.h), on the host
    if (R.ex_n) {
        for (uint32_t i = (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 56) & (R.ex_cells - 1);; i = (i + 1) & (R.ex_cells - 1)) {
            const uint64_t x = R.ex_hash[i];
            if (x == h) return R.ex_shard[i];
            if (x == 0ull) break;
        }
    }
    auto mulhi = [](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); };
    const uint64_t h63 = h >> 1;
    uint64_t w = mulhi(h63, R.inv_step);
    if ((w + 1) * R.step <= h63) ++w;
    if (w >= R.n_shards) w = R.n_shards - 1;
    uint64_t sub = mulhi(h63 - w * R.step, R.inv_sub);
    if (sub >= R.per) sub = R.per - 1;
    return R.table[(size_t)(w * R.per + sub)];
}
namespace {
struct WplThreadBatch { guber_wire_batch_t* wb = nullptr; bool failed = false; ~WplThreadBatch() { if (wb) guber_wire_batch_destroy(wb); } };
constexpr size_t WPL_DIRECT_MAX_BYTES = 2048;
constexpr uint32_t WPL_DIRECT_ITEMS = 4;      // requests an RPC may hold to be evaluated by its caller (the per-request pool's GUBER_POOL_DIRECT_MAX)
}
// returns true when the call has been answered (rc, used); false: not for this path, nothing has happened
static bool wpl_direct(guber_wire_pool* p, const uint8_t* req, size_t len, int is_owner, int wrap_errors, uint8_t* resp, size_t cap, size_t* used, int* rc_out) {
    thread_local WplThreadBatch tl;
    if (tl.failed) return false;
    if (!tl.wb) {
        (void)hipSetDevice(p->device);
        if (guber_wire_batch_create(4, (uint32_t)(2 * WPL_DIRECT_MAX_BYTES), GUBER_WIRE_PINNED, &tl.wb) != GUBER_OK) { tl.wb = nullptr; tl.failed = true; return false; }
    }
    guber_wire_batch_reset(tl.wb, wpl_now_ms(p));
    uint32_t f0 = 0, c0 = 0;
    int rc = guber_wire_decode_requests(tl.wb, req, len, p->max_per_rpc, is_owner ? 1 : 0, &f0, &c0);
    if (rc == GUBER_E_WIRE_FULL || (rc == GUBER_OK && c0 > WPL_DIRECT_ITEMS)) return false;                  // (not what the bound promised: the stages take it)
    *used = 0;
    if (rc == GUBER_OK && c0 >= 1) {
        const guber_batch_t* v = guber_wire_batch_view(tl.wb);
        guber_result_t* r = guber_wire_batch_result(tl.wb);
        const uint32_t ne = (uint32_t)p->eng.size();
        uint32_t eng[WPL_DIRECT_ITEMS] = {0, 0, 0, 0};
        bool same = true;
        if (ne > 1) {
            for (uint32_t i = 0; i < c0; ++i) {
                const uint32_t klen = v->key_off[i + 1] - v->key_off[i];
                uint32_t e = i ? eng[i - 1] : 0u;                     // (an item without a key never reaches a bucket: it goes along with its neighbour)
                if (klen != 0) {
                    if (p->hrule.global_engine >= 0 && v->behavior && (v->behavior[i] & 2u)) e = (uint32_t)p->hrule.global_engine;
                    else if (p->hrule.n_shards > 1) e = wpl_route_host(p->hrule, guber_xxhash64(v->key_bytes + v->key_off[i], klen, 0));
                    else e = 0;
                    if (e >= ne) e = 0;
                }
                eng[i] = e;
                same = same && e == eng[0];
            }
        }
        if (same) rc = guber_eval_batch(p->eng[eng[0]], v, r);         // one launch for the RPC
        else for (uint32_t i = 0; i < c0 && rc == GUBER_OK; ++i) {      // its requests live on several tables: one after the other, in the RPC's order
            guber_batch_t sv = *v; guber_result_t sr = *r;
            sv.n = 1; sv.key_off = v->key_off + i; sv.hits = v->hits + i; sv.limit = v->limit + i; sv.duration = v->duration + i;
            if (v->burst) sv.burst = v->burst + i;
            if (v->created_at) sv.created_at = v->created_at + i;
            sv.algorithm = v->algorithm + i; sv.behavior = v->behavior + i;
            if (v->is_owner) sv.is_owner = v->is_owner + i;
            if (v->greg_expire) sv.greg_expire = v->greg_expire + i;
            if (v->greg_duration) sv.greg_duration = v->greg_duration + i;
            sr.status = r->status + i; sr.limit = r->limit + i; sr.remaining = r->remaining + i; sr.reset_time = r->reset_time + i; sr.err = r->err + i;
            rc = guber_eval_batch(p->eng[eng[i]], &sv, &sr);
        }
        if (rc == GUBER_OK) {
            rc = guber_wire_encode_responses(tl.wb, 0, c0, wrap_errors, resp, cap, used);
            if (rc == GUBER_E_NOMEM) fail(GUBER_E_NOMEM, "response buffer too small (the decisions HAVE been applied): guber_wire_pool_response_bound()");
        }
        p->st_items.fetch_add(c0, std::memory_order_relaxed);
    } else if (rc != GUBER_OK) fail(rc, "guber_wire_pool: the message is turned away whole");
    // (a batch of its own: counted among the stages that left because nothing else was there)
    p->st_rpcs.fetch_add(1, std::memory_order_relaxed); p->st_stages.fetch_add(1, std::memory_order_relaxed); p->st_eager.fetch_add(1, std::memory_order_relaxed);
    *rc_out = rc;
    return true;
}

extern "C" size_t guber_wire_pool_response_bound(const uint8_t* req, size_t len) {
    // per item: tag + length + four varint fields (37 bytes), or an error: wrapper text + message (<= 310 bytes) + its key (the keys of a payload: <= len + one '_' each)
    const size_t items = req ? wpl_item_bound(req, len, 4096) : 0;
    return items * 348 + len + 16;
}

extern "C" int guber_wire_pool_stats(guber_wire_pool_t* p, guber_wire_pool_stats_t* o) {
    if (!p || !o) return fail(GUBER_E_INVALID_ARG, "null argument");
    o->rpcs = p->st_rpcs.load(); o->items = p->st_items.load(); o->stages = p->st_stages.load(); o->sealed_full = p->st_full.load(); o->sealed_wait = p->st_wait.load();
    o->sealed_idle = p->st_eager.load(); o->open_waits = p->st_open_waits.load(); o->decode_us_sum = p->st_decode_us.load(); o->eval_us_sum = p->st_eval_us.load();
    o->fill_us_sum = p->st_fill_us.load();
    o->host_decode_ns = p->st_host_decode_ns.load(); o->host_route_ns = p->st_host_route_ns.load(); o->host_eval_ns = p->st_host_eval_ns.load();
    return GUBER_OK;
}

// V1Instance.GetRateLimits / GetPeerRateLimits on the serialized messages (gubernator.go:183-306, :470-520 for the items this instance owns).
extern "C" int guber_wire_pool_get_rate_limits(guber_wire_pool_t* p, const uint8_t* req, size_t len, int is_owner, int wrap_errors, uint8_t* resp, size_t cap,
                                               size_t* resp_len) {
    using WP = guber_wire_pool;
    if (!p || (!req && len) || !resp_len || (!resp && cap)) return fail(GUBER_E_INVALID_ARG, "null argument");
    *resp_len = 0;
    if (len == 0) return GUBER_OK;                                     // no requests: an empty response
    const uint32_t units = (uint32_t)((len + 15) / 16);
    if (len > (size_t)p->max_b16 * 16) return fail(GUBER_E_WIRE_FULL, "payload larger than a stage");
    const uint32_t bound = std::max(1u, wpl_item_bound(req, len, p->item_cap));
    if (cap < (size_t)bound * 37 && cap < guber_wire_pool_response_bound(req, len)) { *resp_len = guber_wire_pool_response_bound(req, len); return fail(GUBER_E_NOMEM, "response buffer below guber_wire_pool_response_bound()"); }
    struct Inside { std::atomic<uint32_t>& c; uint32_t n; explicit Inside(std::atomic<uint32_t>& x) : c(x), n(x.fetch_add(1, std::memory_order_relaxed) + 1) {} ~Inside() { c.fetch_sub(1, std::memory_order_relaxed); } } inside(p->inside);
    if (bound <= WPL_DIRECT_ITEMS && inside.n <= p->direct_max && len <= WPL_DIRECT_MAX_BYTES && !p->closed.load(std::memory_order_acquire)) {
        size_t used = 0; int rc = GUBER_OK;
        if (wpl_direct(p, req, len, is_owner, wrap_errors, resp, cap, &used, &rc)) { *resp_len = used; return rc; }
    }
    // looking (spinning) is for callers that have a CPU to themselves: the pool's two threads need theirs, the others sleep at once
    const bool may_spin = p->spin_us && inside.n + 1 <= p->cpus / 2;
    // ---- a place in the open stage
    WP::Stage* sp = nullptr;
    uint64_t mine = 0;
    for (uint32_t tries = 0;; ++tries) {
        const uint32_t ow = p->open_word.load(std::memory_order_acquire), oi = ow & 15u;
        bool must_wait = oi == WPL_NONE;
        if (!must_wait) {
            WP::Stage& s = p->stages[oi];
            uint64_t w = s.word.load(std::memory_order_acquire);
            if (w & WPL_CLOSED) must_wait = true;
            else if (wpl_rpcs(w) + 1 > p->max_rpcs || wpl_items(w) + bound > p->max_items || wpl_b16(w) + units > p->max_b16) {
                if (wpl_rpcs(w) == 0) return fail(GUBER_E_WIRE_FULL, "payload larger than a stage");
                s.word.fetch_or(WPL_CLOSED, std::memory_order_acq_rel);   // full: the intake thread seals it and opens the next
                wpl_kick(p);
                must_wait = true;
            } else if (s.word.compare_exchange_weak(w, w + wpl_word(0, 1, bound, units), std::memory_order_acq_rel, std::memory_order_acquire)) {
                sp = &s; mine = w;
                break;
            } else continue;
        }
        if (p->closed.load(std::memory_order_acquire)) return fail(GUBER_E_WIRE_CLOSED, "guber_wire_pool: closed");
        // no stage is open (all of them are on their way through the GPU, or the intake thread is about to open the next): look, then sleep
        if (tries < (may_spin ? 2000u : 20u)) { wpl_relax(); continue; }
        p->st_open_waits.fetch_add(1, std::memory_order_relaxed);
        p->open_waiters.fetch_add(1, std::memory_order_seq_cst);
        if (p->open_word.load(std::memory_order_seq_cst) == ow) wpl_futex_wait(&p->open_word, ow, 5000);
        p->open_waiters.fetch_sub(1, std::memory_order_seq_cst);
    }
    WP::Stage& s = *sp;
    const uint32_t idx = wpl_rpcs(mine), gen = wpl_gen(mine);
    const size_t off = (size_t)wpl_b16(mine) * 16;
    if (idx == 0) { s.t_first_us.store(wpl_mono_us(), std::memory_order_relaxed); wpl_kick(p); }
    memcpy(s.buf + off, req, len);
    s.meta.p[idx] = (uint32_t)off; s.meta.p[p->max_rpcs + idx] = (uint32_t)len; s.owner.p[idx] = is_owner ? 1 : 0;
    s.filled.fetch_add(1, std::memory_order_release);
    // ---- the stage's way through the GPU
    {
        uint32_t v;
        const int64_t t0 = may_spin ? wpl_mono_us() : 0;
        uint32_t spins = 0;
        while ((v = s.done_gen.load(std::memory_order_acquire)) != gen) {
            if (may_spin && (++spins & 63u || wpl_mono_us() - t0 < (int64_t)p->spin_us)) { wpl_relax(); continue; }
            s.sleepers.fetch_add(1, std::memory_order_seq_cst);
            while ((v = s.done_gen.load(std::memory_order_seq_cst)) != gen) wpl_futex_wait(&s.done_gen, v, -1);
            if (s.sleepers.fetch_sub(1, std::memory_order_seq_cst) > 1) wpl_futex_wake(&s.done_gen, 2);
            break;
        }
    }
    // ---- my slice of the answers -> GetRateLimitsResp
    int rc = s.rc;
    size_t used = 0;
    if (!rc) {
        const int32_t st = s.status[idx];
        if (st != GUBER_OK) rc = st;                                   // GUBER_E_WIRE_MALFORMED / GUBER_E_WIRE_TOO_LARGE: the whole message is turned away
        else {
            const uint32_t first = s.first[idx], count = s.count[idx];
            // (a message without a single RateLimitReq — unknown fields only — has no answers; when nothing of its stage had any, no kernel wrote enc_len)
            const uint32_t el = p->host_encode ? guber::WIRE_ENC_RAW : count ? s.enc_len.p[idx] : 0u;
            bool any_err = !p->host_encode;                            // (not encoded on the device: an item carries an error, the raw answers are in s.r)
            if (p->host_encode) { const uint8_t* e8 = s.r.err + first; for (uint32_t i = 0; i < count; ++i) any_err = any_err || e8[i] != GUBER_ITEM_OK; }
            if (el != guber::WIRE_ENC_RAW) {
                // the device has written this RPC's GetRateLimitsResp (k_wire_enc): one memcpy.  (cap >= el: the check at the door — the
                // response bound, or 37 bytes per item, which is the most an item without an error takes)
                if (cap < el) rc = fail(GUBER_E_NOMEM, "response buffer too small (the decisions HAVE been applied): guber_wire_pool_response_bound()");
                else { memcpy(resp, s.enc.p + guber::wire_enc_off(first, idx), el); used = el; }
            } else if (!any_err && cap >= (size_t)count * 37) {     // (laboratory build, GUBER_WIRE_HOST_ENCODE=1: the caller writes the varints)
                const uint8_t* st8 = s.r.status + first; const int64_t* lim = s.r.limit + first; const int64_t* rem = s.r.remaining + first; const int64_t* rst = s.r.reset_time + first;
                uint8_t* o = resp;
                for (uint32_t i = 0; i < count; ++i) {
                    uint8_t* body = o + 2;                             // (a body is at most 35 bytes: its length is one byte)
                    size_t b = 0;
                    if (st8[i]) { body[b++] = 0x08; b += wpl_put_varint(body + b, st8[i]); }
                    if (lim[i]) { body[b++] = 0x10; b += wpl_put_varint(body + b, (uint64_t)lim[i]); }
                    if (rem[i]) { body[b++] = 0x18; b += wpl_put_varint(body + b, (uint64_t)rem[i]); }
                    if (rst[i]) { body[b++] = 0x20; b += wpl_put_varint(body + b, (uint64_t)rst[i]); }
                    o[0] = 0x0a; o[1] = (uint8_t)b;
                    o += 2 + b;
                }
                used = (size_t)(o - resp);
            } else {
                // an item with an error (or a buffer too small for the plain form): the texts need the item's key and raw algorithm — the payload is
                // parsed once more, here, by the host transcoder (wire.cpp), and the answers are encoded through it
                guber_wire_batch_t* wb = nullptr;
                rc = guber_wire_batch_create(std::max(1u, count), (uint32_t)std::min<size_t>(len + count + 16, 0xffffffffu), 0, &wb);
                if (!rc) {
                    guber_wire_batch_reset(wb, s.now_ms);
                    uint32_t f0 = 0, c0 = 0;
                    rc = guber_wire_decode_requests(wb, req, len, p->max_per_rpc, is_owner ? 1 : 0, &f0, &c0);
                    if (!rc && c0 != count) rc = fail(GUBER_E_HIP, "guber_wire_pool: the device's and the host's decoders disagree about a payload");
                    if (!rc) {
                        guber_result_t* hr = guber_wire_batch_result(wb);
                        memcpy(hr->status, s.r.status + first, count); memcpy(hr->err, s.r.err + first, count);
                        memcpy(hr->limit, s.r.limit + first, (size_t)count * 8); memcpy(hr->remaining, s.r.remaining + first, (size_t)count * 8);
                        memcpy(hr->reset_time, s.r.reset_time + first, (size_t)count * 8);
                        rc = guber_wire_encode_responses(wb, 0, count, wrap_errors, resp, cap, &used);
                        if (rc == GUBER_E_NOMEM) fail(GUBER_E_NOMEM, "response buffer too small (the decisions HAVE been applied): guber_wire_pool_response_bound()");
                    }
                    guber_wire_batch_destroy(wb);
                }
            }
        }
    } else {
        fail(rc, "guber_wire_pool: the stage failed on the device");
    }
    s.readers.fetch_sub(1, std::memory_order_release);                // (the stage is not touched after this)
    *resp_len = used;
    return rc;
}
