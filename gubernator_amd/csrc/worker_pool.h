// worker_pool.h — C++ host layer above the C ABI, mirroring the reference's call surface for the path:
//   gubernator::GPUWorkerPool   <->  WorkerPool            workers.go:54-626
//   gubernator::V1Instance      <->  V1Instance.GetRateLimits (local-owner slice)  gubernator.go:183-306
// Same names, argument meaning and error behaviour; what changes is the mechanism.  The reference hands every request to
// its worker's goroutine through a channel (workers.go:261-291); here the CALLERS do the per-request work, in parallel:
// a caller hashes its keys, looks up the shard of each (guber_placement: hash slots + individually placed hot keys),
// reserves a contiguous range of request slots (and key bytes) in the open stage of the key's DEVICE — its front — with ONE
// compare-and-swap per RPC, takes its requests' places in their shards' shares from the stage's per-shard counters, writes
// its requests IN PLACE into the stage's arrays (device-visible host memory, include/guber_gpu.h guber_stage_*) together
// with `shard << 24 | place` per request, and later reads its responses straight out of the stage's result arrays.  The GPU
// hands the requests to the shards: the copy kernel that brings a stage to HBM places every shard's share contiguously
// (guber_stage_submit_routed).  GUBER_POOL_ROUTED=0 (or more than 16 engines per device): every shard has stages of its own,
// the callers sort their requests by shard and pay one compare-and-swap per RPC and shard.
//
// ONE dispatcher thread per device (round 2 had a batcher thread and stream per shard: the launches of different streams
// overlap badly on the GPU and the threads fought for the host).  It seals the stage when it is due — as soon as the device
// has room (the reference's workers take a request the moment it arrives; a batch then collects what arrives while the
// previous one runs, so batches grow with the load by themselves), at the stage's capacity (batch_limit per shard), at
// batch_wait after the first reservation (the policy of peer_client.go:284-337), when a caller found no room — and submits
// it: four launches for a generation whatever the number of shards, ONE for a generation of <= 256 requests, nothing is
// waited for.  It polls for completions (guber_stage_poll), announces finished generations (futex; wake-ups fan out as a
// tree), and never touches a request.
// Nothing is allocated per flush.  Lightly loaded (at most shards/2 calls in progress) an RPC of a handful of requests does
// not travel through stages at all: its caller evaluates it through guber_eval_batch and answers.
// The host's CPUs are the pool's bottleneck, so it counts them (a cgroup CPU quota included): at most that many callers are
// in the CPU part of a call at a time, and callers of large batches sleep at once instead of spinning.
//
// Placement: `devices` x `shards_per_device`.  A key's device is its owner on the reference's replicated consistent hash
// over the peers "gpu0".."gpuN-1" (replicated_hash.go:78-119, 512 vnodes, fnv1) — the N GPUs of a node are N peers.
// Inside a device the shard comes from the device's guber_placement_t: XXH64-range slots whose initial table IS the
// reference's worker rule (workers.go:153-155,180-184), plus the keys that turn out to carry a large share of the traffic,
// observed online (every 64th request feeds the placement's sketch) and isolated on the least loaded shard.  Moving a
// resident key is done by the dispatcher at a batch boundary: every open stage of the device is sealed, the batches in
// flight drain, the key's bucket is taken from the old shard's table and added to the new one's, the new exception list is
// published and the stages reopen carrying the new placement version — a caller that routed with the old version cannot
// reserve in them (the version is part of the word its compare-and-swap expects) and routes again.  Per-key request order
// is therefore kept across a move.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/guber_gpu.h"

namespace gubernator {

struct RateLimitReq {            // gubernator.proto:137-182
    std::string name, unique_key;
    int64_t hits = 0, limit = 0, duration = 0, burst = 0;
    int32_t algorithm = 0;
    uint32_t behavior = 0;
    int64_t created_at = 0;      // 0 = unset (optional field)
    std::string HashKey() const { return name + "_" + unique_key; }   // client.go:39-41
};
struct RateLimitResp {           // gubernator.proto:189-203
    int32_t status = 0;
    int64_t limit = 0, remaining = 0, reset_time = 0;
    std::string error;
};
struct RateLimitReqState { bool is_owner = true; };   // workers.go / gubernator.go:246

constexpr uint32_t kMaxBatchSize = 1000;              // gubernator.go:40

class GPUWorkerPool {
 public:
    // `shards` = Config.Workers of the reference (config.go:110, workers.go:125-151) per device; cfg.cache_size is per pool, as
    // in the reference (each shard gets cache_size / shards, workers.go:132).  devices empty = {cfg.device}.  With
    // GUBER_FLAG_GLOBAL in cfg.flags every device gets one more engine that holds the keys of GLOBAL-behaviour requests (the
    // replica the GLOBAL manager synchronises: GlobalSync()); the other engines are created without the flag.
    GPUWorkerPool(const guber_config_t& cfg, uint32_t batch_limit, uint32_t batch_wait_us, uint32_t shards = 1,
                  const std::vector<int32_t>& devices = {});
    ~GPUWorkerPool();
    bool ok() const { return !shards_.empty() && create_rc_ == 0; }
    int create_error() const { return create_rc_; }

    // WorkerPool.GetRateLimit (workers.go:261): blocks until the request's batch has been evaluated.
    // Returns false with resp->error set when the reference would return an error.
    bool GetRateLimit(const RateLimitReq& r, RateLimitReqState st, RateLimitResp* resp);
    // Submit many requests, answer all (used by V1Instance::GetRateLimits; order of same-key requests is kept; a mixed batch
    // is split by owning device / shard and the answers land in the callers' slots: functional_test.go:1638-1686).
    void GetRateLimitMany(const std::vector<const RateLimitReq*>& reqs, const std::vector<RateLimitReqState>& st,
                          std::vector<RateLimitResp*>& out);
    // The same for a binding's structure-of-arrays (names / unique keys as packed strings + offsets, numeric columns), with the
    // V1Instance.GetRateLimits front end folded in (1000-item cap, empty-field errors, CreatedAt default, error wrapping):
    // include/guber_gpu.h guber_pool_get_rate_limits.  No per-request allocation, no intermediate objects.
    int GetRateLimitsSoA(uint32_t n, const uint8_t* name_bytes, const uint32_t* name_off, const uint8_t* ukey_bytes, const uint32_t* ukey_off,
                         const int64_t* hits, const int64_t* limit, const int64_t* duration, const int64_t* burst, const int64_t* created_at,
                         const int32_t* algorithm, const uint32_t* behavior, guber_result_t* out, char* err_text, uint32_t err_stride,
                         const uint8_t* is_owner = nullptr);
    int AddCacheItem(const guber_item_t& item, int behavior = -1);               // workers.go:537 (behavior: see worker_pool.cpp)
    int GetCacheItem(const std::string& key, guber_item_t* out, bool* found);    // workers.go:583
    // WorkerPool.Load (workers.go:329-449): hand every item of a Loader to the cache; WorkerPool.Store (workers.go:451-534):
    // visit every resident item (what Loader.Save receives).  Bulk paths: guber_add_items / guber_dump.
    int Load(const guber_item_t* items, uint32_t n, const uint8_t* global_hint = nullptr);
    int Store(const std::function<void(const guber_item_t&)>& save);
    int64_t Size();
    void Close();                                                                // workers.go:157
    // Config.Store (config.go:99, store.go:49-65): call before the first request; nullptr = none
    void SetStore(const guber_store_callbacks_t* cb) { if (cb) store_ = *cb; has_store_.store(cb != nullptr); }
    // clock.Freeze / clock.Advance of the reference's tests: 0 = wall clock
    void SetClockMs(int64_t now_ms) { frozen_ms_.store(now_ms); }
    int64_t NowMs() const;
    guber_engine_t* engine(uint32_t shard = 0) { return shard < shards_.size() ? shards_[shard]->engine : nullptr; }
    uint32_t shards() const { return (uint32_t)shards_.size(); }
    uint32_t devices() const { return n_devices_; }
    // the device a key belongs to (ReplicatedConsistentHash.Get, replicated_hash.go:104-119) and its shard index
    // (device * shards_per_device + the placement's shard: WorkerPool.getWorker generalised, workers.go:180-184)
    uint32_t DeviceOf(const uint8_t* key, uint32_t len) const;
    uint32_t ShardOf(const std::string& key) const { return ShardOf((const uint8_t*)key.data(), (uint32_t)key.size()); }
    uint32_t ShardOf(const uint8_t* key, uint32_t len, uint32_t behavior = 0) const;
    uint64_t batches_flushed() const;
    void Metrics(guber_pool_metrics_t* out) const;   // gubernator_batch_queue_length / gubernator_batch_send_duration analogues (gubernator.go:96-107)
    // one GlobalSyncWait tick over the devices' GLOBAL engines (global.go:91-283 natively: guber_global_sync)
    int GlobalSync(guber_global_sync_stats_t* stats);
    guber_engine_t* GlobalEngine(uint32_t device);   // for a communicator of one's own (guber_comm_create_rank: one rank per daemon)
    // ask the dispatchers for a placement pass now (tests, operators); the periodic one runs every GUBER_POOL_REBALANCE_MS (250)
    void RebalanceNow();

 private:
    static constexpr uint32_t kStages = 4;                // filling / sealed or on the GPU (up to two) / being read out
    static constexpr uint32_t kMaxEngines = 16;           // engines behind one front stage (guber_stage_submit_routed)
    static constexpr uint64_t kClosed = 1ull << 63;       // Stage::word: not accepting reservations
    static constexpr uint32_t kOpenNone = kStages, kOpenDead = kStages + 1;
    // Stage::word = kClosed | placement version (7 bits) << 56 | key bytes reserved (24 bits) << 32 | slots reserved
    static uint32_t word_count(uint64_t w) { return (uint32_t)w; }
    static uint32_t word_bytes(uint64_t w) { return (uint32_t)(w >> 32) & 0xffffffu; }
    static uint32_t word_ver(uint64_t w) { return (uint32_t)(w >> 56) & 0x7fu; }
    struct Shard;
    struct Stage {                                        // one of a shard's stages and the generation it currently carries
        guber_stage_t* stage = nullptr;
        guber_batch_t* b = nullptr; guber_result_t* r = nullptr;
        Shard* shard = nullptr;
        std::vector<uint16_t> name_len;                   // per slot: length of the request's name (Store callbacks)
        uint32_t* dest = nullptr;                         // a device's front stage: per slot, engine index << 24 | rank in that engine's share
        std::atomic<uint32_t> eng_n[kMaxEngines] = {};    // ... and the shares' sizes so far (callers take their ranks here)
        std::atomic<uint64_t> word{kClosed};
        std::atomic<uint32_t> written{0}, consumed{0};    // slots filled by their callers / responses picked up
        std::atomic<uint64_t> gen{0};                     // generation carried
        std::atomic<uint32_t> done_gen{0};                // low half of the last generation whose responses are ready (futex word)
        std::atomic<uint32_t> sleepers{0};                // callers asleep on done_gen
        std::atomic<int64_t> first_us{0}, last_us{0};     // the generation's first / latest reservation (batch_wait, idle flush)
        std::atomic<bool> flush_now{false};               // a caller found no room: do not wait
        uint32_t n = 0; int rc = 0; int64_t t0_us = 0;    // sealed size, result code, flush start (dispatcher only; rc read after done_gen)
        int64_t t_written_us = 0, t_submitted_us = 0;     // (GUBER_POOL_DEBUG breakdown)
        uint32_t dev_gen = 0;                             // the device generation (one submission) it travelled in
        enum State : uint8_t { kFree, kOpen, kSealed, kInFlight, kDraining } state = kFree;   // dispatcher only
        bool submitted = false;                           // guber_stages_submit succeeded, guber_stage_wait is due
        bool routing = false;                             // guber_stage_route enqueued, the shares' sizes not yet back (then guber_stage_submit_routed)
    };
    struct Device;
    struct Shard {              // one "worker" of the reference: its own cache (engine)
        guber_engine_t* engine = nullptr;
        Stage st[kStages];
        Device* dev = nullptr;
        int32_t device = 0;
        bool global = false;                              // the device's GLOBAL engine
        bool owns_stream = false;                         // its engine created the stream other shards' engines share: destroyed last
        bool front = false;                               // a device's front: stages only (on the first shard's engine), no table of its own
        uint32_t cur = 0;                                 // dispatcher: index of the open stage
        std::atomic<uint32_t> open{kOpenNone};            // index of the stage accepting reservations, kOpenNone, or kOpenDead for good
        std::atomic<uint32_t> open_seq{0}, open_waiters{0};   // futex word of callers waiting for a stage to open, and how many sleep on it
        uint32_t stages_in_flight = 0;                    // dispatcher: stages of this shard on the GPU
        std::atomic<uint64_t> flushed{0}, requests{0}, queue_max{0}, send_us_sum{0}, send_us_max{0}, batch_max{0}, in_flight{0},
            key_too_long{0}, flush_on_key_bytes{0}, direct{0};
        std::atomic<bool> direct_busy{false};             // a caller is evaluating a handful of requests at this shard itself
    };
    struct Device {             // one GPU (or logical device): its shards, their placement, the dispatcher thread
        uint32_t index = 0; int32_t ordinal = 0;
        std::vector<Shard*> shards;                       // plain shards first; the GLOBAL engine, if any, last
        std::unique_ptr<Shard> front;                     // routed pools: the ONE set of stages all callers of the device write into
        std::vector<Shard*> staging;                      // where the callers' reservations go: {front}, or the shards themselves
        uint32_t n_plain = 0;
        guber_placement_t* place = nullptr;
        std::atomic<uint32_t> ver{0};                     // placement version the stages are tagged with
        std::shared_mutex place_mu;                       // cache operations from other threads vs a move in progress
        std::thread thread;
        std::atomic<uint32_t> wake{0};                    // futex word: bumped by callers (first reservation, full stage), Close, RebalanceNow
        std::atomic<bool> sleeping{false}, closing{false}, rebalance_now{false};
        std::atomic<uint64_t> rebalances{0}, moves{0}, submit_us{0}, submits{0};
        uint32_t gen_seq = 0, gen_left[8] = {0}, gens_in_flight = 0;   // dispatcher: submissions whose batches are not all back
        bool rule_dirty = true;                           // dispatcher: the device has not seen the current placement yet (guber_stage_route)
        std::deque<Stage*> routing_q;                     // dispatcher: generations waiting for their shares' sizes (guber_stage_route), oldest first
    };
    struct Ticket2 { Stage* st; uint64_t gen; uint32_t first_slot, count, list_begin, key_base; bool consumed; };
    struct Scratch;                                       // per-thread buffers of a call
    template <class Src, class Sink> struct Call;         // one GetRateLimitMany / GetRateLimitsSoA call (worker_pool.cpp)
    void run(Device& d);
    int create_stages(Shard& sh);
    void open_stage(Shard& sh, uint32_t k, uint32_t ver);
    int find_free(Shard& sh);
    void seal(Shard& sh, Stage& s, std::vector<Stage*>& due);
    void seal_if_due(Shard& sh, int64_t now, bool force, bool eager_ok, std::vector<Stage*>& due, int64_t* deadline);
    void submit_with_store(Shard& sh, Stage& s);
    void route_on_host(Device& d, Stage& s);
    bool submit_routed_now(Device& d, Stage& s, const uint32_t* counts);
    void finish(Stage& s);
    int store_eval(guber_engine_t* engine, const guber_batch_t& B, guber_result_t& R, const uint16_t* name_len);
    void announce(Shard& sh, Stage& s);
    bool poll(std::vector<Stage*>& inflight);
    void drain(std::vector<Stage*>& inflight);
    void submit_due(Device& d, std::vector<Stage*>& due, std::vector<Stage*>& inflight);
    void rebalance(Device& d, std::vector<Stage*>& inflight);
    void wake(Device& d);
    void enter();
    void leave();
    uint32_t route(const Device& d, uint64_t h, uint32_t behavior) const;
    void destroy_engines();

    std::vector<std::unique_ptr<Shard>> shards_;
    std::vector<std::unique_ptr<Device>> devs_;
    guber_ring_t* ring_ = nullptr;
    guber_comm_t* comm_ = nullptr; std::mutex comm_mu_;
    uint32_t n_devices_ = 1, shards_per_device_ = 1, plain_per_device_ = 1;
    bool has_global_ = false;
    bool routed_ = false;                                 // one front stage per device, the GPU hands the requests to the shards
    bool dev_route_ = false;                              // ... and decides which shard a request belongs to (guber_stage_route): callers neither hash nor sort
    uint32_t stage_cap_ = 0;                              // requests a stage takes (routed: up to batch_limit per shard, at most 65536)
    std::vector<Shard*> staging_;                         // every device's staging shards, index = what a caller's routing round counts by
    int create_rc_ = 0;
    uint32_t batch_limit_, batch_wait_us_, idle_us_ = 0, rebalance_ms_ = 250, max_key_ = 1024, key_cap_ = 0;
    uint32_t depth_ = 2, eager_min_ = 4096, spin_us_ = 40, max_active_ = 0x7fffffffu; bool eager_ = true, nt_stores_ = true, one_pass_ = true;
    uint32_t direct_max_ = 4, direct_callers_ = 0;
    std::atomic<uint32_t> in_calls_{0};                   // calls in progress
    std::atomic<uint32_t> spinners_{0}; uint32_t max_spinners_ = 4;   // callers looking at a word instead of sleeping on it
    // callers in the CPU part of a call, bounded by max_active_ and admitted FIRST COME, FIRST SERVED: a caller takes a ticket and sleeps
    // on the turn word of its ticket; whoever leaves grants the next ticket and wakes exactly that sleeper.  (A counting semaphore woke
    // an arbitrary waiter: with 256 callers on 12 slots some waited for many rounds — the p99 of an RPC was 8.8 ms against a p50 of 1 ms.)
    static constexpr uint32_t kTurns = 1024;              // more callers than this in the queue at a time: the late ones spin on a word that is not theirs yet
    bool limit_active_ = false;
    std::atomic<uint64_t> next_ticket_{0}, granted_{0};   // tickets handed out / tickets allowed in (ticket t may enter once t < granted_)
    std::unique_ptr<std::atomic<uint32_t>[]> turn_;       // turn_[t % kTurns] = (uint32_t)(t + 1) once ticket t has been granted (futex word)
    mutable std::atomic<uint64_t> d_dbg_[6] = {};         // GUBER_POOL_DEBUG: where a batch's time goes
    std::atomic<bool> closed_{false};
    std::atomic<int64_t> frozen_ms_{0};
    std::atomic<bool> has_store_{false};
    guber_store_callbacks_t store_{};
};

class V1Instance {
 public:
    explicit V1Instance(GPUWorkerPool* pool) : pool_(pool) {}
    // V1Instance.GetRateLimits (gubernator.go:183-306) for items this instance owns.  Returns false with
    // *rpc_error set for the RPC-level failure (more than 1000 items, codes.OutOfRange).
    bool GetRateLimits(std::vector<RateLimitReq>& reqs, std::vector<RateLimitResp>* resps, std::string* rpc_error);

 private:
    GPUWorkerPool* pool_;
};

}  // namespace gubernator
