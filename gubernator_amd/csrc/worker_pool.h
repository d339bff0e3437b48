// worker_pool.h — C++ host layer above the C ABI, mirroring the reference's call surface for the path:
//   gubernator::GPUWorkerPool   <->  WorkerPool            workers.go:54-626
//   gubernator::V1Instance      <->  V1Instance.GetRateLimits (local-owner slice)  gubernator.go:183-306
// Same names, argument meaning and error behaviour; what changes is the mechanism.  The reference hands every request to
// its worker's goroutine through a channel (workers.go:261-291); here the CALLERS do the per-request work, in parallel:
// a caller reserves a contiguous range of request slots (and key bytes) in the open stage of its key's shard with one
// compare-and-swap per RPC and shard, writes its requests IN PLACE into the stage's arrays (device-visible host memory,
// include/guber_gpu.h guber_stage_*), and later reads its responses straight out of the stage's result arrays.  The
// shard's batcher thread never touches a request: it seals the open stage at batch_limit items or batch_wait after the
// first one (the policy of peer_client.go:284-337), submits it, opens the next of its three stages (one filling, one on
// the GPU, one being read out by its callers), and announces completed generations.  Nothing is allocated per flush.
// (Optional, off by default: GUBER_POOL_IDLE_US=n also flushes when nobody has reserved anything for n microseconds and all
// reserved slots are written — lower latency under light load; profiles/r02_y_pool_throughput.txt.)
//
// Shards: `devices` x `shards_per_device`.  A key's device is its owner on the reference's replicated consistent hash
// over the peers "gpu0".."gpuN-1" (replicated_hash.go:78-119, 512 vnodes, fnv1) — the N GPUs of a node are N peers —
// and inside a device the shard follows the reference's worker rule (XXH64 range, workers.go:153-155,180-184).  The same
// device ordinal may be listed several times: logical devices on one GPU (single-GPU boxes, tests).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
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
    // in the reference (each shard gets cache_size / shards, workers.go:132).  devices empty = {cfg.device}.
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
    int AddCacheItem(const guber_item_t& item);                                  // workers.go:537
    int GetCacheItem(const std::string& key, guber_item_t* out, bool* found);    // workers.go:583
    // WorkerPool.Load (workers.go:329-449): hand every item of a Loader to the cache; WorkerPool.Store (workers.go:451-534):
    // visit every resident item (what Loader.Save receives).  Bulk paths: guber_add_items / guber_dump.
    int Load(const guber_item_t* items, uint32_t n);
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
    // (device * shards_per_device + WorkerPool.getWorker, workers.go:180-184)
    uint32_t DeviceOf(const uint8_t* key, uint32_t len) const;
    uint32_t ShardOf(const std::string& key) const { return ShardOf((const uint8_t*)key.data(), (uint32_t)key.size()); }
    uint32_t ShardOf(const uint8_t* key, uint32_t len) const;
    uint64_t batches_flushed() const;
    void Metrics(guber_pool_metrics_t* out) const;   // gubernator_batch_queue_length / gubernator_batch_send_duration analogues (gubernator.go:96-107)

 private:
    static constexpr uint32_t kStages = 3;                // filling / on the GPU / being read out
    static constexpr uint64_t kClosed = 1ull << 63;       // Stage::word: not accepting reservations
    struct Stage {                                        // one of a shard's stages and the generation it currently carries
        guber_stage_t* stage = nullptr;
        guber_batch_t* b = nullptr; guber_result_t* r = nullptr;
        std::vector<uint16_t> name_len;                   // per slot: length of the request's name (Store callbacks)
        std::atomic<uint64_t> word{kClosed};              // kClosed | key bytes reserved << 32 | slots reserved
        std::atomic<uint32_t> written{0}, consumed{0};    // slots filled by their callers / responses picked up
        std::atomic<uint64_t> gen{0};                     // generation carried
        std::atomic<uint32_t> done_gen{0};                // low half of the last generation whose responses are ready (futex word)
        std::atomic<int64_t> first_us{0};                 // when the generation's first reservation was made (batch_wait)
        std::atomic<bool> flush_now{false};               // a caller found no room: do not wait for batch_wait
        uint32_t n = 0; int rc = 0; int64_t t0_us = 0;    // sealed size, result code, flush start (batcher only; rc read after done_gen)
        bool submitted = false;                           // guber_stage_submit succeeded, guber_stage_wait is due
    };
    struct Shard {              // one "worker" of the reference: its own cache (engine) and goroutine (batcher thread)
        guber_engine_t* engine = nullptr;
        Stage st[kStages];
        int32_t device = 0;
        std::atomic<uint32_t> open{kStages};              // index of the stage accepting reservations; kStages = none right now, kStages + 1 = closed for good
        std::mutex mu;
        std::condition_variable cv_batcher, cv_callers;   // batcher: first item / full / closing; callers in reserve(): a stage opened
        bool closing = false;
        std::thread thread;
        std::atomic<uint64_t> flushed{0}, requests{0}, queue_max{0}, send_us_sum{0}, send_us_max{0}, batch_max{0}, in_flight{0},
            key_too_long{0}, flush_on_key_bytes{0};
    };
    struct Ticket { Shard* sh; Stage* st; uint64_t gen; uint32_t first_slot, count, list_begin, key_base; const std::vector<uint32_t>* list; bool consumed; };
    struct Job;                                           // one GetRateLimitMany call: its per-shard request lists and tickets
    void run(Shard& sh);
    void open_stage(Shard& sh, uint32_t k);
    void submit(Shard& sh, Stage& s);
    void submit_with_store(Shard& sh, Stage& s);
    void complete(Shard& sh, Stage& s);
    uint32_t reserve(Job& job, Shard& sh, const std::vector<uint32_t>& list, uint32_t begin, Ticket* out);
    void write_requests(Job& job, const Ticket& t, const std::vector<uint32_t>& list);
    bool try_consume(Job& job, Ticket& t, bool block);
    void fail_rest(Job& job, const std::vector<uint32_t>& list, uint32_t begin, const char* why);

    std::vector<std::unique_ptr<Shard>> shards_;
    guber_ring_t* ring_ = nullptr;
    uint32_t n_devices_ = 1, shards_per_device_ = 1;
    uint64_t ring_step_ = 0;
    int create_rc_ = 0;
    uint32_t batch_limit_, batch_wait_us_, idle_us_ = 0, max_key_ = 1024, key_cap_ = 0;
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
