// worker_pool.h — C++ host layer above the C ABI, mirroring the reference's call surface for the path:
//   gubernator::GPUWorkerPool   <->  WorkerPool            workers.go:54-626
//   gubernator::V1Instance      <->  V1Instance.GetRateLimits (local-owner slice)  gubernator.go:183-306
// Same names, argument meaning and error behaviour; what changes is the mechanism: callers from any
// number of threads are collected by one batcher thread (flush at batch_limit items or batch_wait after
// the first one — the policy of peer_client.go:284-337) and evaluated with one guber_eval_batch.
#pragma once
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
    // `shards` = Config.Workers of the reference (config.go:110, workers.go:125-151): the key space is split by
    // hash range into that many independent caches, here one HBM table + HIP stream + batcher thread each, so that
    // batches of different shards overlap on the GPU (4 saturate an MI355X).  cfg.cache_size is per pool, as in the
    // reference (each shard gets cache_size / shards, workers.go:132).
    GPUWorkerPool(const guber_config_t& cfg, uint32_t batch_limit, uint32_t batch_wait_us, uint32_t shards = 1);
    ~GPUWorkerPool();
    bool ok() const { return !shards_.empty() && create_rc_ == 0; }
    int create_error() const { return create_rc_; }

    // WorkerPool.GetRateLimit (workers.go:261): blocks until the request's batch has been evaluated.
    // Returns false with resp->error set when the reference would return an error.
    bool GetRateLimit(const RateLimitReq& r, RateLimitReqState st, RateLimitResp* resp);
    // Submit many requests, answer all (used by V1Instance::GetRateLimits; order of same-key requests is kept).
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
    // clock.Freeze / clock.Advance of the reference's tests: 0 = wall clock
    // Config.Store (config.go:99, store.go:49-65): call before the first request; nullptr = none
    void SetStore(const guber_store_callbacks_t* cb) { has_store_ = cb != nullptr; if (cb) store_ = *cb; }
    void SetClockMs(int64_t now_ms) { frozen_ms_ = now_ms; }
    int64_t NowMs() const;
    guber_engine_t* engine(uint32_t shard = 0) { return shard < shards_.size() ? shards_[shard]->engine : nullptr; }
    uint32_t shards() const { return (uint32_t)shards_.size(); }
    // WorkerPool.getWorker (workers.go:180-184): shard = (XXH64(key) >> 1) / (2^63 / shards)
    uint32_t ShardOf(const std::string& key) const;
    uint64_t batches_flushed() const;

 private:
    struct Call { std::mutex mu; std::condition_variable cv; size_t remaining = 0; };
    struct Pending { const RateLimitReq* req; RateLimitReqState st; RateLimitResp* resp; Call* call; };
    struct Shard {              // one "worker" of the reference: its own cache (engine), queue and goroutine (thread)
        guber_engine_t* engine = nullptr;
        std::mutex mu;
        std::condition_variable cv;
        std::vector<Pending> queue;
        bool closing = false;
        std::thread thread;
        uint64_t flushed = 0;
    };
    void run(Shard& sh);
    void flush(Shard& sh, std::vector<Pending>& batch);

    std::vector<std::unique_ptr<Shard>> shards_;
    uint64_t ring_step_ = 0;
    int create_rc_ = 0;
    uint32_t batch_limit_, batch_wait_us_;
    bool closed_ = false;
    volatile int64_t frozen_ms_ = 0;
    bool has_store_ = false;
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
