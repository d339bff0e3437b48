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
    GPUWorkerPool(const guber_config_t& cfg, uint32_t batch_limit, uint32_t batch_wait_us);
    ~GPUWorkerPool();
    bool ok() const { return engine_ != nullptr; }
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
    guber_engine_t* engine() { return engine_; }
    uint64_t batches_flushed() const { return flushed_; }

 private:
    struct Call { std::mutex mu; std::condition_variable cv; size_t remaining = 0; };
    struct Pending { const RateLimitReq* req; RateLimitReqState st; RateLimitResp* resp; Call* call; };
    void run();
    void flush(std::vector<Pending>& batch);

    guber_engine_t* engine_ = nullptr;
    int create_rc_ = 0;
    uint32_t batch_limit_, batch_wait_us_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Pending> queue_;
    bool closing_ = false;
    std::thread thread_;
    volatile int64_t frozen_ms_ = 0;
    uint64_t flushed_ = 0;
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
