// guber_placement_impl.h — the data behind guber_placement_t (placement.cpp) and its wait-free reader, shared with the pool's
// callers (worker_pool.cpp) so that routing a request is a few inlined instructions, not a call.
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/guber_gpu.h"

namespace guber_placement_detail {
constexpr uint32_t kMaxHot = 64, kExCells = 256, kSketchBits = 14, kSketchRows = 2;
constexpr uint32_t kColdRounds = 3;        // online passes in a row a pinned key must be missing from the heavy hitters before its pin goes
constexpr uint64_t kMinWindow = 4096;      // observations a pass must have seen before it lets pins age (an idle period forgets nothing)
constexpr size_t kKeepRetired = 64;        // published snapshots kept alive for readers that loaded the pointer before a publish

struct Exceptions {                     // open addressing on the key hash, immutable once published
    uint32_t n = 0;
    uint64_t h[kExCells] = {0};
    uint16_t s[kExCells] = {0};
    uint8_t cold[kExCells] = {0};       // online passes in a row the key was not heavy (host only: the device sees h[] and s[])
    static uint32_t home(uint64_t x) { return (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 56); }
    void put(uint64_t key, uint16_t shard) {
        uint32_t i = home(key);
        while (h[i] != 0 && h[i] != key) i = (i + 1) & (kExCells - 1);
        if (h[i] == 0) n++;
        h[i] = key; s[i] = shard; cold[i] = 0;
    }
    void set_cold(uint64_t key, uint8_t c) {
        for (uint32_t i = home(key); h[i] != 0; i = (i + 1) & (kExCells - 1)) if (h[i] == key) { cold[i] = c; return; }
    }
    int get(uint64_t key) const {
        if (n == 0) return -1;
        for (uint32_t i = home(key);; i = (i + 1) & (kExCells - 1)) {
            if (h[i] == key) return s[i];
            if (h[i] == 0) return -1;
        }
    }
};
struct Cell { std::atomic<uint64_t> h{0}; std::atomic<uint32_t> c{0}; };
}  // namespace guber_placement_detail

struct guber_placement {
    using Exceptions = guber_placement_detail::Exceptions; using Cell = guber_placement_detail::Cell;
    uint32_t n_shards = 1, n_slots = 1, per = 1;            // n_slots = n_shards x per
    uint64_t step = 0;                                      // 2^63 / n_shards: hashRingStep, workers.go:132
    uint64_t inv_step = 0, inv_sub = 0;                     // floor(2^64 / step), floor(2^64 * per / step)
    std::unique_ptr<std::atomic<uint16_t>[]> table;        // slot -> shard
    std::atomic<const Exceptions*> ex{nullptr};
    std::vector<std::unique_ptr<Exceptions>> retired;       // the last kKeepRetired snapshots published (2.8 KB each): a reader uses the pointer it loaded for a few instructions
    std::unique_ptr<std::atomic<uint64_t>[]> slot_w;        // requests observed per slot
    std::unique_ptr<Cell[]> sketch;                         // [rows][1 << bits]
    std::unique_ptr<Exceptions> pending;                    // guber_placement_plan's list, waiting for guber_placement_commit
    std::atomic<uint32_t> version{0};
    std::mutex mu;                                          // rebalance vs rebalance
    // slot = the reference's worker (workers.go:153-155,180-184: hash63 / (2^63 / n_shards), the last worker taking the
    // remainder) x `per` equal sub-ranges of that worker's range: the initial table (slot -> slot / per) is getWorker EXACTLY
    // (no division on this path: multiply-high by precomputed reciprocals, the worker index corrected to the exact quotient)
    uint32_t slot_of(uint64_t h) const {
        const uint64_t h63 = h >> 1;
        uint64_t w = (uint64_t)(((unsigned __int128)h63 * inv_step) >> 64);
        if ((w + 1) * step <= h63) ++w;                              // inv_step rounds down: the estimate is the quotient or one less
        if (w >= n_shards) w = n_shards - 1;
        uint64_t sub = (uint64_t)(((unsigned __int128)(h63 - w * step) * inv_sub) >> 64);
        if (sub >= per) sub = per - 1;
        return (uint32_t)(w * per + sub);
    }
};


// guber_placement_shard, inlined
static inline uint32_t guber_placement_shard_inl(const guber_placement* p, uint64_t key_hash) {
    if (const guber_placement::Exceptions* e = p->ex.load(std::memory_order_acquire)) {
        const int s = e->get(key_hash);
        if (s >= 0) return (uint32_t)s;
    }
    return p->table[p->slot_of(key_hash)].load(std::memory_order_relaxed);
}
