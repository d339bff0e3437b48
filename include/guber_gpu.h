/*
 * guber_gpu.h — C ABI of the MI355X rate-limit evaluation engine.
 *
 * This is the drop-in boundary for ONE hot path of mailgun/gubernator: everything
 * below `s.workerPool.GetRateLimit(ctx, r, reqState)` (reference gubernator.go:598),
 * i.e. workers.go (WorkerPool) + algorithms.go (tokenBucket / leakyBucket) +
 * lrucache.go / cache.go (Cache, CacheItem).  A Go `GPUWorkerPool` binds these
 * entry points through cgo (see INTEGRATION.md and go/gpu_worker_pool.go); each
 * entry point cites the reference interface it replaces.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; every pointer is caller-owned
 *    unless the function name says `alloc`.
 *  - return value: 0 (GUBER_OK) or a negative GUBER_E_* code; never throws.
 *  - no caller memory is retained after a call returns (cgo pointer rule).
 *  - `*_dev` variants take DEVICE pointers (HBM-resident SoA) and enqueue on the
 *    engine stream without synchronising; the plain variants take HOST pointers,
 *    stage through pinned buffers and return after the results are on the host.
 *  - responses are positionally aligned with requests (gubernator.proto:51-54).
 */
#ifndef GUBER_GPU_H
#define GUBER_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (values = gubernator.proto:56-135) ------------------------------ */
#define GUBER_ALGO_TOKEN_BUCKET 0u
#define GUBER_ALGO_LEAKY_BUCKET 1u

#define GUBER_STATUS_UNDER_LIMIT 0u
#define GUBER_STATUS_OVER_LIMIT 1u

#define GUBER_BEHAVIOR_NO_BATCHING 1u
#define GUBER_BEHAVIOR_GLOBAL 2u
#define GUBER_BEHAVIOR_DURATION_IS_GREGORIAN 4u
#define GUBER_BEHAVIOR_RESET_REMAINING 8u
#define GUBER_BEHAVIOR_MULTI_REGION 16u
#define GUBER_BEHAVIOR_DRAIN_OVER_LIMIT 32u

/* Gregorian interval selectors (interval.go:74-81) carried in `duration`. */
#define GUBER_GREGORIAN_MINUTES 0
#define GUBER_GREGORIAN_HOURS 1
#define GUBER_GREGORIAN_DAYS 2
#define GUBER_GREGORIAN_WEEKS 3
#define GUBER_GREGORIAN_MONTHS 4
#define GUBER_GREGORIAN_YEARS 5

/* ---- call-level return codes ------------------------------------------------ */
#define GUBER_OK 0
#define GUBER_E_INVALID_ARG (-1)
#define GUBER_E_NO_DEVICE (-2)     /* HIP runtime / GPU missing: the product path never falls back to CPU */
#define GUBER_E_HIP (-3)           /* a HIP call failed; guber_last_error() has the text */
#define GUBER_E_BATCH_TOO_LARGE (-4)
#define GUBER_E_TABLE_FULL (-5)    /* guber_add_items: no directory entry for an item although nothing is evictable; guber_move_items_by_hash: the destination took not every bucket (those went back to the source: nothing is lost) */
#define GUBER_E_NOMEM (-6)
#define GUBER_E_KEY_TOO_LONG (-7)
#define GUBER_E_NOT_FOUND (-8)

/* ---- per-item error codes (guber_result_t.err) ------------------------------
 * The reference surfaces these as RateLimitResp.Error strings; guber_item_strerror()
 * returns the exact reference text. */
#define GUBER_ITEM_OK 0u
#define GUBER_ITEM_E_INVALID_ALGORITHM 1u /* workers.go:318  "Invalid rate limit algorithm '%d'" */
#define GUBER_ITEM_E_GREGORIAN_WEEKS 2u   /* interval.go:93,136 */
#define GUBER_ITEM_E_GREGORIAN_INVALID 3u /* interval.go:107,147 */
#define GUBER_ITEM_E_EMPTY_KEY 4u         /* gubernator.go:208-217 (host validation normally catches it) */
#define GUBER_ITEM_E_RETRY 5u             /* internal: two new keys with one 64-bit hash in one batch; the
                                             host layer re-submits these items, callers never see it */
#define GUBER_ITEM_E_TABLE_FULL 6u        /* no free directory entry / key arena exhausted */
#define GUBER_ITEM_E_KEY_TOO_LONG 7u      /* key longer than guber_config_t.max_key_bytes */

typedef struct guber_engine guber_engine_t;

/* guber_config_t.flags */
#define GUBER_FLAG_GLOBAL 8u          /* keep per-bucket pending GLOBAL hits / updates (guber_global_take) */
#define GUBER_FLAG_NO_PART 128u        /* never take the owner-partitioned pipeline: batches of 257 .. 65 536 requests run the
                                          two-launch pipeline with per-batch claims (round 3's; kept as the retry round) */
/* (bits 1, 2, 4, 32 and 64 select code paths for the test suite — gubernator_amd/csrc/guber_test_flags.h —, bit 16 is accepted and
 *  ignored: a binding passes none of them) */

/* Engine configuration.  Replaces Config.{CacheSize,Workers,CacheFactory}
 * (reference config.go:73-123, workers.go:125-147). */
typedef struct guber_config {
    uint32_t struct_size;    /* sizeof(guber_config_t), for ABI growth */
    int32_t device;          /* HIP device ordinal */
    uint64_t cache_size;     /* max resident rate limits (Config.CacheSize, default 50_000) */
    uint64_t table_slots;    /* 0 = derive: next pow2 >= 4 x (cache_size + max_batch) while the table stays below 4 GB (load <= 0.25: nine keys in ten at
                                their home position), else >= 2 x (load <= 0.5) */
    uint32_t max_batch;      /* largest n accepted by one eval call (0 = 65536) */
    uint32_t max_key_bytes;  /* longest single key accepted (0 = 1024) */
    void* stream;            /* optional caller-owned hipStream_t; NULL = engine creates its own */
    uint32_t flags;          /* GUBER_FLAG_* */
    uint32_t reserved;
} guber_config_t;

/* One batch of rate-limit checks, structure-of-arrays, all arrays length n.
 * Field meaning = RateLimitReq (gubernator.proto:137-182).  The key is the
 * reference's HashKey(): name + "_" + unique_key (client.go:39-41). */
typedef struct guber_batch {
    uint32_t n;
    uint32_t reserved;
    const uint8_t* key_bytes;   /* concatenated keys; the buffer must stay readable for 8 bytes past the
                                   last key (the kernels load keys as 8-byte words) */
    const uint32_t* key_off;    /* n+1 offsets into key_bytes */
    const int64_t* hits;
    const int64_t* limit;
    const int64_t* duration;
    const int64_t* burst;       /* NULL = all 0 (leaky then defaults burst to limit, algorithms.go:264) */
    const int64_t* created_at;  /* NULL = all now_ms (gubernator.go:218-220) */
    const uint8_t* algorithm;   /* 0 token, 1 leaky; anything else -> GUBER_ITEM_E_INVALID_ALGORITHM */
    const uint32_t* behavior;   /* Behavior bit set */
    const uint8_t* is_owner;    /* RateLimitReqState.IsOwner; NULL = all 1 */
    /* Host-precomputed calendar values for DURATION_IS_GREGORIAN items (interval.go:84-148),
     * NULL when no item carries the bit.  greg_duration < 0 encodes the reference's error:
     * -GUBER_ITEM_E_GREGORIAN_WEEKS or -GUBER_ITEM_E_GREGORIAN_INVALID.
     * With both NULL the kernels compute the interval from now_ms themselves, IN UTC.  The reference uses now.Location()
     * (interval.go:97-142), the daemon's local zone: a daemon that does not run in UTC passes the two arrays. */
    const int64_t* greg_expire;
    const int64_t* greg_duration;
    int64_t now_ms;             /* MillisecondNow() at batch evaluation (lrucache.go:106, cache.go:44) */
} guber_batch_t;

/* Results, SoA, arrays length n; field meaning = RateLimitResp (gubernator.proto:189-203). */
typedef struct guber_result {
    uint8_t* status;
    int64_t* limit;
    int64_t* remaining;
    int64_t* reset_time;
    uint8_t* err;               /* GUBER_ITEM_* */
    /* per-batch aggregates the Go shim adds to the reference's prometheus vars
     * (algorithms.go:165,185,243,391,409,471; lrucache.go:117,121,126,144). Host variants fill
     * them on return; device variants leave them 0 — read guber_stats() after synchronising. */
    uint64_t over_limit_count;
    uint64_t cache_hits;
    uint64_t cache_misses;
    uint64_t unexpired_evictions;
    int64_t cache_size;
} guber_result_t;

/* A materialised CacheItem (cache.go:29-41) with its TokenBucketItem / LeakyBucketItem
 * (store.go:29-43) flattened.  Used by AddCacheItem / GetCacheItem / Load / Store. */
typedef struct guber_item {
    uint8_t algorithm;   /* CacheItem.Algorithm */
    uint8_t status;      /* TokenBucketItem.Status (token only) */
    uint16_t reserved0;
    uint32_t key_len;
    const uint8_t* key;  /* CacheItem.Key bytes; for outputs points into the caller's key arena */
    int64_t limit;
    int64_t duration;
    int64_t remaining;        /* token: TokenBucketItem.Remaining */
    double remaining_f;       /* leaky: LeakyBucketItem.Remaining */
    int64_t stamp;            /* token: CreatedAt; leaky: UpdatedAt */
    int64_t burst;            /* leaky only */
    int64_t expire_at;        /* CacheItem.ExpireAt */
    int64_t invalid_at;       /* CacheItem.InvalidAt */
} guber_item_t;

typedef struct guber_stats {
    uint64_t over_limit_count;
    uint64_t cache_hits;
    uint64_t cache_misses;
    uint64_t unexpired_evictions;
    int64_t cache_size;       /* LRUCache.Size() (lrucache.go:159) */
    uint64_t table_slots;
    uint64_t tags_used;       /* claimed directory entries (live + tombstoned keys) */
    uint64_t batches;
    uint64_t retries;         /* GUBER_ITEM_E_RETRY re-submissions */
    uint64_t compactions;     /* table rebuilds (guber_compact or automatic) */
    uint64_t small_batches;   /* batches of <= 256 requests answered by the one-launch path */
    uint64_t fused_batches;   /* batches that shared their two launches with other engines' batches (guber_eval_batches_routed_dev) */
    uint64_t eviction_passes; /* eviction pre-passes that evicted (a call that would have overflowed the cache: "Bounded cache" below) */
    uint64_t tail_rebuilds;   /* times the recency order of the live items was rebuilt from the table (one scan + one sort) */
    uint64_t batch_cuts;      /* batches evaluated in pieces: larger than the cache (pieces of cache_size requests), or holding a request that changes the list's length ("bounded cache" below) */
} guber_stats_t;

/* ---- lifecycle: NewWorkerPool / WorkerPool.Close (workers.go:125,157) -------- */
int guber_engine_create(const guber_config_t* cfg, guber_engine_t** out);
void guber_engine_destroy(guber_engine_t* e);

/* ---- the hot path: WorkerPool.GetRateLimit for a whole batch (workers.go:261-324,
 *      algorithms.go:37-493).  Host pointers in, host results out. */
int guber_eval_batch(guber_engine_t* e, const guber_batch_t* batch, guber_result_t* result);

/* Same, every pointer inside batch/result is a DEVICE pointer (the structs themselves live on
 * the host).  Asynchronous on the engine stream. */
int guber_eval_batch_dev(guber_engine_t* e, const guber_batch_t* batch, guber_result_t* result);

/* ---- stages: the overlapped end-to-end path.  A stage is one batch's worth of request / response arrays in device-visible
 *      host memory.  The caller (a batcher goroutine) writes requests straight into guber_stage_batch()'s arrays as they
 *      arrive — no Go pointers are retained, nothing is copied or allocated per batch on the host — sets n and now_ms, and
 *      submits.  How the arrays reach the kernels depends on the batch: <= 256 requests: one launch reads and writes them in
 *      place over PCIe; a batch that fills at least half of the stage: DMA copies on a copy stream bring the request
 *      columns into the stage's HBM mirror while the previous batches' kernels run, the two-launch pipeline works on HBM
 *      and writes the responses straight into the host arrays; in between: the pipeline's kernels read / write the host
 *      arrays in place.  guber_stage_submit returns at
 *      once; guber_stage_wait blocks until the responses are there (a polled sequence number for batches <= 256, a HIP
 *      event otherwise), resolves internal retries and fills the per-batch aggregates.  With several stages per engine one
 *      is filled while the GPU evaluates another (the pool keeps four per device: filling / up to two on the GPU / being read out).  Stages of one engine are evaluated in submission order; the one
 *      exception are items that hit the internal retry (two keys under one 64-bit hash inside a batch, ~1e-6 per batch): they
 *      are re-run by the guber_stage_wait of their stage, i.e. after whatever is already in flight.  Later requests of such a key
 *      meet the same resident key, are retried too and re-run by THEIR stage's wait: per key the order is the order of the
 *      guber_stage_wait calls — waits in submission order keep the submission order also with several stages in flight; a caller
 *      that waits in completion order (the pool) may apply a later stage's requests of such a key first: the order two concurrent
 *      GetRateLimits calls have in the reference too (none).  (tests/test_gpu_parity.py
 *      test_internal_retries_follow_the_order_of_the_waits pins both.)
 *      Optional request arrays may be switched off by setting the pointer in guber_stage_batch() to NULL (burst, created_at,
 *      is_owner, behavior, algorithm: the guber_batch_t defaults apply); key_bytes_cap = 0 -> 64 bytes per request. */
typedef struct guber_stage guber_stage_t;
int guber_stage_create(guber_engine_t* e, uint32_t max_n, uint32_t key_bytes_cap, guber_stage_t** out);
void guber_stage_destroy(guber_stage_t* s);
guber_batch_t* guber_stage_batch(guber_stage_t* s);
guber_result_t* guber_stage_result(guber_stage_t* s);
uint32_t guber_stage_capacity(guber_stage_t* s, uint32_t* key_bytes_cap);
int guber_stage_submit(guber_stage_t* s);
int guber_stage_wait(guber_stage_t* s);
/* Several stages in one submission — what ONE dispatcher serving all logical shards of a GPU calls (GPUWorkerPool): at most one
 * stage per engine; batches of > 256 requests of engines that share device and stream travel as fused launches (one copy
 * kernel bringing the request columns of the group's stages to HBM — up to 16 — then k_front_multi / k_eval2_multi, whose
 * argument blocks travel by value for up to four stages and through device memory beyond; one completion event per group), batches of <= 256
 * requests take the one-launch path WITHOUT waiting for it.  Never blocks on the GPU.  With GUBER_STAGES_NO_AGGREGATES the
 * per-batch aggregates of guber_result_t are not produced (no counter read-back launches; guber_stats has the totals);
 * without it the stages are submitted one by one exactly as guber_stage_submit does.  *done (optional) = stages enqueued.
 * guber_stage_poll: 1 = the stage's responses are in its result arrays (guber_stage_wait then returns at once, after resolving
 * the rare internal retry), 0 = still running, < 0 = error.  Polling is how the dispatcher learns of completions without
 * parking a thread per batch. */
#define GUBER_STAGES_NO_AGGREGATES 1u
int guber_stages_submit(guber_stage_t* const* stages, uint32_t n, uint32_t flags, uint32_t* done);
int guber_stage_poll(guber_stage_t* s);
/* ONE stage for the requests of several engines — the logical shards of a GPU behind one reservation word (GPUWorkerPool:
 * WorkerPool.getWorker / dispatch, workers.go:180-258, with the hand-over to the worker done by the GPU).  The callers write
 * their requests in arrival order and, per request, into guber_stage_dest()[i] the index of its engine in `engines` << 24 |
 * its rank inside that engine's share (the ranks of an engine are 0 .. counts[engine]-1, each exactly once; requests of one
 * key carry ranks in the order they are to be applied).  The copy kernel that brings the request columns to HBM places every
 * share contiguously in rank order, the shares run as the batches of ONE k_front_multi_mem + ONE k_eval2_multi_mem, and the
 * answers land in the result arrays at the index the request was written at.  Four launches and one event whatever the number
 * of engines (<= 16; they share device and stream, the stage's own engine is one of them).  The call does not wait for ITS work;
 * it can wait — bounded: a polled sequence number, a stream synchronise after 2 s — for an EARLIER one-launch stage of one of
 * the engines whose outcome has not been looked at yet (its declined shares must be re-run before later requests of their keys),
 * and for a completion-event slot of sixteen to be free.  A stage of <= 256 requests goes as ONE launch (k_small_routed); ranks that
 * are not a permutation of 0 .. counts[engine]-1 make that share report fallback (it is re-run by index list), never an access
 * outside the stage.  Completion through guber_stage_poll / guber_stage_wait (no per-batch aggregates).  Every request column of the stage is
 * present (none switched off). */
uint32_t* guber_stage_dest(guber_stage_t* s);
int guber_stage_submit_routed(guber_stage_t* s, guber_engine_t* const* engines, uint32_t n_engines, const uint32_t* counts);
/* The routing itself done by the device: WorkerPool.getWorker (workers.go:153-155 ComputeHash63, :180-184) for every request of a
 * front stage at once, so that a caller's per-request work is writing the request and nothing else (no hash, no table lookup,
 * no sort by shard, no rank).  The callers fill the stage in arrival order and leave guber_stage_dest alone;
 *   guber_stage_route       enqueues two launches on the stage's engine: XXH64 of every HashKey + the rule -> engine, the shares'
 *                           sizes, and dest[i] = engine << 24 | rank in arrival order.  `rule` = the placement as exported by
 *                           guber_placement_export (copied to the device: waits for the stream, so hand it over only when the
 *                           placement changed; NULL = the rule given last); global_engine >= 0: requests with
 *                           GUBER_BEHAVIOR_GLOBAL go to that engine index.  At most 65 536 requests.  The stage's keys are brought to
 *                           its HBM mirror on the way and stay there for guber_stage_submit_routed: the stage is not modified in between.
 *   guber_stage_route_poll  1 = counts[0 .. n_engines) hold the shares' sizes, 0 = still running.  guber_stage_submit_routed may
 *                           follow at once (stream order makes dest complete before it is read). */
typedef struct guber_route_rule {
    uint32_t n_shards, per;                       /* hash slots = n_shards x per; slot -> shard through table[] */
    uint64_t step, inv_step, inv_sub;             /* 2^63 / n_shards (workers.go:132 hashRingStep) and the reciprocals slot_of multiplies by */
    const uint16_t* table;
    uint32_t ex_cells, ex_n;                      /* individually placed keys: open addressing on the key hash (0 = empty cell) */
    const uint64_t* ex_hash; const uint16_t* ex_shard;
    int32_t global_engine;                        /* -1 = none */
} guber_route_rule_t;
int guber_stage_route(guber_stage_t* s, const guber_route_rule_t* rule, uint32_t n_engines);
int guber_stage_route_poll(guber_stage_t* s, uint32_t* counts);

/* A queue of device-resident batches enqueued back to back on the engine stream in one call (what a batcher goroutine
 * that has several full batches waiting does, peer_client.go:284-337): batches[i] -> results[i], i = 0..count-1, in
 * order.  Asynchronous like guber_eval_batch_dev; stops at the first batch that fails to enqueue and returns its code
 * (*done, optional, receives the number of batches enqueued). */
int guber_eval_batches_dev(guber_engine_t* e, const guber_batch_t* batches, guber_result_t* results, uint32_t count,
                           uint32_t* done);

/* The same for several engines from ONE dispatcher (the logical shards of one GPU, workers.go:125-151: one cache per
 * worker): batch k belongs to engines[which[k]].  Per engine the array order is kept; batches of different engines share no
 * state (disjoint keys) and are enqueued round by round — the next batch of every engine that has one — so that up to four
 * engines' batches travel in ONE pair of launches when the engines were created on the same device and stream
 * (guber_config_t.stream) and the batches take the two-launch pipeline (n <= 65 536).  Results are identical to enqueueing
 * every batch on its own; what changes is the rate (the launches of one batch leave most of the GPU idle).
 * On failure *done counts the batches enqueued (each engine's enqueued batches are a prefix of its own). */
int guber_eval_batches_routed_dev(guber_engine_t* const* engines, uint32_t n_engines, const uint32_t* which,
                                  const guber_batch_t* batches, guber_result_t* results, uint32_t count, uint32_t* done);

/* ---- the front of a GPU's logical shards, on the device: WorkerPool.GetRateLimit's choice of the worker (workers.go:261-289,
 *      getWorker :180-184: XXH64 of the HashKey -> worker) and GetRateLimits' answers in request order (gubernator.go:203-300,
 *      gubernator.proto:51-54), for requests that already lie in HBM.  guber_eval_batches_routed_dev takes batches somebody has
 *      split by shard and leaves the answers in the shards' order; a front takes ONE stream of requests in ARRIVAL order:
 *        gens[k]     a generation: up to max_n requests (what a batcher has collected; all columns DEVICE pointers, no host-computed
 *                    calendar columns), one now_ms
 *        on the GPU  XXH64 of every key + the placement's rule (guber_placement_export; NULL with one engine) -> engine; every engine's
 *                    share contiguous and in arrival order (a key's requests keep their order); the shares run through the engines'
 *                    pipelines, engines that share a stream sharing launches, generation after generation; the answers return to
 *                    results[k]'s arrays in arrival order.  The routing runs two generations ahead on a stream of its own, the host
 *                    reads the shares' sizes from pinned memory and never waits for the GPU in steady state.
 *      Results are those of evaluating the generation's requests one by one in arrival order (what gubernator.go:203 does).  depth =
 *      generations in flight (0 = 4).  Asynchronous like guber_eval_batch_dev; guber_front_synchronize waits for the engines' streams
 *      and the routing stream.  On failure *done counts the generations whose answers are on their way. */
typedef struct guber_front guber_front_t;
typedef struct guber_front_stats {
    uint64_t generations;      /* evaluated */
    uint64_t forced_flushes;   /* times a slot was needed before its generation's held-back evaluation had been launched */
    uint64_t host_waits;       /* times the host found a generation's share sizes not yet reported ... */
    uint64_t host_wait_us;     /* ... and how long it waited in total */
} guber_front_stats_t;
int guber_front_create(guber_engine_t* const* engines, uint32_t n_engines, const struct guber_route_rule* rule, uint32_t max_n,
                       uint32_t depth, guber_front_t** out);
void guber_front_destroy(guber_front_t* f);
int guber_front_set_rule(guber_front_t* f, const struct guber_route_rule* rule);   /* after a placement commit (waits for the routing stream) */
int guber_front_eval_dev(guber_front_t* f, const guber_batch_t* gens, guber_result_t* results, uint32_t count, uint32_t* done);
int guber_front_synchronize(guber_front_t* f);
void* guber_front_stream(guber_front_t* f);                                        /* the routing stream (hipStream_t): the answers' last hop runs on it */
int guber_front_stats(guber_front_t* f, guber_front_stats_t* out);
/* With guber_profile_enable on the front's FIRST engine while generations ran: per generation the time from its first routing kernel's
 * start to the end of its answers' last hop, in microseconds, since the last call (us may be NULL, *n_out = how many there were).
 * The routing kernels appear in that engine's guber_profile_read as k_fr_count / k_fr_scatter / k_fr_out.  Waits for the routing stream. */
int guber_front_latencies(guber_front_t* f, float* us, uint32_t cap, uint32_t* n_out);

/* ---- WorkerPool.AddCacheItem (workers.go:537; callers gubernator.go:452 UpdatePeerGlobals,
 *      workers.go:329 Load).  Add semantics = LRUCache.Add (lrucache.go:88): replace if present.
 *      existed[i] (optional) receives Add's return value. */
int guber_add_items(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed);

/* ---- WorkerPool.GetCacheItem (workers.go:583) = LRUCache.GetItem (lrucache.go:111): an expired
 *      item is removed and reported absent.  *found = 0/1.  out->key is not filled. */
int guber_get_item(guber_engine_t* e, const uint8_t* key, uint32_t key_len, int64_t now_ms,
                   guber_item_t* out, int* found);

/* ---- LRUCache.Remove (lrucache.go:131) */
int guber_remove_item(guber_engine_t* e, const uint8_t* key, uint32_t key_len);

/* ---- LRUCache.Size (lrucache.go:159) */
int64_t guber_size(guber_engine_t* e);

/* ---- WorkerPool.Store / LRUCache.Each (workers.go:451, lrucache.go:76): dump every resident
 *      item.  items[0..cap) and key_arena[0..arena_cap) are caller buffers; *n_out gets the item
 *      count.  Returns GUBER_E_NOMEM if a buffer is too small (then *n_out / *arena_out hold the
 *      needed sizes). */
int guber_dump(guber_engine_t* e, guber_item_t* items, uint64_t cap, uint8_t* key_arena,
               uint64_t arena_cap, uint64_t* n_out, uint64_t* arena_out);

/* ---- Config.Store (store.go:49-65), the persistent write-through store.  The reference calls, from inside
 *      the algorithms: Store.Get on a cache miss (algorithms.go:45-51, :274-280), Store.OnChange(r, item) after
 *      a request was applied when the node owns the key — `item` = the CacheItem as it is AFTER that request
 *      (deferred at :149-153 / :382-386, direct at :252-254 / :488-490) — and Store.Remove(key) when an item is
 *      dropped for RESET_REMAINING (token, :79-84) or because the algorithm changed (:96-100, :311-315).
 *      These are user Go code, so they stay on the host; the engine tells the host WHICH calls are due:
 *        1. guber_probe_missing  -> missing[i] = 1: the key of request i is not resident (absent or expired
 *           at now_ms).  The host asks Store.Get for the first request of each such key and hands what it
 *           finds to guber_add_items (= `c.Add(item)`, algorithms.go:49).
 *        2. guber_eval_batch_store = guber_eval_batch + per request i: flags[i] (GUBER_STORE_*) and, when
 *           GUBER_STORE_ONCHANGE is set, items[i] = the item right after request i (exact also for requests
 *           in the middle of a run on one key; items[i].key points into the batch's key_bytes).
 *           The host then issues, in request order: Remove (if flagged), then OnChange (if flagged).
 *      Divergence: an item the Store returns that is already expired at now_ms is treated as absent (the
 *      reference would use it without re-checking IsExpired). */
#define GUBER_STORE_ONCHANGE 1
#define GUBER_STORE_REMOVE 2
typedef struct guber_store_events {
    uint8_t* flags;          /* n */
    guber_item_t* items;     /* n */
} guber_store_events_t;
int guber_probe_missing(guber_engine_t* e, const guber_batch_t* b, uint8_t* missing);
int guber_eval_batch_store(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r, guber_store_events_t* ev);

/* ---- bounded cache: the reference keeps at most CacheSize items in a list ordered by last access (Add and GetItem move an
 *      item to the front, lrucache.go:88-128) and removes the item at the back the moment an insert makes the list longer than
 *      that (:98-100, :138-149) — in the middle of a stream of requests: a key evicted by request i is a new item for request
 *      j > i — counting the evictions of items that had not expired yet (gubernator_unexpired_evictions_count, :142-146).
 *      The engine keeps exactly that list: every bucket carries the sequence number of the last request that touched it (53
 *      bits; request i of a batch after request i - 1, the items of an Add in their order), and a call that may overflow the
 *      cache (live items + requests > cache_size) first goes through an eviction pre-pass that decides, from the order of the
 *      batch's first accesses and the recency order of the resident items, which resident keys are gone before the batch first
 *      asks for them (their buckets become absent: the batch sees a new key, as the reference does) and which untouched items
 *      leave; a batch larger than cache_size is evaluated in pieces of cache_size requests (guber_stats_t.batch_cuts).
 *      Expired items keep their place until somebody asks for them or they reach the back, as in the reference; items with
 *      pending GLOBAL work are evicted like any other (the pending record stays queued, as the reference's queues do).
 *      Every answer, LRUCache.Size() and the unexpired evictions equal the reference's on the same request sequence
 *      (tests/test_gpu_parity.py test_evicted_keys_that_return_meet_the_reference_list,
 *      test_a_batch_larger_than_the_cache_is_evaluated_in_pieces; tests/test_kernels_devsim.py on the CPU).  A request with an
 *      invalid algorithm never reaches the cache (workers.go:317-321): it counts no access and leaves its key's place in the list
 *      alone (test_an_invalid_algorithm_request_does_not_refresh_recency); duplicates of one key inside one guber_add_items
 *      call leave the keys in the order of their LAST places in the call (test_duplicates_inside_one_add_keep_the_calls_order)
 *      — both closed in round 5 —; and a request that fails inside the algorithm BEFORE c.Add — DURATION_IS_GREGORIAN with a
 *      duration that is no interval constant, or GregorianWeeks (interval.go:93,97,107,125,130,148) — inserts nothing: a key that
 *      is not resident and whose requests in the batch all fail that way never enters the list, a key with a failing request
 *      first and a good one later is inserted by the good one (test_a_new_key_whose_requests_all_fail_in_the_algorithm_is_no_insert;
 *      closed at the end of round 5).  Two kinds of request change the LENGTH of the reference's list instead of moving a key to its
 *      front — a TOKEN_BUCKET request with RESET_REMAINING for a key that is in the cache (algorithms.go:78-90: the item is removed,
 *      nothing is inserted) and a resident key's FIRST request that fails before c.Add (above) when the key had been pushed out before
 *      that request or had expired (lrucache.go:111-128: GetItem misses / removes, nothing takes the place) —; while the cache binds the
 *      pre-pass reports the first such request of a batch and the engine evaluates the requests before it, that request alone, and the
 *      rest as batches of their own (guber_stats_t.batch_cuts counts them): exact by construction
 *      (test_requests_that_change_the_lists_length_are_evaluated_on_their_own; round 5 had documented the second kind as not
 *      reproduced).  Nothing of lrucache.go:88-171 is left unreproduced.
 *      Cost: nothing while live items + requests <= cache_size (size the cache with a batch of headroom); beyond that one
 *      pre-pass per batch (a dozen small launches and one stream synchronisation; the recency order of the live items comes
 *      from one table scan + sort per ~live/(2 x batch) batches: guber_stats_t.tail_rebuilds).
 *      The directory entries of evicted / removed keys are reclaimed by a rebuild of the table (guber_compact,
 *      also automatic when the directory passes 7/8 full; GLOBAL engines included, pending records move with their
 *      buckets; the long-key arena is rebuilt too).  Only when the items themselves cannot be placed do new keys get
 *      GUBER_ITEM_E_TABLE_FULL — per item; resident keys are always served.
 *      The engine has no clock: "now" arrives with every batch; maintenance between batches (eviction right after
 *      guber_add_items) classifies items as expired against the latest value seen, which guber_set_clock overrides
 *      (the reference's clock.Freeze / clock.Advance). */
int guber_compact(guber_engine_t* e, int64_t now_ms);
int guber_set_clock(guber_engine_t* e, int64_t now_ms);

int guber_stats(guber_engine_t* e, guber_stats_t* out);
int guber_synchronize(guber_engine_t* e);

/* ---- per-kernel timing for the roofline report (replaces the reference's prometheus
 *      metricFuncTimeDuration timers, algorithms.go:38, workers.go:294).  While enabled every launch of
 *      the batch sequence is bracketed by HIP events on the engine stream; guber_profile_read()
 *      synchronises and returns, per kernel, the launch count and the summed duration. */
typedef struct guber_kernel_time {
    char name[32];
    uint64_t launches;
    double total_ms;
    uint64_t units;      /* requests those launches processed (a fused launch carries several engines' batches) */
} guber_kernel_time_t;
int guber_profile_enable(guber_engine_t* e, int enable);
/* *n_out = the kernels the library times (17 today; it grows with the library), of which the first min(cap, *n_out) are filled. */
int guber_profile_read(guber_engine_t* e, guber_kernel_time_t* out, uint32_t cap, uint32_t* n_out);
/* After guber_profile_read(): the duration of every pipeline pass the profiled region made on this engine's stream — the launches
 * of one batch, or of one fused group of up to four tables' batches (recorded by the group's first engine) — from the first
 * kernel's start to the last kernel's end, in microseconds: what a batch spends on the GPU while the other streams are busy too.
 * us = NULL: only the count.  Reading clears the list. */
int guber_profile_passes(guber_engine_t* e, float* us, uint32_t cap, uint32_t* n_out);

/* ---- GLOBAL behaviour (global.go): the engine accumulates, per bucket, what the reference's
 *      globalManager keeps in its hitsQueue / broadcastQueue maps.  A request carrying
 *      GUBER_BEHAVIOR_GLOBAL with hits != 0 and no error is queued when it is evaluated:
 *        is_owner = 0  -> hits are summed per key; template = FIRST queued request; RESET_REMAINING is
 *                         OR-ed in (global.go:100-111)                                  role 1
 *        is_owner = 1  -> the key is marked for broadcast; template = LAST request (global.go:200)  role 2
 *      guber_global_take() returns one row per pending key of the requested roles and clears those queues (the flush of
 *      runAsyncHits / runBroadcasts, global.go:114-139, 217-232).  Buffers belong to the engine and stay
 *      valid until the next call.  The caller ships role-1 rows to the owning GPU (evaluate with is_owner
 *      = 1 and DRAIN_OVER_LIMIT, gubernator.go:510-512) and, for role-2 rows, re-reads the state with hits
 *      = 0 and installs it on the other GPUs with guber_add_items (gubernator.go:425-459). */
typedef struct guber_global_rows {
    uint32_t n;
    uint32_t key_stride;        /* key i = key_bytes[i*key_stride .. + key_len[i]) */
    const uint8_t* key_bytes;
    const uint32_t* key_len;
    const int64_t *hits, *limit, *duration, *burst, *created_at;
    const uint32_t* behavior;
    const uint8_t* algorithm;
    const uint8_t* role;
} guber_global_rows_t;
int guber_global_take(guber_engine_t* e, uint32_t role_mask /* bit 1: hits rows, bit 2: update rows */,
                      guber_global_rows_t* out);

/* ---- C++ host layer (gubernator_amd/csrc/worker_pool.h): GPUWorkerPool (micro-batching WorkerPool,
 *      workers.go:54-626 + peer_client.go:284-337 flush policy) and the V1Instance.GetRateLimits slice
 *      (gubernator.go:183-306: 1000-item cap, empty-field errors, CreatedAt default).  Exported for
 *      bindings and tests; thread-safe: any number of threads may call guber_pool_get_rate_limits. */
typedef struct guber_pool guber_pool_t;
int guber_pool_create(const guber_config_t* cfg, uint32_t batch_limit, uint32_t batch_wait_us, guber_pool_t** out);
/* `shards` = Config.Workers (config.go:110): the key space split by XXH64 range (workers.go:153-155,180-184) over that many
 * engines (tables) inside one GPU, whose batches share launches; cfg->cache_size is the pool total.  With 2 .. 16 shards per
 * device (GLOBAL engine included) the callers of a device share ONE set of stages and the GPU hands the requests to the shards
 * (guber_stage_submit_routed); batch_limit is per shard: a device's batch carries up to batch_limit x shards requests (at most
 * 65 536).  GUBER_POOL_ROUTED=0 in the environment: every shard has stages of its own and the callers sort by shard. */
int guber_pool_create_sharded(const guber_config_t* cfg, uint32_t shards, uint32_t batch_limit, uint32_t batch_wait_us,
                              guber_pool_t** out);
/* Several devices: devices[i] = HIP ordinal of peer "gpu<i>" on the reference's replicated consistent hash (replicated_hash.go:
 * 78-119, 512 vnodes, fnv1) — a key belongs to the device the ring assigns it to, and inside the device to the shard of the
 * worker rule above.  A mixed batch is split by owner on the host and every request is answered in its own slot
 * (functional_test.go:1638-1686).  The same ordinal may appear several times: logical devices on one GPU.  n_devices = 0 ->
 * {cfg->device}.  cfg->cache_size is the pool total. */
int guber_pool_create_multi(const guber_config_t* cfg, const int32_t* devices, uint32_t n_devices, uint32_t shards_per_device,
                            uint32_t batch_limit, uint32_t batch_wait_us, guber_pool_t** out);
uint32_t guber_pool_shards(guber_pool_t* p);
uint32_t guber_pool_device_of(guber_pool_t* p, const uint8_t* key, uint32_t key_len);   /* ReplicatedConsistentHash.Get over gpu0..gpuN-1 */
guber_engine_t* guber_pool_engine_at(guber_pool_t* p, uint32_t shard);
/* With GUBER_FLAG_GLOBAL in cfg->flags every device gets ONE more engine, which holds the keys of GLOBAL-behaviour requests (the
 * replica the GLOBAL manager synchronises); the plain shards are created without the flag.  guber_pool_global_sync = one
 * GlobalSyncWait tick (global.go:91-283) over the pool's devices; a daemon that is one rank of a multi-node ring builds its
 * own communicator over guber_pool_global_engine(device) (guber_comm_create_rank). */
struct guber_global_sync_stats;
guber_engine_t* guber_pool_global_engine(guber_pool_t* p, uint32_t device);
int guber_pool_global_sync(guber_pool_t* p, struct guber_global_sync_stats* stats /* optional */);
/* ask the dispatchers for a placement pass now (hot keys observed since the last one get a shard of their own choice; resident
 * buckets migrate at a batch boundary).  The periodic pass runs every GUBER_POOL_REBALANCE_MS (default 250; 0 = never). */
void guber_pool_rebalance(guber_pool_t* p);
/* batcher metrics: the pool's analogues of gubernator_batch_queue_length / gubernator_batch_send_duration (gubernator.go:96-107) */
typedef struct guber_pool_metrics {
    uint64_t batches, requests;            /* flushed so far */
    uint64_t queue_length;                 /* requests waiting in the shards' queues right now */
    uint64_t queue_length_max;             /* high-water mark of one shard's queue */
    uint64_t send_duration_us_sum;         /* flush start -> responses delivered, summed over batches (divide by `batches`) */
    uint64_t send_duration_us_max;
    uint64_t batch_size_max;
    uint64_t in_flight;                    /* batches submitted and not yet delivered */
    uint64_t key_too_long;                 /* requests answered GUBER_ITEM_E_KEY_TOO_LONG without touching the device */
    uint64_t flush_on_key_bytes;           /* batches flushed early because the next key did not fit the stage's key buffer */
    uint64_t rebalances;                   /* placement passes of the dispatchers */
    uint64_t keys_moved;                   /* hot keys whose bucket changed its logical shard */
    uint64_t submits, submit_us_sum;       /* dispatcher submissions (one may carry the batches of all shards) and the host time inside them */
    uint64_t direct_batches;               /* RPCs of a handful of requests evaluated by their caller's thread (counted in `batches` too) */
    uint32_t shards, devices;
} guber_pool_metrics_t;
int guber_pool_metrics(guber_pool_t* p, guber_pool_metrics_t* out);
void guber_pool_destroy(guber_pool_t* p);
void guber_pool_set_clock(guber_pool_t* p, int64_t now_ms);
/* Config.Store for the pool: the batcher drives guber_probe_missing / guber_add_items / guber_eval_batch_store and
 * calls back in the reference's order.  The request view carries what the reference hands to Store.Get /
 * Store.OnChange (`r *RateLimitReq`): key = name + "_" + unique_key, name = key[0..name_len). */
typedef struct guber_store_req {
    const uint8_t* key; uint32_t key_len; uint32_t name_len;
    int64_t hits, limit, duration, burst, created_at;
    int32_t algorithm; uint32_t behavior;
} guber_store_req_t;
typedef struct guber_store_callbacks {
    int (*get)(void* user, const guber_store_req_t* r, guber_item_t* out);            /* 1 = found, *out filled (key ignored) */
    void (*on_change)(void* user, const guber_store_req_t* r, const guber_item_t* item);
    void (*remove)(void* user, const uint8_t* key, uint32_t key_len);
    void* user;
} guber_store_callbacks_t;
void guber_pool_set_store(guber_pool_t* p, const guber_store_callbacks_t* cb);
/* WorkerPool.Load / WorkerPool.Store (workers.go:329-449, 451-534): the Loader's items into the cache at start-up; every
 * resident item to Loader.Save at shutdown (`save` is called once per item; item->key is valid during the call). */
uint32_t guber_pool_shard_of(guber_pool_t* p, const uint8_t* key, uint32_t key_len);   /* WorkerPool.getWorker, workers.go:180-184 */
int guber_pool_load(guber_pool_t* p, const guber_item_t* items, uint32_t n);
/* The same with a hint per item (NULL = none): global_hint[i] != 0 restores item i into its device's GLOBAL engine.  A CacheItem
 * carries no behaviour (cache.go:29-41), so guber_pool_load restores everything into the plain shards: a bucket that GLOBAL requests
 * had built up is then found by plain requests only, and the first GLOBAL request for its key starts a fresh bucket (as after a
 * restart without a Loader).  A Loader that persists where an item came from (guber_pool_store reports the GLOBAL engines' items
 * last) can pass it back here. */
int guber_pool_load_hinted(guber_pool_t* p, const guber_item_t* items, uint32_t n, const uint8_t* global_hint);
int guber_pool_store(guber_pool_t* p, void (*save)(void* user, const guber_item_t* item), void* user);      /* before the first request; NULL = none */   /* clock.Freeze of the reference tests; 0 = wall clock */
guber_engine_t* guber_pool_engine(guber_pool_t* p);
uint64_t guber_pool_batches(guber_pool_t* p);
/* names / unique keys as SoA strings; created_at[i] = 0 means unset; err_text (optional) receives the
 * reference's per-item error strings, err_stride bytes each (RPC-level error text in the first slot
 * when GUBER_E_BATCH_TOO_LARGE is returned). */
int guber_pool_get_rate_limits(guber_pool_t* p, uint32_t n, const uint8_t* name_bytes, const uint32_t* name_off,
                               const uint8_t* ukey_bytes, const uint32_t* ukey_off, const int64_t* hits,
                               const int64_t* limit, const int64_t* duration, const int64_t* burst,
                               const int64_t* created_at, const int32_t* algorithm, const uint32_t* behavior,
                               guber_result_t* out, char* err_text, uint32_t err_stride);
/* The same with RateLimitReqState.IsOwner per request (WorkerPool.GetRateLimit(ctx, req, reqState), workers.go:261; NULL = every
 * request owned): what V1Instance hands over for GLOBAL requests it answers from its replica (gubernator.go:395-421). */
int guber_pool_get_rate_limits_owner(guber_pool_t* p, uint32_t n, const uint8_t* name_bytes, const uint32_t* name_off,
                                     const uint8_t* ukey_bytes, const uint32_t* ukey_off, const int64_t* hits,
                                     const int64_t* limit, const int64_t* duration, const int64_t* burst,
                                     const int64_t* created_at, const int32_t* algorithm, const uint32_t* behavior,
                                     const uint8_t* is_owner, guber_result_t* out, char* err_text, uint32_t err_stride);
/* WorkerPool.AddCacheItem / GetCacheItem (workers.go:537-626) and the sum of the workers' cache sizes: the item goes to / comes
 * from the shard the placement gives its key */
int guber_pool_add_item(guber_pool_t* p, const guber_item_t* item);
/* With GUBER_FLAG_GLOBAL a key lives either in its device's GLOBAL engine (where requests with Behavior_GLOBAL are evaluated) or in
 * the plain shard its hash selects; the reference has ONE cache per worker and needs no such distinction.  guber_pool_add_item puts the
 * item where the key already is (the GLOBAL engine is asked first) and in the plain shard otherwise; guber_pool_get_item asks the
 * GLOBAL engine first.  guber_pool_add_item_for says which requests the item belongs to — `behavior` as in RateLimitReq: what
 * UpdatePeerGlobals (gubernator.go:425-459, the broadcast of an owner's GLOBAL state) must use, with GUBER_BEHAVIOR_GLOBAL. */
int guber_pool_add_item_for(guber_pool_t* p, const guber_item_t* item, uint32_t behavior);
int guber_pool_get_item(guber_pool_t* p, const uint8_t* key, uint32_t key_len, guber_item_t* out, int* found);
int64_t guber_pool_size(guber_pool_t* p);

/* ---- placement of one GPU's keys on its logical shards (replaces WorkerPool.getWorker, workers.go:180-184, where the
 *      reference picks the worker by XXH64 range).  key_hash = XXH64(HashKey, seed 0) as in workers.go:153-155.  Keys map to
 *      n_slots hash slots by the reference's rule (slot = hash63 / (2^63 / n_slots)); slots map to shards through a table
 *      that starts as contiguous runs (= getWorker exactly); keys that alone outweigh heavy_fraction of a shard's fair share
 *      are placed individually (at most 64).  Which shard holds a key never shows in a response.
 *        guber_placement_observe[_keys]   feed observed traffic (thread-safe, approximate: relaxed atomics)
 *        guber_placement_rebalance        longest-processing-time-first over what was observed since the last call.
 *                                         move_slots = 1: slots and hot keys are placed afresh — only while no key of this
 *                                         placement is resident (before the first request, offline);
 *                                         move_slots = 0: the slot table stays; keys that became heavy get the least loaded
 *                                         shard, and a key that has not been heavy for three passes in a row (each over at
 *                                         least 4096 observations) loses its individual place and follows its slot again, so
 *                                         that a drifting hot set never exhausts the 64 places.  moves[] lists the keys whose
 *                                         shard changed, either way (at most cap per pass; the rest waits for the next): the
 *                                         caller migrates their buckets (GPUWorkerPool does, at a batch boundary) before
 *                                         serving them there.
 *      Readers are wait-free; guber_placement_version changes with every rebalance. */
typedef struct guber_placement guber_placement_t;
typedef struct guber_placement_move { uint64_t key_hash; uint32_t from, to; } guber_placement_move_t;
int guber_placement_create(uint32_t n_shards, uint32_t n_slots /* 0 = 4096 */, guber_placement_t** out);
void guber_placement_destroy(guber_placement_t* p);
uint32_t guber_placement_shard(const guber_placement_t* p, uint64_t key_hash);
uint32_t guber_placement_version(const guber_placement_t* p);
int guber_placement_route_keys(const guber_placement_t* p, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t n,
                               uint32_t* shard_out /* optional */, uint64_t* hash_out /* optional */);
void guber_placement_observe(guber_placement_t* p, uint64_t key_hash, uint32_t weight);
int guber_placement_observe_keys(guber_placement_t* p, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t n);
int guber_placement_rebalance(guber_placement_t* p, double heavy_fraction /* <= 0: 0.125 */, int move_slots,
                              guber_placement_move_t* moves, uint32_t cap, uint32_t* n_moves);
int guber_placement_info(const guber_placement_t* p, uint32_t* n_shards, uint32_t* n_slots, uint32_t* n_hot);
/* guber_placement_rebalance(move_slots = 0) in two steps, for a caller that must quiesce between learning which RESIDENT keys
 * move and letting requests follow the new placement: plan changes nothing a reader sees; commit publishes the planned list
 * (version + 1) and starts a fresh observation round. */
int guber_placement_plan(guber_placement_t* p, double heavy_fraction, guber_placement_move_t* moves, uint32_t cap, uint32_t* n_moves);
int guber_placement_commit(guber_placement_t* p);
/* between plan and commit: drop one planned move (its bucket could not be migrated): the key stays where the published placement
 * has it (following its slot, or at its individual place if that was to be taken back) */
int guber_placement_cancel(guber_placement_t* p, uint64_t key_hash);
/* the published placement in the form guber_stage_route takes (global_engine = -1).  out->table points into the placement (valid
 * until it is destroyed; its entries are atomics that a rebalance with move_slots rewrites in place); out->ex_hash / ex_shard point
 * into the snapshot published by the LAST commit / rebalance and are valid only until the NEXT one on this placement — use the rule
 * at once (guber_stage_route copies it to the device before it returns) or copy the arrays; export again after every commit.
 * (Retired snapshots are kept alive for wait-free readers that loaded the pointer a few instructions before a publish — the last
 * 64 of them — not for holders of an exported rule.) */
int guber_placement_export(const guber_placement_t* p, struct guber_route_rule* out);
/* The buckets of the keys with these XXH64 hashes leave `from`'s table and enter `to`'s (two logical shards of ONE GPU), on
 * the device.  The caller guarantees that neither engine has a batch with those keys being formed or in flight.  *moved
 * (optional) = buckets that were live and moved. */
int guber_move_items_by_hash(guber_engine_t* from, guber_engine_t* to, const uint64_t* key_hashes, uint32_t n, uint32_t* moved);
/* The HIP stream the engine enqueues on (hipStream_t), to be handed to guber_config_t.stream of further engines: engines that
 * share device and stream can share launches (guber_stages_submit, guber_eval_batches_routed_dev).  The engine that owns the
 * stream must be destroyed last. */
void* guber_engine_stream(guber_engine_t* e);

/* ---- pinned staging memory for the cgo side (no Go pointers may be retained) */
void* guber_alloc_pinned(size_t bytes);
void guber_free_pinned(void* p);

/* ---- key -> shard routing = ReplicatedConsistentHash (replicated_hash.go:29-119).
 *      hash_kind 0 = fnv1 (library default, replicated_hash.go:33), 1 = fnv1a (config.go:429). */
typedef struct guber_ring guber_ring_t;
int guber_ring_create(const char* const* peer_names, uint32_t n_peers, uint32_t replicas,
                      int hash_kind, guber_ring_t** out);
void guber_ring_destroy(guber_ring_t* r);
/* owner[i] = index into peer_names of the peer owning key i (host arrays). */
int guber_ring_route(const guber_ring_t* r, const uint8_t* key_bytes, const uint32_t* key_off,
                     uint32_t n, uint32_t* owner);
/* Same on DEVICE arrays, enqueued on the engine's stream (router kernel: fnv1-64 + binary search
 * of the sorted ring staged in LDS). */
int guber_ring_route_dev(guber_engine_t* e, const guber_ring_t* r, const uint8_t* key_bytes,
                         const uint32_t* key_off, uint32_t n, uint32_t* owner);
uint32_t guber_ring_points(const guber_ring_t* r, uint64_t* hashes, uint32_t* owners, uint32_t cap);

/* ---- the same queues without leaving HBM (the GLOBAL exchange of one node runs GPU -> RCCL/xGMI -> GPU):
 *      guber_global_pending     upper bound of the rows a take can return (size the arrays with it)
 *      guber_global_take_dev    rows into caller-provided DEVICE arrays: key i = key_bytes[i*key_stride ..+key_len[i]),
 *                               key_stride >= max_key_bytes; *n_out rows written (GUBER_E_NOMEM + *n_out = rows needed
 *                               when cap is too small)
 *      guber_ring_route_rows_dev  owning peer of every row key (replicated_hash.go:104-119), asynchronous on the engine stream
 *      guber_add_items_dev      LRUCache.Add (lrucache.go:88-103) of device-resident item columns — the receiving side of
 *                               UpdatePeerGlobals (gubernator.go:425-459); keys must be distinct within a call;
 *                               result[i] (device) = 0 / 1 existed, 0xFF resubmit, 0xFE no directory entry; asynchronous */
typedef struct guber_global_rows_dev {
    uint32_t cap; uint32_t key_stride;
    uint8_t* key_bytes; uint32_t* key_len;
    int64_t *hits, *limit, *duration, *burst, *created_at;
    uint32_t* behavior; uint8_t* algorithm; uint8_t* role;
} guber_global_rows_dev_t;
typedef struct guber_items_dev {
    uint32_t n; uint32_t reserved;
    const uint8_t* key_bytes;    /* packed keys, readable 8 bytes past the end */
    const uint32_t* key_off;     /* n + 1 */
    const uint8_t* algorithm; const uint8_t* status;          /* status may be NULL (all UNDER_LIMIT) */
    const int64_t *limit, *duration, *remaining; const double* remaining_f;
    const int64_t *stamp, *burst, *expire_at, *invalid_at;    /* burst / invalid_at may be NULL (all 0) */
} guber_items_dev_t;
int guber_global_pending(guber_engine_t* e, uint32_t* n_out);
int guber_global_take_dev(guber_engine_t* e, uint32_t role_mask, const guber_global_rows_dev_t* out, uint32_t* n_out);
int guber_ring_route_rows_dev(guber_engine_t* e, const guber_ring_t* r, const uint8_t* key_rows, uint32_t key_stride,
                              const uint32_t* key_len, uint32_t n, uint32_t* owner);
int guber_add_items_dev(guber_engine_t* e, const guber_items_dev_t* items, uint8_t* result);

/* ---- GLOBAL behaviour across the GPUs of one node, natively (BASELINE config 5; global.go:91-283, gubernator.go:395-459,
 *      510-512).  Every GPU ("rank") holds a replica of the GLOBAL keys in an engine created with GUBER_FLAG_GLOBAL; ranks
 *      answer from their replica and queue hits for the owner (replicated consistent hash over the ranks, peer i = rank i).
 *      guber_global_sync is one GlobalSyncWait tick: pending hits -> owners (one exchange: grouped ncclSend / ncclRecv over
 *      RCCL / xGMI, every pair directly) -> owners apply them (IsOwner, GLOBAL => DRAIN_OVER_LIMIT) -> owners read back the
 *      state of every changed key (hits = 0) and build UpdatePeerGlobals items -> all ranks receive and install them.
 *      Rows stay in HBM end to end; the host waits only for the row counts.
 *        guber_comm_create_local   every rank lives in this process (a daemon driving all GPUs of the node):
 *                                  use_rccl = 1 -> ncclCommInitAll over the engines' devices (distinct GPUs);
 *                                  use_rccl = 0 -> device-to-device copies (also correct for several logical ranks on
 *                                  ONE GPU, which RCCL refuses: tests and single-GPU boxes)
 *        guber_comm_unique_id + guber_comm_create_rank   one rank per process (torchrun-style): rank 0 creates the
 *                                  128-byte id, the application distributes it, every rank joins with its engine
 *      RCCL is loaded with dlopen at first use (GUBER_RCCL_LIB overrides the name), so the library itself does not
 *      depend on it. */
typedef struct guber_comm guber_comm_t;
typedef struct guber_global_sync_stats {
    uint64_t hits_rows_sent;      /* aggregated non-owner hit rows leaving the local ranks (global.go:144-187) */
    uint64_t hits_rows_applied;   /* rows the local owners applied */
    uint64_t update_rows;         /* keys the local owners broadcast (global.go:234-283) */
    uint64_t items_installed;     /* items the local ranks installed from other owners (gubernator.go:425-459) */
    uint64_t bytes_moved;         /* bytes received by the local ranks in both exchanges */
    uint64_t fallbacks;           /* batches that needed the host path (64-bit hash collision inside the exchange) */
    double ms;                    /* wall time of the call */
} guber_global_sync_stats_t;
int guber_comm_create_local(guber_engine_t* const* engines, uint32_t n, const guber_ring_t* ring, int use_rccl, guber_comm_t** out);
int guber_comm_unique_id(uint8_t* id128);
int guber_comm_create_rank(guber_engine_t* e, uint32_t rank, uint32_t world, const uint8_t* id128, const guber_ring_t* ring,
                           guber_comm_t** out);
void guber_comm_destroy(guber_comm_t* c);
int guber_global_sync(guber_comm_t* c, int64_t now_ms, guber_global_sync_stats_t* stats);       /* stats (optional): summed over the local ranks */
int guber_comm_last_stats(guber_comm_t* c, uint32_t local_index, guber_global_sync_stats_t* out);   /* one local rank's share of the last sync */

/* ---- the daemon's time zone.  interval.go:97-142 build the civil dates of DURATION_IS_GREGORIAN intervals with now.Location():
 *      a "day" ends at 23:59:59.999 of the daemon's zone, not of UTC.  guber_set_timezone hands the engine that zone — the UTC
 *      offset in effect before the first listed transition and, per transition (ascending, at most 16), the instant in UTC seconds
 *      and the offset in seconds from then on: the shape of Go's time.Location — for the kernels (requests without greg_expire /
 *      greg_duration) and for the two host helpers below.  Civil times are resolved as Go's time.Date does (also inside the
 *      gap / overlap of a transition).  NULL, or n = 0 with offset0_s = 0: UTC (the default).  One zone per process; call it
 *      before traffic, or while no batch is in flight, e.g. once a year with the coming transitions. */
typedef struct { uint32_t n; int32_t offset0_s; const int64_t* when_s; const int32_t* offset_s; } guber_tz_t;
int guber_set_timezone(const guber_tz_t* tz);

/* ---- calendar helpers the host layer uses to fill greg_expire / greg_duration
 *      (interval.go:84-148), in the zone of guber_set_timezone. Return 0 or -GUBER_ITEM_E_GREGORIAN_*. */
int guber_gregorian_expiration(int64_t now_unix_nano, int64_t d, int64_t* expire_ms);
int guber_gregorian_duration(int64_t now_unix_nano, int64_t d, int64_t* duration);

/* ---- hashes on the path (third-party in the reference, see oracle/README.md) */
uint64_t guber_xxhash64(const uint8_t* p, size_t len, uint64_t seed); /* OneOfOne/xxhash, workers.go:154 */
uint64_t guber_fnv1_64(const uint8_t* p, size_t len);                 /* segmentio/fasthash fnv1 */
uint64_t guber_fnv1a_64(const uint8_t* p, size_t len);                /* segmentio/fasthash fnv1a */

const char* guber_strerror(int code);
const char* guber_item_strerror(uint8_t item_err);
const char* guber_last_error(void);
const char* guber_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GUBER_GPU_H */
