/* guber_wire.h — wire-format front end of the batched rate-limit path (C ABI, host code).
 *
 * What it replaces in the reference: the protobuf unmarshalling of `GetRateLimitsReq` /
 * `GetPeerRateLimitsReq` into one heap-allocated `*RateLimitReq` per item (generated code of
 * gubernator.proto:137-182 / peers.proto:36-44) followed by the per-item validation of
 * `V1Instance.GetRateLimits` (gubernator.go:189-220), and the marshalling of `GetRateLimitsResp` /
 * `GetPeerRateLimitsResp` (gubernator.proto:189-203, peers.proto:46-49).  Here the serialized RPC payload
 * is transcoded straight into the SoA batch `guber_eval_batch` consumes (include/guber_gpu.h) — no per-item
 * allocation — and the SoA results straight back into the serialized response.  Several RPC payloads can
 * be appended to ONE device batch, which lifts the reference's 1000-items-per-RPC cap (gubernator.go:40,189)
 * and 1 MiB receive limit (daemon.go:122) from the device batch size.
 *
 * Wire facts relied upon (proto3): both request messages are `repeated RateLimitReq = 1`, both response
 * messages are `repeated RateLimitResp = 1`, so one decoder / encoder serves the client and the peer RPC.
 *   RateLimitReq : name 1 (LEN), unique_key 2 (LEN), hits 3, limit 4, duration 5 (int64 varint),
 *                  algorithm 6, behavior 7 (enum varint), burst 8 (int64 varint),
 *                  metadata 9 (map entry, skipped: tracing propagation only), created_at 10 (optional int64)
 *   RateLimitResp: status 1 (enum), limit 2, remaining 3, reset_time 4 (int64), error 5 (string),
 *                  metadata 6 (never set on this path)
 * Unknown fields are skipped by wire type, the last occurrence of a singular field wins, zero-valued
 * response fields are omitted and fields are written in field-number order — the bytes equal what the
 * protobuf runtimes produce for the same message.
 *
 * Conventions: as guber_gpu.h (0 or negative code, never throws, caller-owned buffers, no pointer kept).
 */
#ifndef GUBER_WIRE_H
#define GUBER_WIRE_H

#include <stddef.h>
#include <stdint.h>

#include "guber_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GUBER_E_WIRE_MALFORMED (-20)   /* truncated / invalid protobuf; nothing was appended */
#define GUBER_E_WIRE_TOO_LARGE (-21)   /* more than max_per_rpc items: gubernator.go:189-193 (codes.OutOfRange) */
#define GUBER_E_WIRE_FULL (-22)        /* the batch has no room for this payload; nothing was appended: flush, reset, retry */
#define GUBER_E_WIRE_CLOSED (-23)      /* guber_wire_pool: the pool is being destroyed */
#define GUBER_PENDING 1                /* guber_wire_dev_*_collect(wait = 0): the GPU is still at it */

#define GUBER_WIRE_PINNED 1u           /* allocate the SoA with guber_alloc_pinned (needs a HIP device) */

/* per-item validation outcome of gubernator.go:207-217, kept next to the SoA */
#define GUBER_WIRE_PRE_OK 0
#define GUBER_WIRE_PRE_EMPTY_UNIQUE_KEY 1   /* "field 'unique_key' cannot be empty" */
#define GUBER_WIRE_PRE_EMPTY_NAME 2         /* "field 'namespace' cannot be empty" */

typedef struct guber_wire_batch guber_wire_batch_t;

/* A growable-up-to-capacity SoA batch plus the result arrays of its evaluation. */
int guber_wire_batch_create(uint32_t max_items, uint32_t max_key_bytes, uint32_t flags, guber_wire_batch_t** out);
void guber_wire_batch_destroy(guber_wire_batch_t* b);
/* Empty the batch and set the clock of the next evaluation: MillisecondNow() for expiry (lrucache.go:106)
 * and the CreatedAt default of items that carry none (gubernator.go:218-220). */
void guber_wire_batch_reset(guber_wire_batch_t* b, int64_t now_ms);
uint32_t guber_wire_batch_size(const guber_wire_batch_t* b);

/* Append every RateLimitReq of one serialized GetRateLimitsReq / GetPeerRateLimitsReq.
 *   max_per_rpc  0 = no cap; the reference passes 1000 (gubernator.go:40) -> GUBER_E_WIRE_TOO_LARGE
 *   is_owner     RateLimitReqState.IsOwner of these items (gubernator.go:246, 489)
 *   *first,*count  the slice [first, first+count) of the batch this payload occupies (responses are
 *                  positionally aligned, gubernator.proto:51-54)
 * Items failing the reference's validation keep their position with an empty key; they are answered with
 * the reference's error text by guber_wire_encode_responses and never reach a bucket. */
int guber_wire_decode_requests(guber_wire_batch_t* b, const uint8_t* msg, size_t len, uint32_t max_per_rpc,
                               uint8_t is_owner, uint32_t* first, uint32_t* count);

/* The SoA view to hand to guber_eval_batch (valid until the next reset / decode), the result arrays an
 * evaluation fills, and the per-item validation codes. */
const guber_batch_t* guber_wire_batch_view(guber_wire_batch_t* b);
guber_result_t* guber_wire_batch_result(guber_wire_batch_t* b);
const uint8_t* guber_wire_batch_pre_errors(const guber_wire_batch_t* b);

/* guber_eval_batch(e, view, result) on the batch's own arrays. */
int guber_wire_eval(guber_engine_t* e, guber_wire_batch_t* b);

/* Serialize the responses of the slice [first, first+count) as GetRateLimitsResp (= GetPeerRateLimitsResp).
 *   wrap_errors  1 = local path: errors of the evaluation are wrapped as
 *                "Error while apply rate limit for '<key>': <text>" (gubernator.go:250-255); 0 = peer path
 * Returns GUBER_E_NOMEM when cap is too small; *len then holds the size needed. */
size_t guber_wire_encode_bound(const guber_wire_batch_t* b, uint32_t first, uint32_t count);
int guber_wire_encode_responses(const guber_wire_batch_t* b, uint32_t first, uint32_t count, int wrap_errors,
                                uint8_t* out, size_t cap, size_t* len);

/* ---- UpdatePeerGlobals (peers.proto:51-63; sender global.go:234-283 broadcastPeers, receiver gubernator.go:425-459) ----
 * The third payload kind that touches the path: the owner of GLOBAL keys broadcasts their state, every other peer
 * installs it in its cache.
 *   UpdatePeerGlobalsReq: globals 1 (repeated UpdatePeerGlobal)
 *   UpdatePeerGlobal    : key 1 (string), status 2 (RateLimitResp), algorithm 3 (enum), duration 4, created_at 5 (int64)
 *
 * guber_wire_encode_globals  the sender side: one UpdatePeerGlobal per update row (key, algorithm, duration, created_at
 *     of the queued request — what guber_global_take role 2 returns) with `status` = the owner's hits = 0 read of the
 *     bucket; rows whose status read failed (status->err[i] != 0) are skipped (global.go:246-249).
 *     GUBER_E_NOMEM + *len = size needed when cap is too small.
 * guber_wire_decode_globals  the receiver side: the CacheItem of every global exactly as UpdatePeerGlobals builds it
 *     (ExpireAt = status.reset_time; leaky: Remaining = float64(status.remaining), Burst = Limit = status.limit,
 *     UpdatedAt = now; token: Status, Limit, Remaining from status, CreatedAt = now; Duration = the message's; any other
 *     algorithm: an item without a value), as a guber_item_t array ready for guber_add_items.  The array and the key
 *     bytes it points to belong to `items` and stay valid until the next decode into it or its destruction. */
typedef struct guber_wire_items guber_wire_items_t;
int guber_wire_items_create(uint32_t max_items, uint32_t max_key_bytes, guber_wire_items_t** out);
void guber_wire_items_destroy(guber_wire_items_t* items);
int guber_wire_decode_globals(guber_wire_items_t* items, const uint8_t* msg, size_t len, int64_t now_ms,
                              const guber_item_t** out, uint32_t* count);
int guber_wire_encode_globals(const uint8_t* key_bytes, const uint32_t* key_off, const uint8_t* algorithm,
                              const int64_t* duration, const int64_t* created_at, const guber_result_t* status, uint32_t n,
                              uint8_t* out, size_t cap, size_t* len);

/* ---- the same decode ON THE DEVICE (gubernator_amd/csrc/guber_kernels_wire.h): the serialized payloads of many RPCs -> one batch in
 *      HBM, evaluated where it is.  The reference unmarshals every RateLimitReq into a heap object on the CPU (generated code of
 *      gubernator.proto:137-182) and validates it there (gubernator.go:189-220); here a workgroup per 8 KB window of a payload finds its part of
 *      the record chain (a wave per payload walks it where it holds anything but plain records), a thread per item parses its record (the same source as guber_wire_decode_requests) and the items land in the structure of
 *      arrays the engine's kernels read.
 *   guber_wire_dev_create   a decoder bound to an engine: at most max_items items, max_payload_bytes payload bytes and max_rpcs
 *                           payloads per decode (max_items <= the engine's max_batch; an RPC holds at most min(max_items, 4096) items)
 *   guber_wire_dev_decode   nrpc payloads (msgs[r], lens[r]; is_owner[r] = RateLimitReqState.IsOwner of its items, NULL = all owner;
 *                           max_per_rpc as guber_wire_decode_requests) -> per RPC: status[r] (GUBER_OK, GUBER_E_WIRE_MALFORMED,
 *                           GUBER_E_WIRE_TOO_LARGE), first[r], count[r] = the slice of the batch its items occupy (responses are
 *                           positionally aligned); *n_items = the batch size.  Items of a rejected RPC that were already placed stay
 *                           as dead slots (pre_err GUBER_WIRE_PRE_DEAD, empty key): they never reach a bucket.
 *   guber_wire_dev_buffer / guber_wire_dev_decode_staged   the decoder's own pinned staging buffer, and the decode of payloads that
 *                           already lie in it (offs[r] 16-byte aligned, ascending, not overlapping, 16 bytes of room behind the
 *                           last): a receive path that reads its sockets straight into the buffer skips guber_wire_dev_decode's copy
 *   guber_wire_dev_eval     the batch through the engine (as guber_eval_batch_dev), results to host arrays of n_items entries
 *   guber_wire_dev_eval_front  the batch through a front over several engines: decode, routing, evaluation and the answers' order in HBM
 *   guber_wire_dev_columns  the decoded columns copied to host memory (keys as rows of key_stride bytes + key_len): what
 *                           guber_wire_encode_responses-style code and the tests read */
#define GUBER_WIRE_PRE_DEAD 255
typedef struct guber_wire_dev guber_wire_dev_t;
typedef struct {
    uint32_t n, key_stride;
    const uint8_t* key_rows; const uint32_t* key_len;
    const int64_t *hits, *limit, *duration, *burst, *created_at;
    const uint32_t* behavior; const int32_t* algo_raw;
    const uint8_t *algorithm, *is_owner, *pre_err;
} guber_wire_columns_t;
int guber_wire_dev_create(guber_engine_t* e, uint32_t max_items, uint32_t max_payload_bytes, uint32_t max_rpcs, guber_wire_dev_t** out);
void guber_wire_dev_destroy(guber_wire_dev_t* d);
int guber_wire_dev_decode(guber_wire_dev_t* d, const uint8_t* const* msgs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                          uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items);
int guber_wire_dev_buffer(guber_wire_dev_t* d, uint8_t** buf, size_t* cap);
int guber_wire_dev_decode_staged(guber_wire_dev_t* d, const uint32_t* offs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                 uint32_t max_per_rpc, int64_t now_ms, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items);
int guber_wire_dev_eval(guber_wire_dev_t* d, guber_result_t* r);
/* the decoded batch through a front (include/guber_gpu.h guber_front_*) instead: routed to the front's engines on the device, evaluated there,
 * answered in the order of the RPCs' items; results to host arrays of n_items entries (n_items <= the front's max_n) */
int guber_wire_dev_eval_front(guber_wire_dev_t* d, guber_front_t* f, guber_result_t* r);
int guber_wire_dev_columns(guber_wire_dev_t* d, guber_wire_columns_t* c);
/* The same in two halves each, for a caller that keeps several decoders busy and never blocks on the GPU (the payload stage below):
 *   guber_wire_dev_set_stream           the decode's copies and kernels go to `stream` (hipStream_t) instead of the engine's stream
 *   guber_wire_dev_decode_staged_async  guber_wire_dev_decode_staged up to its last enqueue (is_owner stays valid until collected)
 *   guber_wire_dev_decode_collect       wait = 0: GUBER_PENDING while the GPU is at it; GUBER_OK: the verdicts, as guber_wire_dev_decode_staged
 *   guber_wire_dev_eval_front_async     guber_wire_dev_eval_front, enqueued only; r's arrays are DEVICE-VISIBLE (HBM, or pinned host memory the
 *                                       answers' last hop writes in place over PCIe): no copy follows
 *   guber_wire_dev_eval_collect         wait = 0: GUBER_PENDING while the answers are on their way
 *   guber_wire_dev_route_front_async    optional, before guber_wire_dev_eval_front_async: only the front's routing of the decoded batch, enqueued;
 *   guber_wire_dev_route_ready          GUBER_PENDING until the shares' sizes are in host memory (the evaluation's enqueue then waits for nothing).
 *                                       Nothing else may go through the front between the two calls. */
int guber_wire_dev_set_stream(guber_wire_dev_t* d, void* stream);
int guber_wire_dev_decode_staged_async(guber_wire_dev_t* d, const uint32_t* offs, const uint32_t* lens, uint32_t nrpc, const uint8_t* is_owner,
                                       uint32_t max_per_rpc, int64_t now_ms);
int guber_wire_dev_decode_collect(guber_wire_dev_t* d, int wait, int32_t* status, uint32_t* first, uint32_t* count, uint32_t* n_items);
int guber_wire_dev_eval_front_async(guber_wire_dev_t* d, guber_front_t* f, guber_result_t* r);
int guber_wire_dev_eval_collect(guber_wire_dev_t* d, int wait);
int guber_wire_dev_route_front_async(guber_wire_dev_t* d, guber_front_t* f);
int guber_wire_dev_route_ready(guber_wire_dev_t* d, guber_front_t* f);

/* ---- the payload stage: V1Instance.GetRateLimits / GetPeerRateLimits on the SERIALIZED messages (gubernator.go:183-306, :470-520; the batching
 *      shape is peer_client.go:284-337's: a queue that leaves when it is full or BatchWait after its first entry — here RPCs are the entries).
 *      Caller threads (the gRPC handlers: a codec that hands the handler the raw bytes) call guber_wire_pool_get_rate_limits concurrently; per RPC the
 *      host does one compare-and-swap (a place in the open stage) and two memcpys (the payload into pinned memory, the response out of it).
 *      Unmarshalling, validation, the CreatedAt default, HashKey, the worker's choice by XXH64 (workers.go:180-184), the evaluation, the
 *      answers' order and the marshalling of GetRateLimitsResp (gubernator.proto:184-203) all happen on the device: k_wire_* -> guber_front ->
 *      k_wire_enc, which writes every RPC's response bytes in place into host memory.  (An RPC with an item error is marshalled by the host
 *      transcoder from the raw answers: the error's text needs the item's key.)  An RPC of at most FOUR requests that finds at most eight calls inside the
 *      pool skips the stages: its caller evaluates it through the engine's one-launch path (host transcoder, the placement's rule on the host,
 *      guber_eval_batch) — 12 us instead of a stage's 90; same bytes.
 *   guber_wire_pool_create            over the engines of ONE device and the placement's rule (as guber_front_create; rule NULL with one engine);
 *                                     cfg NULL or zero fields = defaults.  Two threads per pool (intake, front); neither ever blocks on the GPU.
 *   guber_wire_pool_get_rate_limits   req/len: one serialized GetRateLimitsReq (= GetPeerRateLimitsReq); is_owner: RateLimitReqState.IsOwner of its
 *                                     items; wrap_errors as guber_wire_encode_responses.  Blocks until the stage the payload joined has been
 *                                     through the GPU.  resp/cap: at least guber_wire_pool_response_bound(req, len) bytes — a smaller buffer is
 *                                     refused before anything is evaluated when it cannot even hold error-free answers; when it turns out too small
 *                                     for the error texts the call returns GUBER_E_NOMEM AFTER the decisions have been applied.
 *                                     GUBER_E_WIRE_MALFORMED / GUBER_E_WIRE_TOO_LARGE: the message is turned away whole (nothing of it evaluated),
 *                                     as the protobuf runtime / gubernator.go:189-193 do.  The bytes equal guber_wire_encode_responses' (and the
 *                                     protobuf runtimes').
 *   guber_wire_pool_set_clock         0 = the wall clock (clock.Now(): a stage's items share the instant it was sealed); otherwise a frozen clock
 *                                     in ms, as the reference's tests use clock.Freeze
 *   What this surface does NOT do: the Store's write-through callbacks (store.go:49-65: guber_pool_set_store / guber_eval_batch_store — a daemon
 *   with conf.Store keeps the per-request pool), metadata propagation (RateLimitReq.metadata is skipped), and the decision WHERE a payload goes
 *   (forwarding to the owning peer, gubernator.go:236-283: the Go front end hands over only what this instance evaluates).  Behavior_GLOBAL items
 *   are evaluated on their table and queued for the GLOBAL exchange by the engine as on every other entry point (guber_global_take / _sync);
 *   a rule whose global_engine is set sends them to the device's GLOBAL engine. */
typedef struct guber_wire_pool guber_wire_pool_t;
typedef struct guber_wire_pool_config {
    uint32_t stages;             /* payload stages in rotation (one fills while the others are on the GPU); 0 = 12, 2 .. 12 */
    uint32_t max_items;          /* items a stage holds; 0 = 49 152 — what the front evaluates as one pair of launches for all tables (<= 1 048 575) */
    uint32_t max_payload_bytes;  /* payload bytes a stage holds; 0 = 8 MiB (<= 16 MiB) */
    uint32_t max_rpcs;           /* RPCs a stage holds; 0 = 1024 (<= 4095) */
    uint32_t batch_wait_us;      /* a stage leaves at the latest this long after its first payload; 0 = 500 (BatchWait, config.go:131) */
    uint32_t max_per_rpc;        /* 0 = 1000 (gubernator.go:40); 0xffffffff = no cap */
    uint32_t spin_us;            /* how long a waiting caller looks before it sleeps — when there are CPUs to look with: callers beyond half of them sleep at once; 0 = 150 */
    uint32_t decodes_queued;     /* a stage leaves early — before it is full or BatchWait is over — while fewer than this many stages are waiting for or
                                  * in their decode: 1 = one decode at a time (smallest latency under light load), 0 = 2 (the decoder's stream never idles) */
} guber_wire_pool_config_t;
typedef struct guber_wire_pool_stats {
    uint64_t rpcs, items, stages;                       /* answered so far */
    uint64_t sealed_full, sealed_wait, sealed_idle;     /* why stages left: full / BatchWait / the decoder was idle */
    uint64_t open_waits;                                /* callers that slept because every stage was on the GPU */
    uint64_t fill_us_sum, decode_us_sum, eval_us_sum;   /* per stage: first payload -> sealed; sealed -> decoded; decoded -> answers in host memory */
    uint64_t host_decode_ns, host_route_ns, host_eval_ns; /* the pool's threads' own time inside the three enqueueing calls */
} guber_wire_pool_stats_t;
int guber_wire_pool_create(guber_engine_t* const* engines, uint32_t n_engines, const struct guber_route_rule* rule, const guber_wire_pool_config_t* cfg,
                           guber_wire_pool_t** out);
void guber_wire_pool_destroy(guber_wire_pool_t* p);     /* no call may be in flight or arrive any more */
int guber_wire_pool_get_rate_limits(guber_wire_pool_t* p, const uint8_t* req, size_t len, int is_owner, int wrap_errors, uint8_t* resp, size_t cap,
                                    size_t* resp_len);
size_t guber_wire_pool_response_bound(const uint8_t* req, size_t len);
int guber_wire_pool_set_clock(guber_wire_pool_t* p, int64_t now_ms);
int guber_wire_pool_stats(guber_wire_pool_t* p, guber_wire_pool_stats_t* out);

#ifdef __cplusplus
}
#endif
#endif
