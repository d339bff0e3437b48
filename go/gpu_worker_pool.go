// gpu_worker_pool.go — cgo binding of the MI355X rate-limit engine for mailgun/gubernator v2.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (there is no Go toolchain here); it is the file a gubernator
// maintainer drops into the `gubernator` package (build tag `gpu`) to replace `WorkerPool`
// (workers.go:54-626) with the engine behind include/guber_gpu.h.  The method set is the one
// `V1Instance` calls today:
//
//	GetRateLimit(ctx, *RateLimitReq, RateLimitReqState) (*RateLimitResp, error)   gubernator.go:598, global.go:245
//	AddCacheItem(ctx, key, *CacheItem) error                                       gubernator.go:452
//	GetCacheItem(ctx, key) (*CacheItem, bool, error)                               workers.go:583
//	Load(ctx) error / Store(ctx) error / Close() error                             gubernator.go:143,161,169
//
// Requests from any number of gRPC goroutines are routed to their key's shard (the reference's worker rule) and
// collected there by one batcher goroutine (same policy as peer_client.go:284-337: flush at BatchLimit items or
// after BatchWait) into C-owned pinned SoA buffers and evaluated with ONE guber_eval_batch call.

//go:build gpu

package gubernator

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/lib -lguber_hip
#include <stdlib.h>
#include "guber_gpu.h"
*/
import "C"

import (
	"context"
	"fmt"
	"runtime"
	"time"
	"unsafe"

	"github.com/mailgun/holster/v4/clock"
	"github.com/pkg/errors"
)

type gpuRequest struct {
	req   *RateLimitReq
	state RateLimitReqState
	resp  chan gpuResponse
}
type gpuResponse struct {
	rl  *RateLimitResp
	err error
}

// GPUWorkerPool satisfies the call surface of *WorkerPool.  Like WorkerPool it splits the key space over
// conf.Workers shards by hash range (workers.go:125-151,180-184); a shard here is one engine (HBM table + HIP
// stream) with its own batcher goroutine, so that batches of different shards overlap on the GPU — 4 saturate an
// MI355X (bench.py --shards).
type GPUWorkerPool struct {
	conf         *Config
	hasher       workerHasher // workers.go:70-72
	hashRingStep uint64       // workers.go:132
	shards       []*gpuShard
}

// gpuShard = one "worker" of the reference: single writer of its own cache.
type gpuShard struct {
	conf   *Config
	engine *C.guber_engine_t
	queue  chan gpuRequest
	done   chan struct{}
	// pinned SoA staging (C memory: cgo forbids the callee to keep Go pointers)
	cap                                     int
	keyBytes                                *C.uint8_t
	keyOff                                  *C.uint32_t
	hits, limit, duration, burst, createdAt *C.int64_t
	gregExpire, gregDuration                *C.int64_t
	algorithm, isOwner                      *C.uint8_t
	behavior                                *C.uint32_t
	status, errCode                         *C.uint8_t
	rLimit, rRemaining, rReset              *C.int64_t
	// Config.Store side channel (only allocated when conf.Store != nil)
	missing, storeFlags *C.uint8_t
	storeItems          *C.guber_item_t
}

func NewGPUWorkerPool(conf *Config, device int, batchLimit int, batchWait time.Duration) (*GPUWorkerPool, error) {
	workers := conf.Workers // config.go:110; default = NumCPU, GUBER_GPU_SHARDS overrides it for the GPU pool (4 is enough)
	if workers <= 0 {
		workers = 1
	}
	p := &GPUWorkerPool{conf: conf, hasher: &hasher{}, hashRingStep: uint64(1<<63) / uint64(workers)} // workers.go:80,132
	for i := 0; i < workers; i++ {
		sh, err := newGPUShard(conf, device, conf.CacheSize/workers+1, batchLimit, batchWait) // workers.go:132
		if err != nil {
			_ = p.Close()
			return nil, err
		}
		p.shards = append(p.shards, sh)
	}
	return p, nil
}

// shardOf = WorkerPool.getWorker (workers.go:180-184)
func (p *GPUWorkerPool) shardOf(key string) *gpuShard {
	return p.shards[p.hasher.ComputeHash63(key)/p.hashRingStep]
}

func (p *GPUWorkerPool) GetRateLimit(ctx context.Context, r *RateLimitReq, s RateLimitReqState) (*RateLimitResp, error) {
	return p.shardOf(r.HashKey()).GetRateLimit(ctx, r, s)
}
func (p *GPUWorkerPool) AddCacheItem(ctx context.Context, key string, item *CacheItem) error {
	return p.shardOf(key).AddCacheItem(ctx, key, item)
}
func (p *GPUWorkerPool) GetCacheItem(ctx context.Context, key string) (*CacheItem, bool, error) {
	return p.shardOf(key).GetCacheItem(ctx, key)
}

// Load drains Loader.Load() into the shards in bulk (workers.go:329-413).
func (p *GPUWorkerPool) Load(ctx context.Context) error {
	ch, err := p.conf.Loader.Load()
	if err != nil {
		return errors.Wrap(err, "Error in loader.Load")
	}
	pending := make(map[*gpuShard][]C.guber_item_t, len(p.shards))
	for item := range ch {
		sh := p.shardOf(item.Key)
		pending[sh] = append(pending[sh], toCItem(item.Key, item))
		if len(pending[sh]) >= 4096 {
			if err := sh.addItems(pending[sh]); err != nil {
				return err
			}
			pending[sh] = pending[sh][:0]
		}
	}
	for sh, items := range pending {
		if err := sh.addItems(items); err != nil {
			return err
		}
	}
	return nil
}

// Store hands every resident item of every shard to Loader.Save (workers.go:451-534).
func (p *GPUWorkerPool) Store(ctx context.Context) error {
	out := make(chan *CacheItem, 500)
	errc := make(chan error, 1)
	go func() {
		defer close(out)
		for _, sh := range p.shards {
			if err := sh.dump(out); err != nil {
				errc <- err
				return
			}
		}
		errc <- nil
	}()
	if err := p.conf.Loader.Save(out); err != nil {
		return errors.Wrap(err, "Error in loader.Save")
	}
	return <-errc
}

func (p *GPUWorkerPool) Close() error {
	for _, sh := range p.shards {
		_ = sh.Close()
	}
	return nil
}

func newGPUShard(conf *Config, device int, cacheSize int, batchLimit int, batchWait time.Duration) (*gpuShard, error) {
	cfg := C.guber_config_t{struct_size: C.uint32_t(unsafe.Sizeof(C.guber_config_t{})), device: C.int32_t(device),
		cache_size: C.uint64_t(cacheSize), max_batch: C.uint32_t(batchLimit)}
	p := &gpuShard{conf: conf, queue: make(chan gpuRequest, batchLimit), done: make(chan struct{}), cap: batchLimit}
	if rc := C.guber_engine_create(&cfg, &p.engine); rc != C.GUBER_OK {
		return nil, fmt.Errorf("guber_engine_create: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
	}
	n := C.size_t(batchLimit)
	p.keyBytes = (*C.uint8_t)(C.guber_alloc_pinned(n*256 + 16))
	p.keyOff = (*C.uint32_t)(C.guber_alloc_pinned((n + 1) * 4))
	alloc64 := func() *C.int64_t { return (*C.int64_t)(C.guber_alloc_pinned(n * 8)) }
	p.hits, p.limit, p.duration, p.burst, p.createdAt = alloc64(), alloc64(), alloc64(), alloc64(), alloc64()
	p.gregExpire, p.gregDuration, p.rLimit, p.rRemaining, p.rReset = alloc64(), alloc64(), alloc64(), alloc64(), alloc64()
	p.algorithm, p.isOwner = (*C.uint8_t)(C.guber_alloc_pinned(n)), (*C.uint8_t)(C.guber_alloc_pinned(n))
	p.status, p.errCode = (*C.uint8_t)(C.guber_alloc_pinned(n)), (*C.uint8_t)(C.guber_alloc_pinned(n))
	p.behavior = (*C.uint32_t)(C.guber_alloc_pinned(n * 4))
	if conf.Store != nil {
		p.missing, p.storeFlags = (*C.uint8_t)(C.guber_alloc_pinned(n)), (*C.uint8_t)(C.guber_alloc_pinned(n))
		p.storeItems = (*C.guber_item_t)(C.guber_alloc_pinned(n * C.sizeof_guber_item_t))
	}
	go p.run(batchLimit, batchWait)
	return p, nil
}

// GetRateLimit enqueues the request and waits for its batch (workers.go:261-291 semantics: ctx honoured
// at both waits).
func (p *gpuShard) GetRateLimit(ctx context.Context, r *RateLimitReq, s RateLimitReqState) (*RateLimitResp, error) {
	g := gpuRequest{req: r, state: s, resp: make(chan gpuResponse, 1)}
	select {
	case p.queue <- g:
	case <-ctx.Done():
		return nil, ctx.Err()
	}
	select {
	case out := <-g.resp:
		return out.rl, out.err
	case <-ctx.Done():
		return nil, ctx.Err()
	}
}

func (p *gpuShard) run(limit int, wait time.Duration) {
	runtime.LockOSThread() // one OS thread owns the HIP context
	pending := make([]gpuRequest, 0, limit)
	timer := time.NewTimer(wait)
	for {
		select {
		case g := <-p.queue:
			pending = append(pending, g)
			if len(pending) == 1 {
				timer.Reset(wait)
			}
			if len(pending) >= limit {
				p.flush(pending)
				pending = pending[:0]
			}
		case <-timer.C:
			if len(pending) > 0 {
				p.flush(pending)
				pending = pending[:0]
			}
		case <-p.done:
			return
		}
	}
}

func at64(p *C.int64_t, i int) *C.int64_t { return (*C.int64_t)(unsafe.Add(unsafe.Pointer(p), i*8)) }
func at8(p *C.uint8_t, i int) *C.uint8_t   { return (*C.uint8_t)(unsafe.Add(unsafe.Pointer(p), i)) }

func (p *gpuShard) flush(batch []gpuRequest) {
	now := clock.Now()
	nowMs := now.UnixNano() / 1000000
	off := 0
	for i, g := range batch {
		r := g.req
		key := r.HashKey() // client.go:39-41
		C.memcpy(unsafe.Add(unsafe.Pointer(p.keyBytes), off), unsafe.Pointer(unsafe.StringData(key)), C.size_t(len(key)))
		*(*C.uint32_t)(unsafe.Add(unsafe.Pointer(p.keyOff), i*4)) = C.uint32_t(off)
		off += len(key)
		*at64(p.hits, i), *at64(p.limit, i), *at64(p.duration, i) = C.int64_t(r.Hits), C.int64_t(r.Limit), C.int64_t(r.Duration)
		*at64(p.burst, i), *at64(p.createdAt, i) = C.int64_t(r.Burst), C.int64_t(*r.CreatedAt)
		alg := r.Algorithm
		if alg < 0 || alg > 1 {
			alg = 255 // workers.go:317: the engine answers GUBER_ITEM_E_INVALID_ALGORITHM
		}
		*at8(p.algorithm, i) = C.uint8_t(alg)
		*(*C.uint32_t)(unsafe.Add(unsafe.Pointer(p.behavior), i*4)) = C.uint32_t(r.Behavior)
		owner := C.uint8_t(0)
		if g.state.IsOwner {
			owner = 1
		}
		*at8(p.isOwner, i) = owner
		*at64(p.gregExpire, i), *at64(p.gregDuration, i) = 0, 0
		if HasBehavior(r.Behavior, Behavior_DURATION_IS_GREGORIAN) { // interval.go:84-148, evaluated once per batch
			var e, d C.int64_t
			if rc := C.guber_gregorian_expiration(C.int64_t(now.UnixNano()), C.int64_t(r.Duration), &e); rc != 0 {
				d = C.int64_t(rc) // negative = the reference's error, surfaced only on the paths that call it
			} else if rc := C.guber_gregorian_duration(C.int64_t(now.UnixNano()), C.int64_t(r.Duration), &d); rc != 0 {
				d = C.int64_t(rc)
			}
			*at64(p.gregExpire, i), *at64(p.gregDuration, i) = e, d
		}
	}
	*(*C.uint32_t)(unsafe.Add(unsafe.Pointer(p.keyOff), len(batch)*4)) = C.uint32_t(off)
	b := C.guber_batch_t{n: C.uint32_t(len(batch)), key_bytes: p.keyBytes, key_off: p.keyOff, hits: p.hits, limit: p.limit,
		duration: p.duration, burst: p.burst, created_at: p.createdAt, algorithm: p.algorithm, behavior: p.behavior,
		is_owner: p.isOwner, greg_expire: p.gregExpire, greg_duration: p.gregDuration, now_ms: C.int64_t(nowMs)}
	res := C.guber_result_t{status: p.status, limit: p.rLimit, remaining: p.rRemaining, reset_time: p.rReset, err: p.errCode}
	var rc C.int
	if p.conf.Store == nil {
		rc = C.guber_eval_batch(p.engine, &b, &res)
	} else {
		rc = p.evalWithStore(batch, &b, &res)
	}
	// prometheus: the engine returns the per-batch aggregates of the reference's counters
	metricOverLimitCounter.Add(float64(res.over_limit_count))             // algorithms.go:165,185,243,391,409,471
	metricCacheAccess.WithLabelValues("hit").Add(float64(res.cache_hits)) // lrucache.go:117,121,126
	metricCacheAccess.WithLabelValues("miss").Add(float64(res.cache_misses))
	metricCacheSize.Set(float64(res.cache_size))
	for i, g := range batch {
		if rc != C.GUBER_OK {
			g.resp <- gpuResponse{nil, errors.Errorf("gpu engine: %s", C.GoString(C.guber_strerror(rc)))}
			continue
		}
		if e := *at8(p.errCode, i); e != 0 {
			msg := C.GoString(C.guber_item_strerror(C.uint8_t(e)))
			if e == C.GUBER_ITEM_E_INVALID_ALGORITHM {
				msg = fmt.Sprintf(msg, g.req.Algorithm) // "Invalid rate limit algorithm '%d'"
			}
			g.resp <- gpuResponse{nil, errors.New(msg)}
			continue
		}
		g.resp <- gpuResponse{&RateLimitResp{Status: Status(*at8(p.status, i)), Limit: int64(*at64(p.rLimit, i)),
			Remaining: int64(*at64(p.rRemaining, i)), ResetTime: int64(*at64(p.rReset, i))}, nil}
	}
}

// evalWithStore is the Config.Store path (store.go:49-65).  The reference calls the store from inside the
// algorithms; here the engine reports which calls are due and this function makes them, in the same order:
//   Store.Get      for the first request of every key that is not resident before the batch (algorithms.go:45-51)
//   Store.Remove   token RESET_REMAINING / algorithm switched                                 (:79-84, :96-100, :311-315)
//   Store.OnChange with the CacheItem as it is right after THAT request, owner only           (:149-153, :252-254, ...)
func (p *gpuShard) evalWithStore(batch []gpuRequest, b *C.guber_batch_t, res *C.guber_result_t) C.int {
	ctx := context.Background()
	if rc := C.guber_probe_missing(p.engine, b, p.missing); rc != C.GUBER_OK {
		return rc
	}
	asked := map[string]struct{}{}
	for i, g := range batch {
		if *at8(p.missing, i) == 0 {
			continue
		}
		key := g.req.HashKey()
		if _, dup := asked[key]; dup {
			continue
		}
		asked[key] = struct{}{}
		if item, ok := p.conf.Store.Get(ctx, g.req); ok {
			if err := p.AddCacheItem(ctx, key, item); err != nil {
				return C.GUBER_E_HIP
			}
		}
	}
	ev := C.guber_store_events_t{flags: p.storeFlags, items: p.storeItems}
	rc := C.guber_eval_batch_store(p.engine, b, res, &ev)
	if rc != C.GUBER_OK {
		return rc
	}
	for i, g := range batch {
		f := *at8(p.storeFlags, i)
		if f&C.GUBER_STORE_REMOVE != 0 {
			p.conf.Store.Remove(ctx, g.req.HashKey())
		}
		if f&C.GUBER_STORE_ONCHANGE != 0 {
			ci := (*C.guber_item_t)(unsafe.Add(unsafe.Pointer(p.storeItems), i*C.sizeof_guber_item_t))
			p.conf.Store.OnChange(ctx, g.req, fromCItem(g.req.HashKey(), ci))
		}
	}
	return C.GUBER_OK
}

// AddCacheItem = LRUCache.Add through the engine (UpdatePeerGlobals, gubernator.go:425-459).
func (p *gpuShard) AddCacheItem(ctx context.Context, key string, item *CacheItem) error {
	ci := toCItem(key, item)
	defer C.free(unsafe.Pointer(ci.key))
	if rc := C.guber_add_items(p.engine, &ci, 1, nil); rc != C.GUBER_OK {
		return errors.Errorf("guber_add_items: %s", C.GoString(C.guber_strerror(rc)))
	}
	return nil
}

// GetCacheItem = LRUCache.GetItem (expired items are removed and reported absent).
func (p *gpuShard) GetCacheItem(ctx context.Context, key string) (*CacheItem, bool, error) {
	ck := C.CString(key)
	defer C.free(unsafe.Pointer(ck))
	var out C.guber_item_t
	var found C.int
	if rc := C.guber_get_item(p.engine, (*C.uint8_t)(unsafe.Pointer(ck)), C.uint32_t(len(key)), C.int64_t(MillisecondNow()), &out, &found); rc != C.GUBER_OK {
		return nil, false, errors.Errorf("guber_get_item: %s", C.GoString(C.guber_strerror(rc)))
	}
	if found == 0 {
		return nil, false, nil
	}
	return fromCItem(key, &out), true, nil
}

// addItems = LRUCache.Add for a chunk of loaded items; frees the C key copies made by toCItem.
func (p *gpuShard) addItems(items []C.guber_item_t) error {
	if len(items) == 0 {
		return nil
	}
	rc := C.guber_add_items(p.engine, &items[0], C.uint32_t(len(items)), nil)
	for i := range items {
		C.free(unsafe.Pointer(items[i].key))
	}
	if rc != C.GUBER_OK {
		return errors.Errorf("guber_add_items: %s", C.GoString(C.guber_strerror(rc)))
	}
	return nil
}

// dump sends every resident item of this shard to `out` (lrucache.go:76-85 Each).
func (p *gpuShard) dump(out chan<- *CacheItem) error {
	var n, arena C.uint64_t
	C.guber_dump(p.engine, nil, 0, nil, 0, &n, &arena) // sizes
	items := make([]C.guber_item_t, int(n)+16)
	keys := C.malloc(C.size_t(arena) + 1024)
	defer C.free(keys)
	if rc := C.guber_dump(p.engine, &items[0], C.uint64_t(len(items)), (*C.uint8_t)(keys), arena+1024, &n, &arena); rc != C.GUBER_OK {
		return errors.Errorf("guber_dump: %s", C.GoString(C.guber_strerror(rc)))
	}
	for i := 0; i < int(n); i++ {
		out <- fromCItem(C.GoStringN((*C.char)(unsafe.Pointer(items[i].key)), C.int(items[i].key_len)), &items[i])
	}
	return nil
}

func (p *gpuShard) Close() error {
	close(p.done)
	C.guber_engine_destroy(p.engine)
	return nil
}

func toCItem(key string, item *CacheItem) C.guber_item_t {
	ci := C.guber_item_t{algorithm: C.uint8_t(item.Algorithm), key_len: C.uint32_t(len(key)),
		key: (*C.uint8_t)(unsafe.Pointer(C.CString(key))), expire_at: C.int64_t(item.ExpireAt), invalid_at: C.int64_t(item.InvalidAt)}
	switch v := item.Value.(type) {
	case *TokenBucketItem:
		ci.status, ci.limit, ci.duration, ci.remaining, ci.stamp = C.uint8_t(v.Status), C.int64_t(v.Limit), C.int64_t(v.Duration), C.int64_t(v.Remaining), C.int64_t(v.CreatedAt)
	case *LeakyBucketItem:
		ci.limit, ci.duration, ci.remaining_f, ci.stamp, ci.burst = C.int64_t(v.Limit), C.int64_t(v.Duration), C.double(v.Remaining), C.int64_t(v.UpdatedAt), C.int64_t(v.Burst)
	default:
		ci.algorithm = 255 // a CacheItem without a usable Value (algorithms.go:55-63)
	}
	return ci
}

func fromCItem(key string, c *C.guber_item_t) *CacheItem {
	item := &CacheItem{Algorithm: Algorithm(c.algorithm), Key: key, ExpireAt: int64(c.expire_at), InvalidAt: int64(c.invalid_at)}
	switch c.algorithm {
	case C.GUBER_ALGO_TOKEN_BUCKET:
		item.Value = &TokenBucketItem{Status: Status(c.status), Limit: int64(c.limit), Duration: int64(c.duration), Remaining: int64(c.remaining), CreatedAt: int64(c.stamp)}
	case C.GUBER_ALGO_LEAKY_BUCKET:
		item.Value = &LeakyBucketItem{Limit: int64(c.limit), Duration: int64(c.duration), Remaining: float64(c.remaining_f), UpdatedAt: int64(c.stamp), Burst: int64(c.burst)}
	}
	return item
}
