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
// Requests from any number of gRPC goroutines are routed to their key's device (the reference's replicated consistent
// hash over the peers gpu0..gpuN-1) and shard (the reference's worker rule).  There the CALLING goroutine does the
// per-request work itself: it reserves a slot (and its key bytes) in the shard's open stage with one compare-and-swap,
// writes its request IN PLACE into the stage — request / response arrays in device-visible host memory owned by the C side
// (cgo: no Go pointer is retained) — and, once the stage's generation has been announced, reads its response out of the
// stage's result arrays.  The shard's batcher goroutine never touches a request: it seals the open stage at BatchLimit
// items or BatchWait after the first one (the policy of peer_client.go:284-337), submits it, opens the next of its three
// stages (one filling, one on the GPU, one being read out) and announces completed generations by closing a channel.
// This file mirrors gubernator_amd/csrc/worker_pool.cpp, which IS compiled and tested in this repository
// (tests/test_gpu_host_layer.py on the GPU, tests/test_pool_cpu.py under ThreadSanitizer), function by function:
// reserve / writeRequest / consume / run / openStage / submit / complete.

//go:build gpu

package gubernator

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/lib -lguber_hip
#include <stdlib.h>
#include <string.h>
#include "guber_gpu.h"
*/
import "C"

import (
	"context"
	"fmt"
	"runtime"
	"sync"
	"sync/atomic"
	"time"
	"unsafe"

	"github.com/mailgun/holster/v4/clock"
	"github.com/pkg/errors"
)

// GPUWorkerPool satisfies the call surface of *WorkerPool.  devices[i] is the HIP ordinal of peer "gpu<i>"; inside a
// device the key space is split over conf.Workers shards by hash range like WorkerPool (workers.go:125-151,180-184).  A
// shard is one engine (HBM table + HIP stream) with its own batcher goroutine, so that batches of different shards
// overlap on the GPU.
type GPUWorkerPool struct {
	conf         *Config
	hasher       workerHasher // workers.go:70-72
	hashRingStep uint64       // workers.go:132
	perDevice    int
	ring         *C.guber_ring_t // nil with one device
	comm         *C.guber_comm_t // GLOBAL exchange between the devices' replicas (nil until EnableGlobalSync)
	shards       []*gpuShard
}

const (
	gpuStages      = 3       // filling / on the GPU / being read out
	gpuStageClosed = 1 << 63 // gpuStage.word: not accepting reservations
	gpuOpenNone    = gpuStages
	gpuOpenDead    = gpuStages + 1
)

// gpuStage = one of a shard's stages and the generation it currently carries (worker_pool.h Stage).
type gpuStage struct {
	stage    *C.guber_stage_t
	b        *C.guber_batch_t
	r        *C.guber_result_t
	reqs     []*RateLimitReq // per slot, written by the slot's owner (Config.Store callbacks need the request)
	word     atomic.Uint64   // gpuStageClosed | key bytes reserved << 32 | slots reserved
	written  atomic.Uint32   // slots filled by their callers
	consumed atomic.Uint32   // responses picked up
	firstNs  atomic.Int64    // when the generation's first reservation was made (BatchWait)
	flushNow atomic.Bool     // a caller found no room: do not wait for BatchWait
	done     chan struct{}   // closed when the generation's responses are ready
	n        int             // sealed size (batcher; callers read it after <-done)
	rc       C.int
	sent     bool      // guber_stage_submit succeeded, guber_stage_wait is due
	t0       time.Time // flush start
	used     bool
}

// gpuShard = one "worker" of the reference: single writer of its own cache.
type gpuShard struct {
	conf   *Config
	engine *C.guber_engine_t
	st     [gpuStages]gpuStage
	open   atomic.Uint32 // index of the stage accepting reservations, gpuOpenNone, or gpuOpenDead after Close
	mu     sync.Mutex
	opened *sync.Cond    // callers in reserve(): a stage opened
	wake   chan struct{} // batcher: first item / full / flush requested
	quit   chan struct{}
	exited chan struct{}
	label  string // metrics label of the shard's device (the reference labels these series by peer address)
	limit  int
	wait   time.Duration
	maxKey int // guber_config_t.max_key_bytes: longer keys are answered per item, never copied
	keyCap int // bytes of a stage's key buffer: a batch whose keys would not fit is flushed early
	// Config.Store side channel (only allocated when conf.Store != nil; that path is synchronous)
	missing, storeFlags *C.uint8_t
	storeItems          *C.guber_item_t
}

func NewGPUWorkerPool(conf *Config, devices []int, batchLimit int, batchWait time.Duration) (*GPUWorkerPool, error) {
	workers := conf.Workers // config.go:110; default = NumCPU, GUBER_GPU_SHARDS overrides it for the GPU pool (4 is enough)
	if workers <= 0 {
		workers = 1
	}
	if len(devices) == 0 {
		devices = []int{0}
	}
	p := &GPUWorkerPool{conf: conf, hasher: &hasher{}, hashRingStep: uint64(1<<63) / uint64(workers), perDevice: workers} // workers.go:80,132
	if len(devices) > 1 { // the GPUs of the node are the peers of the reference's ring (replicated_hash.go:78-119)
		names := make([]*C.char, len(devices))
		for i := range devices {
			names[i] = C.CString(fmt.Sprintf("gpu%d", i))
			defer C.free(unsafe.Pointer(names[i]))
		}
		if rc := C.guber_ring_create(&names[0], C.uint32_t(len(devices)), 512, 0, &p.ring); rc != C.GUBER_OK {
			return nil, fmt.Errorf("guber_ring_create: %s", C.GoString(C.guber_strerror(rc)))
		}
	}
	total := workers * len(devices)
	for _, dev := range devices {
		for i := 0; i < workers; i++ {
			sh, err := newGPUShard(conf, dev, conf.CacheSize/total+1, batchLimit, batchWait) // workers.go:132
			if err != nil {
				_ = p.Close()
				return nil, err
			}
			p.shards = append(p.shards, sh)
		}
	}
	return p, nil
}

// shardOf = device by ReplicatedConsistentHash.Get (replicated_hash.go:104-119), then WorkerPool.getWorker
// (workers.go:180-184) inside the device.
func (p *GPUWorkerPool) shardOf(key string) *gpuShard {
	dev := 0
	if p.ring != nil {
		off := [2]C.uint32_t{0, C.uint32_t(len(key))}
		var owner C.uint32_t
		C.guber_ring_route(p.ring, (*C.uint8_t)(unsafe.Pointer(unsafe.StringData(key))), &off[0], 1, &owner)
		dev = int(owner)
	}
	local := int(p.hasher.ComputeHash63(key) / p.hashRingStep)
	if local >= p.perDevice { // 2^63 is not a multiple of every worker count
		local = p.perDevice - 1
	}
	return p.shards[dev*p.perDevice+local]
}

// EnableGlobalSync prepares the GLOBAL exchange between the devices' replicas (engines must have been created with
// GUBER_FLAG_GLOBAL: conf.Behaviors.ForceGlobal or any GLOBAL traffic).  GlobalSync is then one GlobalSyncWait tick
// (global.go:91-283): hits to their owners over RCCL / xGMI, owners apply and broadcast, every replica installs.
func (p *GPUWorkerPool) EnableGlobalSync() error {
	engines := make([]*C.guber_engine_t, 0, len(p.shards)/p.perDevice)
	for i := 0; i < len(p.shards); i += p.perDevice {
		engines = append(engines, p.shards[i].engine)
	}
	useRccl := C.int(0)
	if len(engines) > 1 {
		useRccl = 1
	}
	if rc := C.guber_comm_create_local(&engines[0], C.uint32_t(len(engines)), p.ring, useRccl, &p.comm); rc != C.GUBER_OK {
		return fmt.Errorf("guber_comm_create_local: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
	}
	return nil
}
func (p *GPUWorkerPool) GlobalSync() error {
	var st C.guber_global_sync_stats_t
	if rc := C.guber_global_sync(p.comm, C.int64_t(MillisecondNow()), &st); rc != C.GUBER_OK {
		return fmt.Errorf("guber_global_sync: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
	}
	metricGlobalSendDuration.Observe(float64(st.ms) / 1000)
	return nil
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
		select { // workers.go:349-360: Load stops when the context is cancelled
		case <-ctx.Done():
			for _, items := range pending {
				for i := range items {
					C.free(unsafe.Pointer(items[i].key))
				}
			}
			return ctx.Err()
		default:
		}
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
	if p.comm != nil {
		C.guber_comm_destroy(p.comm)
		p.comm = nil
	}
	for _, sh := range p.shards {
		_ = sh.Close()
	}
	if p.ring != nil {
		C.guber_ring_destroy(p.ring)
		p.ring = nil
	}
	return nil
}

func newGPUShard(conf *Config, device int, cacheSize int, batchLimit int, batchWait time.Duration) (*gpuShard, error) {
	const maxKey = 1024 // guber_config_t.max_key_bytes default
	cfg := C.guber_config_t{struct_size: C.uint32_t(unsafe.Sizeof(C.guber_config_t{})), device: C.int32_t(device),
		cache_size: C.uint64_t(cacheSize), max_batch: C.uint32_t(batchLimit), max_key_bytes: maxKey, flags: C.GUBER_FLAG_GLOBAL}
	p := &gpuShard{conf: conf, wake: make(chan struct{}, 1), quit: make(chan struct{}), exited: make(chan struct{}),
		label: fmt.Sprintf("gpu%d", device), limit: batchLimit, wait: batchWait, maxKey: maxKey}
	p.opened = sync.NewCond(&p.mu)
	p.open.Store(gpuOpenNone)
	if rc := C.guber_engine_create(&cfg, &p.engine); rc != C.GUBER_OK {
		return nil, fmt.Errorf("guber_engine_create: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
	}
	p.keyCap = batchLimit*96 + maxKey // typical keys; a batch of longer ones is flushed early, never overrun
	for k := range p.st {
		s := &p.st[k]
		s.word.Store(gpuStageClosed)
		if rc := C.guber_stage_create(p.engine, C.uint32_t(batchLimit), C.uint32_t(p.keyCap), &s.stage); rc != C.GUBER_OK {
			return nil, fmt.Errorf("guber_stage_create: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
		}
		s.b, s.r = C.guber_stage_batch(s.stage), C.guber_stage_result(s.stage)
		s.reqs = make([]*RateLimitReq, batchLimit)
	}
	if conf.Store != nil {
		n := C.size_t(batchLimit)
		p.missing, p.storeFlags = (*C.uint8_t)(C.guber_alloc_pinned(n)), (*C.uint8_t)(C.guber_alloc_pinned(n))
		p.storeItems = (*C.guber_item_t)(C.guber_alloc_pinned(n * C.sizeof_guber_item_t))
	}
	go p.run()
	return p, nil
}

// GetRateLimit = reserve a slot, write the request, wait for the generation, read the response (worker_pool.cpp
// GetRateLimitMany for one request; workers.go:261-291 semantics: ctx honoured at both waits).
func (p *gpuShard) GetRateLimit(ctx context.Context, r *RateLimitReq, st RateLimitReqState) (*RateLimitResp, error) {
	klen := len(r.Name) + 1 + len(r.UniqueKey)
	if klen > p.maxKey { // answered here, never reaches the device
		return nil, errors.New(C.GoString(C.guber_item_strerror(C.GUBER_ITEM_E_KEY_TOO_LONG)))
	}
	s, slot, koff, err := p.reserve(ctx, klen)
	if err != nil {
		return nil, err
	}
	p.writeRequest(s, slot, koff, r, st)
	done := s.done
	select {
	case <-done:
	case <-ctx.Done():
		go func() { <-done; s.consumed.Add(1) }() // the slot still counts towards the stage being read out
		return nil, ctx.Err()
	}
	resp, err := p.answer(r, s.rc, *at8(s.r.err, slot), *at8(s.r.status, slot), *at64(s.r.limit, slot), *at64(s.r.remaining, slot), *at64(s.r.reset_time, slot))
	s.consumed.Add(1)
	return resp, err
}

// reserve takes one slot and klen key bytes of the open stage: ONE compare-and-swap on its reservation word.
func (p *gpuShard) reserve(ctx context.Context, klen int) (*gpuStage, int, int, error) {
	for {
		k := p.open.Load()
		if k == gpuOpenDead {
			return nil, 0, 0, errors.New("worker pool is closed")
		}
		if k < gpuStages {
			s := &p.st[k]
			for w := s.word.Load(); w&gpuStageClosed == 0; w = s.word.Load() {
				cnt, kb := int(uint32(w)), int(w>>32)
				if cnt >= p.limit || kb+klen > p.keyCap { // no slot or no key bytes left: flush it now, take the next stage
					if s.flushNow.CompareAndSwap(false, true) {
						p.kick()
					}
					break
				}
				if s.word.CompareAndSwap(w, w+1+uint64(klen)<<32) {
					if cnt == 0 {
						s.firstNs.Store(time.Now().UnixNano())
					}
					if cnt == 0 || cnt+1 >= p.limit {
						p.kick()
					}
					return s, cnt, kb, nil
				}
			}
		}
		if err := ctx.Err(); err != nil {
			return nil, 0, 0, err
		}
		p.mu.Lock() // wait for the batcher to open the next stage (it broadcasts under mu)
		if p.open.Load() == k {
			p.opened.Wait()
		}
		p.mu.Unlock()
	}
}

// orWord = fetch-or (sync/atomic has no Or for Uint64 before Go 1.23)
func orWord(w *atomic.Uint64, bits uint64) uint64 {
	for {
		old := w.Load()
		if w.CompareAndSwap(old, old|bits) {
			return old
		}
	}
}

func (p *gpuShard) kick() {
	select {
	case p.wake <- struct{}{}:
	default:
	}
}

func at64(p *C.int64_t, i int) *C.int64_t { return (*C.int64_t)(unsafe.Add(unsafe.Pointer(p), i*8)) }
func at32(p *C.uint32_t, i int) *C.uint32_t { return (*C.uint32_t)(unsafe.Add(unsafe.Pointer(p), i*4)) }
func at8(p *C.uint8_t, i int) *C.uint8_t   { return (*C.uint8_t)(unsafe.Add(unsafe.Pointer(p), i)) }

// writeRequest writes one request into its slot in place (HashKey = name + "_" + unique_key, client.go:39-41).
func (p *gpuShard) writeRequest(s *gpuStage, i, koff int, r *RateLimitReq, st RateLimitReqState) {
	b := s.b
	*at32(b.key_off, i) = C.uint32_t(koff)
	dst := unsafe.Add(unsafe.Pointer(b.key_bytes), koff)
	C.memcpy(dst, unsafe.Pointer(unsafe.StringData(r.Name)), C.size_t(len(r.Name)))
	*(*byte)(unsafe.Add(dst, len(r.Name))) = '_'
	C.memcpy(unsafe.Add(dst, len(r.Name)+1), unsafe.Pointer(unsafe.StringData(r.UniqueKey)), C.size_t(len(r.UniqueKey)))
	*at64(b.hits, i), *at64(b.limit, i), *at64(b.duration, i) = C.int64_t(r.Hits), C.int64_t(r.Limit), C.int64_t(r.Duration)
	*at64(b.burst, i), *at64(b.created_at, i) = C.int64_t(r.Burst), C.int64_t(*r.CreatedAt)
	alg := r.Algorithm
	if alg < 0 || alg > 1 {
		alg = 255 // workers.go:317: the engine answers GUBER_ITEM_E_INVALID_ALGORITHM
	}
	*at8(b.algorithm, i) = C.uint8_t(alg)
	*at32(b.behavior, i) = C.uint32_t(r.Behavior)
	owner := C.uint8_t(0)
	if st.IsOwner {
		owner = 1
	}
	*at8(b.is_owner, i) = owner
	s.reqs[i] = r
	s.written.Add(1)
}

// openStage lets callers reserve in stage k (worker_pool.cpp open_stage).
func (p *gpuShard) openStage(k int) {
	s := &p.st[k]
	s.n, s.rc, s.sent, s.used = 0, C.GUBER_OK, false, true
	s.written.Store(0)
	s.consumed.Store(0)
	s.firstNs.Store(0)
	s.flushNow.Store(false)
	s.done = make(chan struct{})
	s.word.Store(0)
	p.open.Store(uint32(k))
	p.mu.Lock()
	p.opened.Broadcast()
	p.mu.Unlock()
}

// run is the shard's batcher (worker_pool.cpp GPUWorkerPool::run): it seals the open stage at `limit` requests, when a
// caller found no room in it, or `wait` after its first reservation, submits it and opens the next one; with nothing due
// it delivers the batch in flight.
func (p *gpuShard) run() {
	runtime.LockOSThread() // one OS thread owns the HIP context
	defer close(p.exited)
	cur, inflight := 0, -1
	p.openStage(cur)
	timer := time.NewTimer(time.Hour)
	for {
		s := &p.st[cur]
		due, closing := false, false
		for !due {
			select {
			case <-p.quit:
				closing = true
			default:
			}
			cnt := int(uint32(s.word.Load()))
			if cnt >= p.limit || (cnt > 0 && (closing || s.flushNow.Load())) {
				due = true
				break
			}
			left := time.Hour
			if cnt > 0 {
				if first := s.firstNs.Load(); first != 0 {
					left = p.wait - time.Duration(time.Now().UnixNano()-first)
				} else {
					left = p.wait
				}
				if left <= 0 {
					due = true
					break
				}
			}
			if inflight >= 0 || closing {
				break // nothing due: deliver the batch in flight / finish
			}
			if !timer.Stop() {
				select {
				case <-timer.C:
				default:
				}
			}
			timer.Reset(left)
			select {
			case <-p.wake:
			case <-timer.C:
			case <-p.quit:
			}
		}
		if !due {
			if inflight >= 0 {
				p.complete(&p.st[inflight])
				inflight = -1
				continue
			}
			// closing, nothing reserved, nothing in flight: stop taking reservations; a caller may have slipped one in meanwhile
			p.open.Store(gpuOpenDead)
			w := orWord(&s.word, gpuStageClosed)
			if n := int(uint32(w)); n > 0 {
				p.seal(s, w)
				p.submit(s)
				p.complete(s)
			}
			p.mu.Lock()
			p.opened.Broadcast()
			p.mu.Unlock()
			return
		}
		// the next stage takes the reservations from here on; it was announced two flushes ago and has been read out since
		next := (cur + 1) % gpuStages
		nx := &p.st[next]
		for nx.used && int(nx.consumed.Load()) != nx.n {
			runtime.Gosched()
		}
		p.openStage(next)
		p.seal(s, orWord(&s.word, gpuStageClosed))
		p.submit(s)
		if inflight >= 0 {
			p.complete(&p.st[inflight])
		}
		inflight, cur = cur, next
	}
}

// seal fixes the stage's size and waits for the callers that are still copying their requests in.
func (p *gpuShard) seal(s *gpuStage, w uint64) {
	s.n = int(uint32(w))
	for int(s.written.Load()) != s.n {
		runtime.Gosched()
	}
	*at32(s.b.key_off, s.n) = C.uint32_t(w >> 32)
}

// submit hands a sealed stage to the engine (asynchronous); with a persistent Store configured the batch takes the
// synchronous path that makes the Store's calls.
func (p *gpuShard) submit(s *gpuStage) {
	s.t0 = time.Now()
	s.b.n, s.b.now_ms = C.uint32_t(s.n), C.int64_t(clock.Now().UnixNano()/1000000) // MillisecondNow(); DURATION_IS_GREGORIAN is derived from it on the device
	metricBatchQueueLength.WithLabelValues(p.label).Set(float64(s.n)) // gubernator.go:100-103
	if p.conf.Store != nil {
		s.rc = p.evalWithStore(s.reqs[:s.n], s.b, s.r) // synchronous: Store.Get / OnChange / Remove are made in request order
		return
	}
	s.rc = C.guber_stage_submit(s.stage)
	s.sent = s.rc == C.GUBER_OK
}

// complete waits for a submitted stage and announces its generation: the callers read their responses themselves.
func (p *gpuShard) complete(s *gpuStage) {
	if s.sent {
		s.rc, s.sent = C.guber_stage_wait(s.stage), false
	}
	if s.rc == C.GUBER_OK { // prometheus: the engine returns the per-batch aggregates of the reference's counters
		res := s.r
		metricOverLimitCounter.Add(float64(res.over_limit_count))             // algorithms.go:165,185,243,391,409,471
		metricCacheAccess.WithLabelValues("hit").Add(float64(res.cache_hits)) // lrucache.go:117,121,126
		metricCacheAccess.WithLabelValues("miss").Add(float64(res.cache_misses))
		metricCacheSize.Set(float64(res.cache_size))
		metricCacheUnexpiredEvictions.Add(float64(res.unexpired_evictions)) // lrucache.go:142-146
	}
	metricBatchSendDuration.WithLabelValues(p.label).Observe(time.Since(s.t0).Seconds()) // gubernator.go:104-110
	close(s.done)
}

func (p *gpuShard) answer(r *RateLimitReq, rc C.int, e, status C.uint8_t, limit, remaining, reset C.int64_t) (*RateLimitResp, error) {
	if rc != C.GUBER_OK {
		return nil, errors.Errorf("gpu engine: %s", C.GoString(C.guber_strerror(rc)))
	}
	if e != 0 {
		msg := C.GoString(C.guber_item_strerror(e))
		if e == C.GUBER_ITEM_E_INVALID_ALGORITHM {
			msg = fmt.Sprintf(msg, r.Algorithm) // "Invalid rate limit algorithm '%d'"
		}
		return nil, errors.New(msg)
	}
	return &RateLimitResp{Status: Status(status), Limit: int64(limit), Remaining: int64(remaining), ResetTime: int64(reset)}, nil
}

// evalWithStore is the Config.Store path (store.go:49-65).  The reference calls the store from inside the
// algorithms; here the engine reports which calls are due and this function makes them, in the same order:
//   Store.Get      for the first request of every key that is not resident before the batch (algorithms.go:45-51)
//   Store.Remove   token RESET_REMAINING / algorithm switched                                 (:79-84, :96-100, :311-315)
//   Store.OnChange with the CacheItem as it is right after THAT request, owner only           (:149-153, :252-254, ...)
func (p *gpuShard) evalWithStore(batch []*RateLimitReq, b *C.guber_batch_t, res *C.guber_result_t) C.int {
	ctx := context.Background()
	if rc := C.guber_probe_missing(p.engine, b, p.missing); rc != C.GUBER_OK {
		return rc
	}
	asked := map[string]struct{}{}
	for i, req := range batch {
		if *at8(p.missing, i) == 0 {
			continue
		}
		key := req.HashKey()
		if _, dup := asked[key]; dup {
			continue
		}
		asked[key] = struct{}{}
		if item, ok := p.conf.Store.Get(ctx, req); ok {
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
	for i, req := range batch {
		f := *at8(p.storeFlags, i)
		if f&C.GUBER_STORE_REMOVE != 0 {
			p.conf.Store.Remove(ctx, req.HashKey())
		}
		if f&C.GUBER_STORE_ONCHANGE != 0 {
			ci := (*C.guber_item_t)(unsafe.Add(unsafe.Pointer(p.storeItems), i*C.sizeof_guber_item_t))
			p.conf.Store.OnChange(ctx, req, fromCItem(req.HashKey(), ci))
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

// Close (workers.go:157): the batcher evaluates and announces what has been reserved, later callers are refused, and the
// stages go away once their responses have been read out.
func (p *gpuShard) Close() error {
	close(p.quit)
	<-p.exited
	for k := range p.st {
		s := &p.st[k]
		for s.used && int(s.consumed.Load()) != s.n {
			runtime.Gosched()
		}
		C.guber_stage_destroy(s.stage)
	}
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
