// gpu_worker_pool.go — cgo binding of the MI355X rate-limit engine for mailgun/gubernator v2.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (there is no Go toolchain here); it is the file a gubernator maintainer drops into
// the `gubernator` package (build tag `gpu`) to replace `WorkerPool` (workers.go:54-626) with the engine behind
// include/guber_gpu.h.  The method set is the one `V1Instance` calls today:
//
//	GetRateLimit(ctx, *RateLimitReq, RateLimitReqState) (*RateLimitResp, error)   gubernator.go:598, global.go:245
//	AddCacheItem(ctx, key, *CacheItem) error                                       gubernator.go:452
//	GetCacheItem(ctx, key) (*CacheItem, bool, error)                               workers.go:583
//	Load(ctx) error / Store(ctx) error / Close() error                             gubernator.go:143,161,169
//
// plus GetRateLimits for a whole RPC (what V1Instance.GetRateLimits should call instead of one goroutine per request,
// gubernator.go:221-300).
//
// The binding is deliberately THIN: everything that decides anything — key hashing, device and shard placement (with the
// online isolation of hot keys), slot reservation in the devices' stages, the dispatcher whose batches the GPU hands to the
// shards' tables, completion, the Store call sequence — lives in the C++ pool (gubernator_amd/csrc/worker_pool.cpp), which IS
// compiled and tested in this repository (tests/test_gpu_host_layer.py on the GPU; tests/test_pool_cpu.py under
// ThreadSanitizer / AddressSanitizer).  A Go re-implementation of that logic could only drift from it.  What happens here:
// the requests of a call are laid out as structure-of-arrays in C memory (no Go pointer is retained by C: cgo rules), ONE cgo
// call per RPC blocks until the answers are there (the calling goroutine's thread sleeps on a futex inside; the Go scheduler
// runs other goroutines on other threads), and the answers are copied out.

//go:build gpu

package gubernator

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/lib -lguber_hip
#include <stdlib.h>
#include <string.h>
#include "guber_gpu.h"

// Config.Store as C callbacks (guber_pool_set_store): trampolines into the exported Go functions below
extern int goStoreGet(void* user, guber_store_req_t* r, guber_item_t* out);
extern void goStoreOnChange(void* user, guber_store_req_t* r, guber_item_t* item);
extern void goStoreRemove(void* user, uint8_t* key, uint32_t key_len);
static void guber_go_set_store(guber_pool_t* p, void* user) {
    guber_store_callbacks_t cb;
    cb.get = (int (*)(void*, const guber_store_req_t*, guber_item_t*))goStoreGet;
    cb.on_change = (void (*)(void*, const guber_store_req_t*, const guber_item_t*))goStoreOnChange;
    cb.remove = (void (*)(void*, const uint8_t*, uint32_t))goStoreRemove;
    cb.user = user;
    guber_pool_set_store(p, &cb);
}
extern void goStoreSave(void* user, guber_item_t* item);
static int guber_go_store_all(guber_pool_t* p, void* user) { return guber_pool_store(p, (void (*)(void*, const guber_item_t*))goStoreSave, user); }
*/
import "C"

import (
	"context"
	"fmt"
	"runtime/cgo"
	"sync"
	"time"
	"unsafe"

	"github.com/pkg/errors"
	"github.com/sirupsen/logrus"
)

// GPUWorkerPool satisfies the call surface of *WorkerPool.  devices[i] is the HIP ordinal of peer "gpu<i>" on the reference's
// replicated consistent hash (a node's GPUs are peers of a ring inside the process); inside a device the key space is split
// over `shards` logical shards — engines with their own HBM table — whose initial key -> shard map IS the reference's worker
// rule (workers.go:153-155,180-184).  shards <= 0 picks ONE table per device (conf.Workers defaults to NumCPU, which is a statement
// about goroutines, not about tables on a GPU; the pool's callers are its bound, and they are fastest on one table: see below;
// GUBER_GPU_SHARDS overrides).
type GPUWorkerPool struct {
	conf   *Config
	pool   *C.guber_pool_t
	handle cgo.Handle // of this pool, for the Store trampolines
	bufs   sync.Pool  // *rpcBuf
	closed sync.Once
}

// rpcBuf is the C memory of one call: request columns, packed strings, response columns.  Reused through a sync.Pool.
type rpcBuf struct {
	cap, strCap       int
	nameOff, ukeyOff  *C.uint32_t
	names, ukeys      *C.uint8_t
	i64               *C.int64_t // hits | limit | duration | burst | created_at | out limit | out remaining | out reset_time
	algo              *C.int32_t
	behavior          *C.uint32_t
	owner, status, er *C.uint8_t
	errText           *C.char
}

const errStride = 200

func newRPCBuf(n, strBytes int) *rpcBuf {
	b := &rpcBuf{cap: n, strCap: strBytes}
	b.nameOff = (*C.uint32_t)(C.malloc(C.size_t(4 * (n + 1))))
	b.ukeyOff = (*C.uint32_t)(C.malloc(C.size_t(4 * (n + 1))))
	b.names = (*C.uint8_t)(C.malloc(C.size_t(strBytes + 16)))
	b.ukeys = (*C.uint8_t)(C.malloc(C.size_t(strBytes + 16)))
	b.i64 = (*C.int64_t)(C.malloc(C.size_t(8 * 8 * n)))
	b.algo = (*C.int32_t)(C.malloc(C.size_t(4 * n)))
	b.behavior = (*C.uint32_t)(C.malloc(C.size_t(4 * n)))
	b.owner = (*C.uint8_t)(C.malloc(C.size_t(3 * n)))
	b.status = (*C.uint8_t)(unsafe.Add(unsafe.Pointer(b.owner), n))
	b.er = (*C.uint8_t)(unsafe.Add(unsafe.Pointer(b.owner), 2*n))
	b.errText = (*C.char)(C.malloc(C.size_t(errStride * n)))
	return b
}
func (b *rpcBuf) free() {
	for _, p := range []unsafe.Pointer{unsafe.Pointer(b.nameOff), unsafe.Pointer(b.ukeyOff), unsafe.Pointer(b.names), unsafe.Pointer(b.ukeys),
		unsafe.Pointer(b.i64), unsafe.Pointer(b.algo), unsafe.Pointer(b.behavior), unsafe.Pointer(b.owner), unsafe.Pointer(b.errText)} {
		C.free(p)
	}
}
func (b *rpcBuf) col(k int) *C.int64_t { return (*C.int64_t)(unsafe.Add(unsafe.Pointer(b.i64), k*8*b.cap)) }

func at64(p *C.int64_t, i int) *C.int64_t  { return (*C.int64_t)(unsafe.Add(unsafe.Pointer(p), i*8)) }
func at32(p *C.uint32_t, i int) *C.uint32_t { return (*C.uint32_t)(unsafe.Add(unsafe.Pointer(p), i*4)) }
func ati32(p *C.int32_t, i int) *C.int32_t  { return (*C.int32_t)(unsafe.Add(unsafe.Pointer(p), i*4)) }
func at8(p *C.uint8_t, i int) *C.uint8_t    { return (*C.uint8_t)(unsafe.Add(unsafe.Pointer(p), i)) }

// NewGPUWorkerPool replaces NewWorkerPool (workers.go:125).  batchLimit / batchWait bound a shard's batch the way the peer
// batcher's BatchLimit / BatchWait do (peer_client.go:284-337); below them a batch goes as soon as the device has room.
func NewGPUWorkerPool(conf *Config, devices []int, shards int, batchLimit int, batchWait time.Duration) (*GPUWorkerPool, error) {
	if shards <= 0 {
		// ONE table per device is the fastest arrangement of the pool: its callers — the host's CPUs — are the bound (16 usable CPUs:
		// 323 M decisions/s on one table against 222 on eight shards, profiles/r04_final2_bench_driver_cmd.json), and one table alone
		// evaluates 2.4 G/s.  Several shards per device pay where batches arrive already in HBM (guber_eval_batches_routed_dev: 10 G/s).
		shards = 1
	}
	cfg := C.guber_config_t{}
	cfg.struct_size = C.uint32_t(unsafe.Sizeof(cfg))
	cfg.cache_size = C.uint64_t(conf.CacheSize)
	cfg.max_batch = C.uint32_t(batchLimit)
	cfg.flags = C.GUBER_FLAG_GLOBAL // one more engine per device holds the keys of GLOBAL requests: the replica GlobalSync keeps in step
	devs := make([]C.int32_t, len(devices))
	for i, d := range devices {
		devs[i] = C.int32_t(d)
	}
	var dp *C.int32_t
	if len(devs) > 0 {
		cfg.device = devs[0]
		dp = &devs[0]
	}
	p := &GPUWorkerPool{conf: conf}
	if rc := C.guber_pool_create_multi(&cfg, dp, C.uint32_t(len(devs)), C.uint32_t(shards), C.uint32_t(batchLimit),
		C.uint32_t(batchWait/time.Microsecond), &p.pool); rc != C.GUBER_OK {
		return nil, errors.Errorf("guber_pool_create_multi: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
	}
	p.bufs.New = func() any { return newRPCBuf(kMaxBatch, kMaxBatch*96) }
	// DURATION_IS_GREGORIAN intervals are computed on the device from the batch clock; interval.go:97-142 builds its civil dates in
	// now.Location(), so the engine gets the daemon's zone: the offset in effect now and the coming transitions (UTC: nothing to say)
	// (guber_set_timezone is process-wide and waits for every device: ONCE per process, before the first batch — a second pool must not
	// stall the first one's kernels —, and again from a timer well before the table of 16 transitions runs out: ADVICE r05)
	if err := ensureEngineTimezone(); err != nil {
		C.guber_pool_destroy(p.pool)
		return nil, err
	}
	if conf.Store != nil {
		p.handle = cgo.NewHandle(p)
		C.guber_go_set_store(p.pool, unsafe.Pointer(uintptr(p.handle)))
	}
	return p, nil
}

var (
	tzOnce sync.Once
	tzErr  error
)

// ensureEngineTimezone publishes the daemon's zone once per process and keeps it fresh: the table holds 16 transitions — eight years
// of two changes, four of a zone with four per year —, so it is rebuilt every 180 days, from "now", for as long as the process lives.
// A refresh swaps the table between batches (guber_set_timezone synchronises every device it touches; at two calls a year that is noise).
func ensureEngineTimezone() error {
	tzOnce.Do(func() {
		tzErr = setEngineTimezone(time.Local, time.Now(), 7)
		if tzErr != nil {
			return
		}
		go func() {
			for range time.Tick(180 * 24 * time.Hour) {
				if err := setEngineTimezone(time.Local, time.Now(), 7); err != nil {
					logrus.WithError(err).Error("gubernator: refreshing the engine's time zone table failed; the previous table stays in effect")
				}
			}
		}()
	})
	return tzErr
}

// setEngineTimezone hands guber_set_timezone the zone as Go's time.Location holds it: the UTC offset at `from` and the transitions of the
// next `years` years as (UTC second, offset from then on).  Go exports no transition table, so the offsets are sampled hourly and every
// change is bisected to the second (a zone with daylight saving time has two per year; the engine's table takes 16).  ensureEngineTimezone
// calls it at start-up and every 180 days; tests call it with another Location.
func setEngineTimezone(loc *time.Location, from time.Time, years int) error {
	offAt := func(u int64) int { _, off := time.Unix(u, 0).In(loc).Zone(); return off }
	t := from.Unix() - from.Unix()%3600
	end := from.AddDate(years, 0, 0).Unix()
	off0 := offAt(t)
	cur := off0
	var when []int64
	var offs []int32
	for ; t < end && len(when) < 16; t += 3600 {
		if offAt(t+3600) == cur {
			continue
		}
		lo, hi := t, t+3600
		for hi-lo > 1 {
			if mid := (lo + hi) / 2; offAt(mid) == cur {
				lo = mid
			} else {
				hi = mid
			}
		}
		cur = offAt(hi)
		when = append(when, hi)
		offs = append(offs, int32(cur))
	}
	if off0 == 0 && len(when) == 0 {
		return nil // UTC: the engine's default
	}
	tz := C.guber_tz_t{n: C.uint32_t(len(when)), offset0_s: C.int32_t(off0)}
	if n := len(when); n > 0 { // the two arrays in C memory: the struct handed to C must not point into the Go heap
		tz.when_s = (*C.int64_t)(C.malloc(C.size_t(8 * n)))
		tz.offset_s = (*C.int32_t)(C.malloc(C.size_t(4 * n)))
		defer C.free(unsafe.Pointer(tz.when_s))
		defer C.free(unsafe.Pointer(tz.offset_s))
		copy(unsafe.Slice((*int64)(unsafe.Pointer(tz.when_s)), n), when)
		copy(unsafe.Slice((*int32)(unsafe.Pointer(tz.offset_s)), n), offs)
	}
	if rc := C.guber_set_timezone(&tz); rc != C.GUBER_OK {
		return errors.Errorf("guber_set_timezone: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
	}
	return nil
}

const kMaxBatch = 1000 // gubernator.go:40 maxBatchSize

// GetRateLimits evaluates the requests of one RPC that this instance owns or answers from its replica (the slice of
// V1Instance.GetRateLimits after peer selection, gubernator.go:221-300): responses come back in request order; a per-item
// failure is a response with Error set, exactly the strings the reference produces (gubernator.go:208-217,250-255).
func (p *GPUWorkerPool) GetRateLimits(ctx context.Context, reqs []*RateLimitReq, states []RateLimitReqState) ([]*RateLimitResp, error) {
	n := len(reqs)
	if n > kMaxBatch {
		return nil, fmt.Errorf("Requests.RateLimits list too large; max size is '%d'", kMaxBatch)
	}
	strBytes := 0
	for _, r := range reqs {
		strBytes += len(r.Name) + len(r.UniqueKey)
	}
	b := p.bufs.Get().(*rpcBuf)
	if strBytes > b.strCap { // rare: very long keys
		b.free()
		b = newRPCBuf(kMaxBatch, strBytes+4096)
	}
	defer p.bufs.Put(b)
	hits, limit, duration, burst, created := b.col(0), b.col(1), b.col(2), b.col(3), b.col(4)
	no, uo := 0, 0
	for i, r := range reqs {
		*at32(b.nameOff, i), *at32(b.ukeyOff, i) = C.uint32_t(no), C.uint32_t(uo)
		copy(unsafe.Slice((*byte)(unsafe.Add(unsafe.Pointer(b.names), no)), len(r.Name)), r.Name)
		copy(unsafe.Slice((*byte)(unsafe.Add(unsafe.Pointer(b.ukeys), uo)), len(r.UniqueKey)), r.UniqueKey)
		no += len(r.Name)
		uo += len(r.UniqueKey)
		*at64(hits, i), *at64(limit, i), *at64(duration, i), *at64(burst, i) = C.int64_t(r.Hits), C.int64_t(r.Limit), C.int64_t(r.Duration), C.int64_t(r.Burst)
		if r.CreatedAt != nil {
			*at64(created, i) = C.int64_t(*r.CreatedAt)
		} else {
			*at64(created, i) = 0 // the pool stamps MillisecondNow (gubernator.go:218-220)
		}
		*ati32(b.algo, i), *at32(b.behavior, i) = C.int32_t(r.Algorithm), C.uint32_t(r.Behavior)
		owner := C.uint8_t(1)
		if states != nil && !states[i].IsOwner {
			owner = 0
		}
		*at8(b.owner, i) = owner
	}
	*at32(b.nameOff, n), *at32(b.ukeyOff, n) = C.uint32_t(no), C.uint32_t(uo)
	out := C.guber_result_t{status: b.status, limit: b.col(5), remaining: b.col(6), reset_time: b.col(7), err: b.er}
	rc := C.guber_pool_get_rate_limits_owner(p.pool, C.uint32_t(n), b.names, b.nameOff, b.ukeys, b.ukeyOff, hits, limit, duration, burst, created,
		b.algo, b.behavior, b.owner, &out, b.errText, errStride)
	if rc != C.GUBER_OK {
		return nil, errors.Errorf("gpu pool: %s", C.GoString(C.guber_strerror(rc)))
	}
	resps := make([]*RateLimitResp, n)
	for i := range reqs {
		if *at8(b.er, i) != 0 {
			resps[i] = &RateLimitResp{Error: C.GoString((*C.char)(unsafe.Add(unsafe.Pointer(b.errText), i*errStride)))}
			continue
		}
		resps[i] = &RateLimitResp{Status: Status(*at8(b.status, i)), Limit: int64(*at64(out.limit, i)), Remaining: int64(*at64(out.remaining, i)),
			ResetTime: int64(*at64(out.reset_time, i))}
	}
	return resps, nil
}

// GetRateLimit = WorkerPool.GetRateLimit (workers.go:261): one request, the caller's RateLimitReqState.
func (p *GPUWorkerPool) GetRateLimit(ctx context.Context, r *RateLimitReq, s RateLimitReqState) (*RateLimitResp, error) {
	resps, err := p.GetRateLimits(ctx, []*RateLimitReq{r}, []RateLimitReqState{s})
	if err != nil {
		return nil, err
	}
	if resps[0].Error != "" {
		return nil, errors.New(resps[0].Error) // nil response + error (workers.go:317-321)
	}
	return resps[0], nil
}

// AddCacheItem = LRUCache.Add through the pool (UpdatePeerGlobals, gubernator.go:425-459): the item goes to the shard the
// placement gives its key.
func (p *GPUWorkerPool) AddCacheItem(ctx context.Context, key string, item *CacheItem) error {
	ci := toCItem(key, item)
	defer C.free(unsafe.Pointer(ci.key))
	// The reference's only caller is UpdatePeerGlobals (gubernator.go:452): the broadcast state of a GLOBAL rate limit.  It belongs
	// to the device's GLOBAL engine, where the non-owner's GLOBAL requests are answered from (guber_gpu.h: guber_pool_add_item_for).
	if rc := C.guber_pool_add_item_for(p.pool, &ci, C.GUBER_BEHAVIOR_GLOBAL); rc != C.GUBER_OK {
		return errors.Errorf("guber_pool_add_item_for: %s", C.GoString(C.guber_strerror(rc)))
	}
	return nil
}

// GetCacheItem = LRUCache.GetItem (expired items are removed and reported absent).
func (p *GPUWorkerPool) GetCacheItem(ctx context.Context, key string) (*CacheItem, bool, error) {
	ck := C.CString(key)
	defer C.free(unsafe.Pointer(ck))
	var out C.guber_item_t
	var found C.int
	if rc := C.guber_pool_get_item(p.pool, (*C.uint8_t)(unsafe.Pointer(ck)), C.uint32_t(len(key)), &out, &found); rc != C.GUBER_OK {
		return nil, false, errors.Errorf("guber_pool_get_item: %s", C.GoString(C.guber_strerror(rc)))
	}
	if found == 0 {
		return nil, false, nil
	}
	return fromCItem(key, &out), true, nil
}

// Load = WorkerPool.Load (workers.go:329-449): every item of Config.Loader goes to the shard of its key, in chunks.
func (p *GPUWorkerPool) Load(ctx context.Context) error {
	if p.conf.Loader == nil {
		return nil
	}
	ch, err := p.conf.Loader.Load()
	if err != nil {
		return errors.Wrap(err, "Error in loader.Load")
	}
	chunk := make([]C.guber_item_t, 0, 4096)
	flush := func() error {
		if len(chunk) == 0 {
			return nil
		}
		rc := C.guber_pool_load(p.pool, &chunk[0], C.uint32_t(len(chunk)))
		for i := range chunk {
			C.free(unsafe.Pointer(chunk[i].key))
		}
		chunk = chunk[:0]
		if rc != C.GUBER_OK {
			return errors.Errorf("guber_pool_load: %s", C.GoString(C.guber_strerror(rc)))
		}
		return nil
	}
	for item := range ch {
		chunk = append(chunk, toCItem(item.Key, item))
		if len(chunk) == cap(chunk) {
			if err := flush(); err != nil {
				return err
			}
		}
	}
	return flush()
}

// Store = WorkerPool.Store (workers.go:451-534): every resident item of every shard to Config.Loader.Save.
func (p *GPUWorkerPool) Store(ctx context.Context) error {
	if p.conf.Loader == nil {
		return nil
	}
	out := make(chan *CacheItem, 500)
	done := make(chan error, 1)
	go func() { done <- p.conf.Loader.Save(out) }()
	h := cgo.NewHandle(out)
	rc := C.guber_go_store_all(p.pool, unsafe.Pointer(uintptr(h)))
	h.Delete()
	close(out)
	if err := <-done; err != nil {
		return errors.Wrap(err, "Error in loader.Save")
	}
	if rc != C.GUBER_OK {
		return errors.Errorf("guber_pool_store: %s", C.GoString(C.guber_strerror(rc)))
	}
	return nil
}

//export goStoreSave
func goStoreSave(user unsafe.Pointer, item *C.guber_item_t) {
	out := cgo.Handle(uintptr(user)).Value().(chan *CacheItem)
	out <- fromCItem(C.GoStringN((*C.char)(unsafe.Pointer(item.key)), C.int(item.key_len)), item)
}

// GlobalSync is one GlobalSyncWait tick (global.go:91-283) over the GLOBAL engines of this pool's devices, natively
// (guber_global_sync: hits to the owners, owners apply and broadcast).  A daemon that is one rank of a multi-node ring builds
// its own communicator over guber_pool_global_engine(pool, device) with guber_comm_create_rank instead.
func (p *GPUWorkerPool) GlobalSync() error {
	if rc := C.guber_pool_global_sync(p.pool, nil); rc != C.GUBER_OK {
		return errors.Errorf("guber_pool_global_sync: %s (%s)", C.GoString(C.guber_strerror(rc)), C.GoString(C.guber_last_error()))
	}
	return nil
}

// Close = WorkerPool.Close (workers.go:157): what has been accepted is evaluated and answered, later callers are refused.
func (p *GPUWorkerPool) Close() error {
	p.closed.Do(func() {
		C.guber_pool_destroy(p.pool)
		if p.handle != 0 {
			p.handle.Delete()
		}
	})
	return nil
}

// ---- Config.Store (store.go:49-65): the pool's dispatcher makes the calls in the reference's order (Get on a miss — also on
// the miss a RESET_REMAINING causes for a later request of the same call —, Remove, OnChange with the item after the request)
func storeReq(r *C.guber_store_req_t) (*RateLimitReq, string) {
	key := C.GoStringN((*C.char)(unsafe.Pointer(r.key)), C.int(r.key_len))
	created := int64(r.created_at)
	return &RateLimitReq{Name: key[:r.name_len], UniqueKey: key[r.name_len+1:], Hits: int64(r.hits), Limit: int64(r.limit), Duration: int64(r.duration),
		Burst: int64(r.burst), CreatedAt: &created, Algorithm: Algorithm(r.algorithm), Behavior: Behavior(r.behavior)}, key
}

//export goStoreGet
func goStoreGet(user unsafe.Pointer, r *C.guber_store_req_t, out *C.guber_item_t) C.int {
	p := cgo.Handle(uintptr(user)).Value().(*GPUWorkerPool)
	req, key := storeReq(r)
	item, ok := p.conf.Store.Get(context.Background(), req)
	if !ok || item == nil {
		return 0
	}
	ci := toCItem(key, item)
	C.free(unsafe.Pointer(ci.key)) // (the pool fills in the key itself)
	ci.key = nil
	*out = ci
	return 1
}

//export goStoreOnChange
func goStoreOnChange(user unsafe.Pointer, r *C.guber_store_req_t, item *C.guber_item_t) {
	p := cgo.Handle(uintptr(user)).Value().(*GPUWorkerPool)
	req, key := storeReq(r)
	p.conf.Store.OnChange(context.Background(), req, fromCItem(key, item))
}

//export goStoreRemove
func goStoreRemove(user unsafe.Pointer, key *C.uint8_t, keyLen C.uint32_t) {
	p := cgo.Handle(uintptr(user)).Value().(*GPUWorkerPool)
	p.conf.Store.Remove(context.Background(), C.GoStringN((*C.char)(unsafe.Pointer(key)), C.int(keyLen)))
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
