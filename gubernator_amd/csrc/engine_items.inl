// engine_items.inl — part of guber_engine.hip's translation unit (included there, in this order; not a header of its own):
// guber_eval_batch and the cache operations: items in and out, dump, moves between tables, the ring router, GLOBAL rows.
extern "C" int guber_eval_batch(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r) { return eval_batch_host(e, b, r, nullptr); }
extern "C" int guber_eval_batch_store(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r, guber_store_events_t* ev) {
    if (!ev) return fail(GUBER_E_INVALID_ARG, "null store events");
    return eval_batch_host(e, b, r, ev);
}

// Store.Get is due for a request whose key is not resident when the request is applied (algorithms.go:45-51,
// :274-280): report the keys that are absent or expired at now_ms BEFORE the batch, so that the host can ask
// the Store and hand what it finds to guber_add_items first.
extern "C" int guber_probe_missing(guber_engine_t* e, const guber_batch_t* b, uint8_t* missing) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    if (!b || (b->n && (!b->key_bytes || !b->key_off || !missing))) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (b->n == 0) return GUBER_OK;
    if (b->n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const uint32_t n = b->n;
    const size_t kbytes = b->key_off[n] - b->key_off[0];
    if (e->d_keys.ensure(kbytes + 16) || e->d_off.ensure(n + 1) || e->d_out8.ensure((size_t)n * 2) || e->h_stage.ensure(kbytes + 16 + (size_t)(n + 1) * 4 + n + 64))
        return GUBER_E_NOMEM;
    uint32_t* soff = (uint32_t*)e->h_stage.p;
    uint8_t* skeys = e->h_stage.p + (size_t)(n + 1) * 4;
    uint8_t* sout = skeys + ((kbytes + 16 + 7) & ~(size_t)7);
    for (uint32_t i = 0; i <= n; ++i) soff[i] = b->key_off[i] - b->key_off[0];
    memcpy(skeys, b->key_bytes + b->key_off[0], kbytes);
    memset(skeys + kbytes, 0, 16);
    hipStream_t st = e->stream;
    HIPCHK(hipMemcpyAsync(e->d_keys.p, skeys, kbytes + 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->d_off.p, soff, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_probe_missing, dim3((n + 255) / 256), dim3(256), 0, st, e->T, e->d_keys.p, e->d_off.p, n, b->now_ms, e->d_out8.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sout, e->d_out8.p, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(missing, sout, n);
    return GUBER_OK;
}

// ---------------------------------------------------------------------------------------------
static Rec rec_from_item(const guber_item_t& in) {
    Rec s; rec_clear(s);
    s.limit = in.limit; s.duration = in.duration; s.stamp = in.stamp; s.burst = in.burst;
    s.expire_at = in.expire_at; s.invalid_at = in.invalid_at;
    if (in.algorithm == GUBER_ALGO_TOKEN_BUCKET) { s.remaining = in.remaining; s.burst = 0; s.meta = make_meta(K_TOKEN, in.status, ALGO_TOKEN); }
    else if (in.algorithm == GUBER_ALGO_LEAKY_BUCKET) { s.remaining = f2bits(in.remaining_f); s.meta = make_meta(K_LEAKY, 0, ALGO_LEAKY); }
    else s.meta = make_meta(K_NIL, 0, in.algorithm);   // gubernator.go:435-455: no Value for other algorithms
    return s;
}
static void item_from_rec(const Rec& s, guber_item_t* out) {
    memset(out, 0, sizeof(*out));
    out->limit = s.limit; out->duration = s.duration; out->stamp = s.stamp; out->burst = s.burst;
    out->expire_at = s.expire_at; out->invalid_at = s.invalid_at;
    if (rec_kind(s) == K_TOKEN) { out->algorithm = GUBER_ALGO_TOKEN_BUCKET; out->status = (uint8_t)rec_status(s); out->remaining = s.remaining; out->burst = 0; }
    else if (rec_kind(s) == K_LEAKY) { out->algorithm = GUBER_ALGO_LEAKY_BUCKET; out->remaining_f = bits2f(s.remaining); }
    else {   // CacheItem without a Value: only the CacheItem fields exist
        out->algorithm = (uint8_t)rec_algo(s);
        out->limit = out->duration = out->stamp = out->burst = 0;
    }
}

static int add_items_once(guber_engine* e, const guber_item_t* items, const std::vector<uint32_t>& sel, uint8_t* res_out, uint64_t stamp0) {
    const uint32_t n = (uint32_t)sel.size();
    size_t kbytes = 0;
    for (uint32_t j : sel) kbytes += items[j].key_len;
    std::vector<ItemIn> host(n);
    std::vector<uint8_t> keys(kbytes + 16, 0);
    size_t off = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const guber_item_t& it = items[sel[j]];
        host[j].rec = rec_from_item(it);
        rec_set_stamp(host[j].rec, stamp0 + sel[j]);             // the item's place in the CALL (lrucache.go:91,96: Add moves to the front, item by item)
        host[j].key_off = (uint32_t)off; host[j].key_len = it.key_len;
        if (it.key_len) memcpy(keys.data() + off, it.key, it.key_len);
        off += it.key_len;
    }
    // engine-owned scratch (grown on demand, kept): no allocation on the AddCacheItem / UpdatePeerGlobals path
    DevBuf<ItemIn>& d_items = e->d_items; DevBuf<uint8_t>&d_keys = e->d_ikeys, &d_flags = e->d_iflags, &d_res = e->d_ires;
    DevBuf<uint32_t>& d_slots = e->d_islots;
    int rc = 0;
    rc |= d_items.ensure(n); rc |= d_keys.ensure(keys.size()); rc |= d_flags.ensure(n); rc |= d_res.ensure(n); rc |= d_slots.ensure(n);
    auto cleanup = [&]() {};
    if (rc) return GUBER_E_NOMEM;
    hipStream_t st = e->stream;
    hipError_t he;
    if ((he = hipMemcpyAsync(d_items.p, host.data(), n * sizeof(ItemIn), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (he = hipMemcpyAsync(d_keys.p, keys.data(), keys.size(), hipMemcpyHostToDevice, st)) != hipSuccess) {
        cleanup(); return fail(GUBER_E_HIP, "add_items H2D", he);
    }
    hipLaunchKernelGGL(k_items_probe, dim3((n + 255) / 256), dim3(256), 0, st, e->T, d_items.p, d_keys.p, n, d_slots.p, d_flags.p);
    hipLaunchKernelGGL(k_items_commit, dim3((n + 255) / 256), dim3(256), 0, st, e->T, d_items.p, d_keys.p, n, d_slots.p, d_flags.p, d_res.p, ITEMS_KEEP_STAMP);
    std::vector<uint8_t> res(n);
    if ((he = hipMemcpyAsync(res.data(), d_res.p, n, hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipStreamSynchronize(st)) != hipSuccess) {
        cleanup(); return fail(GUBER_E_HIP, "add_items D2H", he);
    }
    cleanup();
    for (uint32_t j = 0; j < n; ++j) res_out[sel[j]] = res[j];
    return 0;
}

static int add_items_locked(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed);
extern "C" int guber_add_items(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed) {
    if (!e || (!items && n)) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n == 0) return GUBER_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    return add_items_locked(e, items, n, existed);
}
static int add_items_locked(guber_engine_t* e, const guber_item_t* items, uint32_t n, uint8_t* existed) {
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    for (uint32_t i = 0; i < n; ++i) {
        if (!items[i].key || items[i].key_len == 0) return fail(GUBER_E_INVALID_ARG, "item without a key");
        if (items[i].key_len > e->max_key) return fail(GUBER_E_KEY_TOO_LONG, "item key longer than max_key_bytes");
    }
    {
        const int rc = maintain(e, n, e->clock_ms);
        if (rc) return rc;
    }
    note_enqueued(e, n);
    // LRUCache.Add is applied item by item (workers.go:566-581): with duplicates of a key in one call
    // the LAST one must win and the later ones report existed = 1.  Waves of distinct keys keep that; every item carries the
    // recency number of its place in the call, so the order among the call's keys is the reference's too (round 4 numbered the
    // items wave by wave: [A, A, D] left A in front of D).
    const uint64_t stamp0 = take_stamps(e, n);
    std::vector<uint8_t> res(n, 0);
    std::vector<uint32_t> pending(n);
    for (uint32_t i = 0; i < n; ++i) pending[i] = i;
    int guard = 0;
    while (!pending.empty()) {
        if (++guard > 64) return fail(GUBER_E_HIP, "add_items did not converge");
        std::unordered_map<std::string, int> seen;
        std::vector<uint32_t> wave, later;
        for (uint32_t i : pending) {
            std::string k((const char*)items[i].key, items[i].key_len);
            if (seen.emplace(std::move(k), 1).second) wave.push_back(i); else later.push_back(i);
        }
        int rc = add_items_once(e, items, wave, res.data(), stamp0);
        if (rc) return rc;
        std::vector<uint32_t> next;
        for (uint32_t i : wave) {
            if (res[i] == 0xFF) next.push_back(i);            // in-call hash collision: resubmit
            else if (res[i] == 0xFE) return fail(GUBER_E_TABLE_FULL, "no directory entry for item");
        }
        // keep original relative order for the next wave
        next.insert(next.end(), later.begin(), later.end());
        std::sort(next.begin(), next.end());
        pending.swap(next);
    }
    if (existed) for (uint32_t i = 0; i < n; ++i) existed[i] = res[i];
    return maintain(e, 0, e->clock_ms);       // Add evicts as soon as the cache is over its size (lrucache.go:98-100)
}

static int item_lookup(guber_engine* e, const uint8_t* key, uint32_t key_len, int64_t now_ms, int mode, guber_item_t* out, int* found) {
    if (!e || !key || !found) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    *found = 0;
    if (key_len == 0 || key_len > e->max_key) return GUBER_OK;
    DevBuf<uint8_t>& d_key = e->d_lkey; DevBuf<Rec>& d_rec = e->d_lrec; DevBuf<int>& d_found = e->d_lfound;   // engine-owned scratch
    int rc = d_key.ensure(key_len + 16) | d_rec.ensure(1) | d_found.ensure(1);
    auto cleanup = [&]() {};
    if (rc) return GUBER_E_NOMEM;
    std::vector<uint8_t> kb(key_len + 16, 0);
    memcpy(kb.data(), key, key_len);
    Rec hrec; int hfound = 0;
    hipStream_t st = e->stream;
    hipError_t he;
    if ((he = hipMemcpyAsync(d_key.p, kb.data(), kb.size(), hipMemcpyHostToDevice, st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "lookup H2D", he); }
    if (mode == 0 && now_ms > e->clock_ms) e->clock_ms = now_ms;
    hipLaunchKernelGGL(k_item_lookup, dim3(1), dim3(64), 0, st, e->T, d_key.p, key_len, now_ms, mode, d_rec.p, d_found.p, take_stamps(e, 1));
    if ((he = hipMemcpyAsync(&hfound, d_found.p, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipMemcpyAsync(&hrec, d_rec.p, sizeof(Rec), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipStreamSynchronize(st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "lookup D2H", he); }
    cleanup();
    *found = hfound;
    if (hfound && out) { item_from_rec(hrec, out); out->key = nullptr; out->key_len = key_len; }
    return GUBER_OK;
}

// A hot key changes its logical shard (GPUWorkerPool's placement): its bucket leaves `from`'s table and enters `to`'s, on the
// device (both engines live on one GPU).  The caller guarantees that no batch of either engine is being formed or is in
// flight for those keys (the pool quiesces its stages first).
extern "C" int guber_move_items_by_hash(guber_engine_t* from, guber_engine_t* to, const uint64_t* hashes, uint32_t n, uint32_t* moved) {
    if (moved) *moved = 0;
    if (!from || !to || from == to || (n && !hashes)) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (from->device != to->device) return fail(GUBER_E_INVALID_ARG, "engines on different devices");
    if (n == 0) return GUBER_OK;
    guber_engine* a = from < to ? from : to; guber_engine* b = from < to ? to : from;     // address order, as every multi-locker
    std::lock_guard<std::mutex> la(a->mu); std::lock_guard<std::mutex> lb(b->mu);
    ep_flush_held(a); ep_flush_held(b);
    if (from->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const uint32_t stride = (std::min(from->max_key, to->max_key) + 23u) & ~7u;
    DevBuf<uint64_t>& d_h = from->d_mvh;
    if (d_h.ensure(n) || from->d_items.ensure(n) || from->d_ikeys.ensure((size_t)n * stride + 16) || to->d_islots.ensure(n) || to->d_iflags.ensure(n) ||
        to->d_ires.ensure(n)) return GUBER_E_NOMEM;
    std::vector<uint8_t> res(n, 0xFF);
    hipError_t he = hipSuccess;
    int rc = 0;
    bool taken = false;
    do {
        if ((he = hipMemcpyAsync(d_h.p, hashes, (size_t)n * 8, hipMemcpyHostToDevice, from->stream)) != hipSuccess) break;
        if ((he = hipMemsetAsync(to->d_ires.p, 0xFF, n, from->stream)) != hipSuccess) break;      // "not taken over" until the commit says otherwise
        hipLaunchKernelGGL(k_items_take_by_hash, dim3((n + 63) / 64), dim3(64), 0, from->stream, from->T, d_h.p, n, stride, from->d_items.p, from->d_ikeys.p);
        if ((he = hipStreamSynchronize(from->stream)) != hipSuccess) break;
        taken = true;
        rc = maintain(to, n, to->clock_ms);
        if (rc) break;
        note_enqueued(to, n);
        hipStream_t st = to->stream;
        hipLaunchKernelGGL(k_items_probe, dim3((n + 255) / 256), dim3(256), 0, st, to->T, from->d_items.p, from->d_ikeys.p, n, to->d_islots.p, to->d_iflags.p);
        hipLaunchKernelGGL(k_items_commit, dim3((n + 255) / 256), dim3(256), 0, st, to->T, from->d_items.p, from->d_ikeys.p, n, to->d_islots.p, to->d_iflags.p,
                           to->d_ires.p, take_stamps(to, n));
        if ((he = hipMemcpyAsync(res.data(), to->d_ires.p, n, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
        he = hipStreamSynchronize(st);
    } while (0);
    if (taken) {
        // whatever the destination did not take over goes back into the source (the commit's verdicts, or 0xFF for everything
        // when it never ran): a migration that fails loses no bucket
        (void)hipStreamSynchronize(to->stream);
        hipLaunchKernelGGL(k_items_restore, dim3((n + 63) / 64), dim3(64), 0, from->stream, from->T, from->d_items.p, from->d_ikeys.p, n, to->d_ires.p);
        (void)hipMemcpyAsync(res.data(), to->d_ires.p, n, hipMemcpyDeviceToHost, from->stream);
        (void)hipStreamSynchronize(from->stream);
    }
    if (he != hipSuccess) return fail(GUBER_E_HIP, "guber_move_items_by_hash", he);
    if (rc) return rc;
    uint32_t m = 0, back = 0;
    for (uint32_t i = 0; i < n; ++i) { m += res[i] <= 1; back += res[i] == 0xFD; }   // (0xFE / 0xFF left: the hash named no live bucket)
    if (moved) *moved = m;
    if (back) return fail(GUBER_E_TABLE_FULL, "guber_move_items_by_hash: the destination did not take every bucket; those went back to their table");
    if (moved) *moved = m;
    return GUBER_OK;
}
extern "C" void* guber_engine_stream(guber_engine_t* e) { return e ? (void*)e->stream : nullptr; }

extern "C" int guber_get_item(guber_engine_t* e, const uint8_t* key, uint32_t key_len, int64_t now_ms, guber_item_t* out, int* found) {
    return item_lookup(e, key, key_len, now_ms, 0, out, found);
}
extern "C" int guber_remove_item(guber_engine_t* e, const uint8_t* key, uint32_t key_len) {
    int found = 0;
    return item_lookup(e, key, key_len, 0, 1, nullptr, &found);
}

extern "C" int guber_stats(guber_engine_t* e, guber_stats_t* out) {
    if (!e || !out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    rc = maintain(e, 0, e->clock_ms);
    if (rc) return rc;
    const DevCounters& c = e->last_ctr;
    out->over_limit_count = c.over; out->cache_hits = c.hits; out->cache_misses = c.misses;
    out->unexpired_evictions = c.evictions; out->cache_size = c.size; out->table_slots = e->slots;
    out->tags_used = c.tags_used; out->batches = e->batches; out->retries = c.retries; out->compactions = e->compactions;
    out->small_batches = e->small_batches; out->fused_batches = e->fused_batches;
    out->eviction_passes = e->lru_applied; out->tail_rebuilds = e->lru_rebuilds; out->batch_cuts = e->lru_cuts;
    return GUBER_OK;
}
extern "C" int64_t guber_size(guber_engine_t* e) {
    guber_stats_t s;
    if (guber_stats(e, &s) != GUBER_OK) return -1;
    return s.cache_size;
}
extern "C" int guber_synchronize(guber_engine_t* e) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamSynchronize(e->stream));
    return GUBER_OK;
}

extern "C" int guber_dump(guber_engine_t* e, guber_item_t* items, uint64_t cap, uint8_t* key_arena, uint64_t arena_cap,
                          uint64_t* n_out, uint64_t* arena_out) {
    if (!e || !n_out || !arena_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    const uint64_t resident = (uint64_t)std::max<long long>(e->last_ctr.size, 0);
    DevBuf<Rec> d_recs; DevBuf<KeyCell> d_cells; DevBuf<unsigned long long> d_count;
    rc = d_recs.ensure(resident + 1) | d_cells.ensure(resident + 1) | d_count.ensure(1);
    auto cleanup = [&]() { d_recs.release(); d_cells.release(); d_count.release(); };
    if (rc) { cleanup(); return GUBER_E_NOMEM; }
    hipStream_t st = e->stream;
    hipError_t he;
    unsigned long long count = 0;
    std::vector<Rec> recs(resident + 1);
    std::vector<KeyCell> cells(resident + 1);
    if ((he = hipMemsetAsync(d_count.p, 0, sizeof(unsigned long long), st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "dump", he); }
    hipLaunchKernelGGL(k_dump, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, st, e->T, e->slots, d_recs.p, d_cells.p, resident + 1, d_count.p);
    if ((he = hipMemcpyAsync(&count, d_count.p, sizeof(count), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (he = hipStreamSynchronize(st)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "dump", he); }
    if (count > resident + 1) count = resident + 1;
    if ((he = hipMemcpy(recs.data(), d_recs.p, count * sizeof(Rec), hipMemcpyDeviceToHost)) != hipSuccess ||
        (he = hipMemcpy(cells.data(), d_cells.p, count * sizeof(KeyCell), hipMemcpyDeviceToHost)) != hipSuccess) { cleanup(); return fail(GUBER_E_HIP, "dump D2H", he); }
    cleanup();
    uint64_t need_arena = 0;
    for (uint64_t i = 0; i < count; ++i) need_arena += (uint32_t)(cells[i].w[7] >> 48);
    *n_out = count; *arena_out = need_arena;
    if (count > cap || need_arena > arena_cap || (!items && count) || (!key_arena && need_arena)) return fail(GUBER_E_NOMEM, "dump buffers too small");
    uint64_t aoff = 0;
    for (uint64_t i = 0; i < count; ++i) {
        item_from_rec(recs[i], &items[i]);
        const uint32_t len = (uint32_t)(cells[i].w[7] >> 48);
        uint8_t* dst = key_arena + aoff;
        if (len <= INLINE_KEY) memcpy(dst, cells[i].w, len);
        else if ((he = hipMemcpy(dst, e->arena.p + cells[i].w[0], len, hipMemcpyDeviceToHost)) != hipSuccess) return fail(GUBER_E_HIP, "dump long key", he);
        items[i].key = dst; items[i].key_len = len;
        aoff += len;
    }
    return GUBER_OK;
}

extern "C" void* guber_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void guber_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

// ---- device routing on the consistent-hash ring ------------------------------------------------
// ring image on the device, uploaded once per (engine, ring)
static int ensure_ring_on_device(guber_engine* e, const guber_ring_t* r) {
    if (e->ring_cached_id == guber_ring_id(r) && e->ring_npts) return 0;
    const uint32_t npts = guber_ring_points(r, nullptr, nullptr, 0);
    if (npts == 0) return fail(GUBER_E_INVALID_ARG, "empty ring");
    if ((size_t)npts * 8 > 150 * 1024) return fail(GUBER_E_INVALID_ARG, "ring does not fit in LDS");
    std::vector<uint64_t> hh(npts); std::vector<uint32_t> oo(npts);
    guber_ring_points(r, hh.data(), oo.data(), npts);
    if (e->d_ring_h.ensure(npts) || e->d_ring_o.ensure(npts)) return GUBER_E_NOMEM;
    HIPCHK(hipMemcpyAsync(e->d_ring_h.p, hh.data(), npts * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_ring_o.p, oo.data(), npts * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->ring_cached_id = guber_ring_id(r); e->ring_npts = npts;
    return 0;
}

extern "C" int guber_ring_route_dev(guber_engine_t* e, const guber_ring_t* r, const uint8_t* key_bytes,
                                    const uint32_t* key_off, uint32_t n, uint32_t* owner) {
    if (!e || !r || (n && (!key_bytes || !key_off || !owner))) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n == 0) return GUBER_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = ensure_ring_on_device(e, r);
    if (rc) return rc;
    hipLaunchKernelGGL(k_route, dim3((n + 255) / 256), dim3(256), (size_t)e->ring_npts * 8, e->stream, key_bytes, key_off, n,
                       e->d_ring_h.p, e->d_ring_o.p, e->ring_npts, guber_ring_kind(r), owner);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return GUBER_OK;
}

extern "C" int guber_ring_route_rows_dev(guber_engine_t* e, const guber_ring_t* r, const uint8_t* key_rows, uint32_t stride,
                                         const uint32_t* key_len, uint32_t n, uint32_t* owner) {
    if (!e || !r || (n && (!key_rows || !key_len || !owner))) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n == 0) return GUBER_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    int rc = ensure_ring_on_device(e, r);
    if (rc) return rc;
    hipLaunchKernelGGL(k_route_rows, dim3((n + 255) / 256), dim3(256), (size_t)e->ring_npts * 8, e->stream, key_rows, stride, key_len, n,
                       e->d_ring_h.p, e->d_ring_o.p, e->ring_npts, guber_ring_kind(r), owner);
    HIPCHK(hipGetLastError());
    return GUBER_OK;
}

extern "C" int guber_global_pending(guber_engine_t* e, uint32_t* n_out) {
    if (!e || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    *n_out = 0;
    if (!e->T.gpend) return fail(GUBER_E_INVALID_ARG, "engine created without GUBER_FLAG_GLOBAL");
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (c.gdirty_overflow) return fail(GUBER_E_NOMEM, "GLOBAL dirty list overflowed");
    *n_out = c.gdirty_n;
    return GUBER_OK;
}

// guber_global_take with the rows left in HBM, in caller-provided device arrays (cap rows each, key rows of
// out->key_stride bytes).  The rows feed guber_ring_route_rows_dev, an RCCL exchange and guber_eval_batch_dev /
// guber_add_items_dev without touching the host.
extern "C" int guber_global_take_dev(guber_engine_t* e, uint32_t role_mask, const guber_global_rows_dev_t* out, uint32_t* n_out) {
    if (!e || !out || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    *n_out = 0;
    if (!e->T.gpend) return fail(GUBER_E_INVALID_ARG, "engine created without GUBER_FLAG_GLOBAL");
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (c.gdirty_overflow) return fail(GUBER_E_NOMEM, "GLOBAL dirty list overflowed");
    const uint32_t n = c.gdirty_n;
    if (n == 0) return GUBER_OK;
    if (n > out->cap) { *n_out = n; return fail(GUBER_E_NOMEM, "row arrays too small"); }
    if (out->key_stride < e->max_key || !out->key_bytes || !out->key_len || !out->hits || !out->limit || !out->duration || !out->burst ||
        !out->created_at || !out->behavior || !out->algorithm || !out->role)
        return fail(GUBER_E_INVALID_ARG, "row arrays missing or key_stride < max_key_bytes");
    GTakeOut O{out->key_bytes, out->key_len, out->hits, out->limit, out->duration, out->burst, out->created_at, out->behavior,
               out->algorithm, out->role, out->key_stride};
    HIPCHK(hipMemsetAsync(e->gtake_ctr.p, 0, 4 * sizeof(uint32_t), e->stream));
    hipLaunchKernelGGL(k_global_take, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, n, role_mask, e->gdirty2.p,
                       (unsigned int*)e->gtake_ctr.p, O);
    unsigned int cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(cnt, e->gtake_ctr.p, sizeof(cnt), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    std::swap(e->gdirty.p, e->gdirty2.p);
    e->T.gdirty = e->gdirty.p;
    HIPCHK(hipMemcpyAsync(&e->ctr.p->gdirty_n, &cnt[1], sizeof(unsigned int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *n_out = cnt[0];
    return GUBER_OK;
}

// LRUCache.Add for device-resident item columns (keys must be distinct within one call: the receiver side of
// UpdatePeerGlobals, where every key comes from exactly one owner).  result[i] (device): 0 / 1 = existed,
// 0xFF = resubmit (in-call 64-bit hash collision or duplicate key), 0xFE = no directory entry.
extern "C" int guber_add_items_dev(guber_engine_t* e, const guber_items_dev_t* it, uint8_t* result) {
    if (!e || !it) return fail(GUBER_E_INVALID_ARG, "null argument");
    const uint32_t n = it->n;
    if (n == 0) return GUBER_OK;
    if (!result || !it->key_bytes || !it->key_off || !it->algorithm || !it->limit || !it->duration || !it->remaining || !it->remaining_f ||
        !it->stamp || !it->expire_at)
        return fail(GUBER_E_INVALID_ARG, "item column missing");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    {
        const int rc = maintain(e, n, e->clock_ms);
        if (rc) return rc;
    }
    note_enqueued(e, n);
    if (e->d_items.ensure(n) || e->d_islots.ensure(n) || e->d_iflags.ensure(n)) return GUBER_E_NOMEM;
    ItemsSoA S{it->key_off, it->algorithm, it->status, it->limit, it->duration, it->remaining, it->remaining_f, it->stamp, it->burst,
               it->expire_at, it->invalid_at};
    hipStream_t st = e->stream;
    hipLaunchKernelGGL(k_items_from_soa, dim3((n + 255) / 256), dim3(256), 0, st, S, n, e->d_items.p);
    hipLaunchKernelGGL(k_items_probe, dim3((n + 255) / 256), dim3(256), 0, st, e->T, e->d_items.p, it->key_bytes, n, e->d_islots.p, e->d_iflags.p);
    hipLaunchKernelGGL(k_items_commit, dim3((n + 255) / 256), dim3(256), 0, st, e->T, e->d_items.p, it->key_bytes, n, e->d_islots.p, e->d_iflags.p, result, take_stamps(e, n));
    HIPCHK(hipGetLastError());
    return GUBER_OK;
}

extern "C" int guber_global_take(guber_engine_t* e, uint32_t role_mask, guber_global_rows_t* out) {
    if (!e || !out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    memset(out, 0, sizeof(*out));
    if (!e->T.gpend) return fail(GUBER_E_INVALID_ARG, "engine created without GUBER_FLAG_GLOBAL");
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (c.gdirty_overflow) return fail(GUBER_E_NOMEM, "GLOBAL dirty list overflowed");
    const uint32_t n = c.gdirty_n;
    const uint32_t stride = (e->max_key + 7u) & ~7u;
    out->key_stride = stride;
    if (n == 0) return GUBER_OK;
    // one device + one pinned arena: keys | 5 x i64 | key_len u32 | behavior u32 | algorithm u8 | role u8
    const size_t o_keys = 0, o_i64 = (size_t)n * stride, o_len = o_i64 + (size_t)n * 40, o_beh = o_len + (size_t)n * 4,
                 o_alg = o_beh + (size_t)n * 4, o_role = o_alg + n, total = o_role + n + 64;
    int rc = e->d_take.ensure(total) | e->h_take.ensure(total);
    if (rc) return GUBER_E_NOMEM;
    uint8_t* d = e->d_take.p;
    GTakeOut O{d + o_keys, (uint32_t*)(d + o_len), (int64_t*)(d + o_i64), (int64_t*)(d + o_i64) + n, (int64_t*)(d + o_i64) + 2 * (size_t)n,
               (int64_t*)(d + o_i64) + 3 * (size_t)n, (int64_t*)(d + o_i64) + 4 * (size_t)n, (uint32_t*)(d + o_beh), d + o_alg, d + o_role, stride};
    HIPCHK(hipMemsetAsync(e->gtake_ctr.p, 0, 4 * sizeof(uint32_t), e->stream));
    hipLaunchKernelGGL(k_global_take, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, n, role_mask, e->gdirty2.p,
                       (unsigned int*)e->gtake_ctr.p, O);
    unsigned int cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(cnt, e->gtake_ctr.p, sizeof(cnt), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // rows that were not asked for stay queued: the kept list becomes the dirty list
    std::swap(e->gdirty.p, e->gdirty2.p);
    e->T.gdirty = e->gdirty.p;
    HIPCHK(hipMemcpyAsync(&e->ctr.p->gdirty_n, &cnt[1], sizeof(unsigned int), hipMemcpyHostToDevice, e->stream));
    const uint32_t m = cnt[0];
    if (m) HIPCHK(hipMemcpyAsync(e->h_take.p, d, total, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const uint8_t* h = e->h_take.p;
    out->n = m; out->key_bytes = h + o_keys; out->key_len = (const uint32_t*)(h + o_len);
    out->hits = (const int64_t*)(h + o_i64); out->limit = out->hits + n; out->duration = out->hits + 2 * (size_t)n;
    out->burst = out->hits + 3 * (size_t)n; out->created_at = out->hits + 4 * (size_t)n;
    out->behavior = (const uint32_t*)(h + o_beh); out->algorithm = h + o_alg; out->role = h + o_role;
    return GUBER_OK;
}
