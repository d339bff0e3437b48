// engine_stages.inl — part of guber_engine.hip's translation unit (included there, in this order; not a header of its own):
// stages: the overlapped end-to-end path, groups of stages in one submission, the stage routed to several engines.
// ---- stages: batch buffers in device-visible host memory that the CALLER fills in place and the kernels read / write in
// place.  A batcher that owns two of them fills one while the GPU evaluates the other: no staging copy, no copy launch,
// no allocation per batch (what SURVEY.md section 8d calls the overlapped end-to-end path).
struct guber_stage {
    guber_engine* e = nullptr;
    uint32_t max_n = 0, key_cap = 0;
    CohBuf<uint8_t> mem;
    guber_batch_t batch{}; guber_result_t result{};
    SmallOut* sout = nullptr;
    DevCounters* rb_ctr = nullptr; BlockCounters* rb_bctr = nullptr;     // this stage's own counter read-back after its batch ...
    DevCounters* rb0_ctr = nullptr; BlockCounters* rb0_bctr = nullptr;   // ... and before it: the difference is exactly this batch
    hipEvent_t ev = nullptr;
    guber_engine::GroupEv* gev = nullptr; uint32_t gev_seq = 0;   // submitted as one of a group: the group's completion event (the slot's use)
    MultiArgsMem* h_margs = nullptr;                               // argument blocks of a group this stage leads (device-visible host memory)
    uint32_t* h_dest = nullptr;                                    // guber_stage_submit_routed: per request, engine index << 24 | rank in that engine's share
    std::vector<guber_engine*> routed;                             // ... and the engines of the submission in flight (retries go back to them)
    // a routed stage of <= 256 requests: one workgroup per engine in ONE launch (k_small_routed); every share has its own outcome
    struct RoutedPart { guber_engine* e; uint32_t engine, n, seq; SmallOut* out; bool pending; int rc; };
    std::vector<RoutedPart> parts; uint8_t* h_parts_out = nullptr;  // (mode 4; a part is only touched under its engine's mutex)
    // guber_stage_route: the shares' sizes + completion flag (host, device-visible), per-request engine and per-tile tables (HBM)
    uint32_t* h_route = nullptr; DevBuf<uint8_t> d_route; uint32_t route_seq = 0; bool route_pending = false; uint32_t route_engines = 0;
    bool keys_resident = false;      // guber_stage_route left this batch's key bytes in the HBM mirror (dmem): guber_stage_submit_routed does not copy them again
    uint32_t seq = 0, n = 0; int64_t now_ms = 0;
    int mode = 0;                    // 0 idle, 1 small path complete, 2 pipeline in flight, 3 small path launched, outcome not looked at yet (guber_stages_submit),
                                     // 4 routed small path launched (guber_stage_submit_routed): outcomes per part
    bool no_agg = false;             // submitted without per-batch aggregates (guber_stages_submit)
    // Large batches: two DMA copies on a copy stream (the fixed-width columns present, the keys) bring the requests into the
    // stage's device mirror while the previous batches' kernels run; the pipeline then works on HBM and k_eval2 writes the
    // responses straight into the host arrays (posted writes).  The link carries the requests at the copy engine's rate
    // (46-48 GB/s) instead of at the rate of k_front's dependent reads (24 GB/s in total with everything in place).
    // Measured and dropped (profiles/archive/r02_v_end_to_end_variants.txt): responses to HBM and a DMA copy back (a hipMemcpyAsync
    // costs 40-60 us of host time), a copy kernel instead of the DMA (kernels of two streams overlap badly).
    DevBuf<uint8_t> dmem;            // device mirror of the in block
    uint8_t *h_in = nullptr, *h_out = nullptr; size_t in_fixed = 0, out_bytes = 0;   // host blocks; in_fixed = bytes before the keys
    hipEvent_t ev_in = nullptr;
};

static int resolve_small(guber_stage* s, bool block);
static int resolve_routed_small(guber_stage* s, bool block);
static int resolve_small_locked(guber_stage* s, bool block, guber_engine* holder = nullptr);
extern "C" int guber_stage_create(guber_engine_t* e, uint32_t max_n, uint32_t key_bytes_cap, guber_stage_t** out) {
    if (!e || !out || max_n == 0) return fail(GUBER_E_INVALID_ARG, "null argument");
    *out = nullptr;
    if (max_n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "stage larger than guber_config_t.max_batch");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    guber_stage* s = new guber_stage();
    s->e = e; s->max_n = max_n; s->key_cap = key_bytes_cap ? key_bytes_cap : max_n * 64u;
    const size_t n = max_n;
    // [counters | in block: key_off, hits, limit, duration, behavior, algorithm, is_owner, burst, created_at, keys |
    //  out block: limit, remaining, reset_time, status, err]; every column starts on a 64-byte boundary and is padded to one
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    const size_t in_fixed = col((n + 1) * 4) + 5 * col(n * 8) + col(n * 4) + 2 * col(n);
    const size_t in_bytes = col(in_fixed + (size_t)s->key_cap + 64);
    const size_t out_bytes = 3 * col(n * 8) + 2 * col(n);
    const size_t head = 256 + 2 * (col(sizeof(DevCounters)) + col((size_t)e->n_bctr * sizeof(BlockCounters))) + col(sizeof(MultiArgsMem)) + col(n * 4) + 16 * 64 + 128;
    const size_t bytes = head + in_bytes + out_bytes + 256;
    if (s->mem.ensure(bytes) || hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming) != hipSuccess) {
        s->mem.release(); delete s; return GUBER_E_NOMEM;
    }
    memset(s->mem.p, 0, bytes);
    uint8_t* p = s->mem.p;
    s->sout = (SmallOut*)p; p += 64;
    s->rb_ctr = (DevCounters*)p; p += col(sizeof(DevCounters));
    s->rb_bctr = (BlockCounters*)p; p += col((size_t)e->n_bctr * sizeof(BlockCounters));
    s->rb0_ctr = (DevCounters*)p; p += col(sizeof(DevCounters));
    s->rb0_bctr = (BlockCounters*)p; p += col((size_t)e->n_bctr * sizeof(BlockCounters));
    s->h_margs = (MultiArgsMem*)p; p += col(sizeof(MultiArgsMem));
    s->h_dest = (uint32_t*)p; p += col(n * 4);
    s->h_parts_out = p; p += 16 * 64;
    s->h_route = (uint32_t*)p;                                     // [0..15] counts, [16] done flag
    p = s->mem.p + head;
    s->h_in = p; s->h_out = p + in_bytes; s->in_fixed = in_fixed; s->out_bytes = out_bytes;
    guber_batch_t& b = s->batch; guber_result_t& r = s->result;
    b.key_off = (uint32_t*)p; p += col((n + 1) * 4);
    b.hits = (int64_t*)p; p += col(n * 8);
    b.limit = (int64_t*)p; p += col(n * 8);
    b.duration = (int64_t*)p; p += col(n * 8);
    b.behavior = (uint32_t*)p; p += col(n * 4);
    b.algorithm = p; p += col(n);
    b.is_owner = p; p += col(n);
    b.burst = (int64_t*)p; p += col(n * 8);
    b.created_at = (int64_t*)p; p += col(n * 8);
    b.key_bytes = p;
    p = s->h_out;
    r.limit = (int64_t*)p; p += col(n * 8);
    r.remaining = (int64_t*)p; p += col(n * 8);
    r.reset_time = (int64_t*)p; p += col(n * 8);
    r.status = p; p += col(n);
    r.err = p;
    *out = s;
    return GUBER_OK;
}
// no engine may keep pointing at a stage that is being abandoned or freed (its next submit would look at it: resolve_small_locked)
static void forget_small_pending(guber_stage* s) {
    std::vector<guber_engine*> engs;
    if (s->e) engs.push_back(s->e);
    for (auto& part : s->parts) if (part.e && std::find(engs.begin(), engs.end(), part.e) == engs.end()) engs.push_back(part.e);
    for (guber_engine* e : engs) {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->small_pending == s) e->small_pending = nullptr;
    }
}
extern "C" void guber_stage_destroy(guber_stage_t* s) {
    if (!s) return;
    if (s->mode) (void)guber_stage_wait(s);
    if (s->route_pending && s->e && !s->e->set_device()) (void)hipStreamSynchronize(s->e->stream);   // (the routing launches write into the stage)
    forget_small_pending(s);
    if (s->ev) (void)hipEventDestroy(s->ev);
    if (s->ev_in) (void)hipEventDestroy(s->ev_in);
    s->dmem.release();
    s->d_route.release();
    s->mem.release();
    delete s;
}
extern "C" guber_batch_t* guber_stage_batch(guber_stage_t* s) { return s ? &s->batch : nullptr; }
extern "C" guber_result_t* guber_stage_result(guber_stage_t* s) { return s ? &s->result : nullptr; }
extern "C" uint32_t guber_stage_capacity(guber_stage_t* s, uint32_t* key_bytes_cap) { if (s && key_bytes_cap) *key_bytes_cap = s->key_cap; return s ? s->max_n : 0; }

extern "C" int guber_stage_submit(guber_stage_t* s) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    if (s->mode) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
    guber_engine* e = s->e;
    const guber_batch_t& b = s->batch;
    s->n = b.n; s->now_ms = b.now_ms; s->no_agg = false;
    if (b.n == 0) { s->mode = 0; return GUBER_OK; }
    if (b.n > s->max_n || b.key_off[b.n] > s->key_cap) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled");
    memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);            // the kernels read keys as 8-byte words
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    if (e->small_pending) { const int rcp = resolve_small_locked(e->small_pending, true, e); if (rcp < 0) return rcp; }
    BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                b.greg_expire, b.greg_duration, b.now_ms};
    ResultView R{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
    if (b.n <= FT && !e->no_small && !lru_may_bind(e, b.n)) {
        s->seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
        s->sout->done = 0;
        int rc = launch_small(e, B, R, s->sout, s->seq);
        if (rc) return rc;
        // One workgroup, a few microseconds: wait for it here.  If the small path declined the batch (requests of one key
        // that differ, a hash collision) the general pipeline has to run it BEFORE anything submitted later, so the
        // decision cannot be left to guber_stage_wait.
        volatile unsigned int* flag = &s->sout->done;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != s->seq) {
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(e->stream)); break; }
        }
        if (!s->sout->fallback) {
            e->last_ctr.over += s->sout->over; e->last_ctr.hits += s->sout->hits; e->last_ctr.misses += s->sout->misses; e->last_ctr.size += s->sout->size_delta;
            s->mode = 1;
            return GUBER_OK;
        }
        e->small_fallbacks++;
    }
    // (maintenance first: it may synchronise and rebuild; the read-back pair must bracket the kernels only)
    int rc = maintain(e, b.n, b.now_ms);
    if (rc) return rc;
    // a batch that fills at least half of the stage reaches HBM by DMA; smaller ones are read in place
    const bool dma = e->stage_dma && s->max_n >= 4096 && (size_t)b.n * 2 >= s->max_n && !b.greg_expire && !b.greg_duration;
    if (dma) {
        if (!e->copy_in && hipStreamCreateWithFlags(&e->copy_in, hipStreamNonBlocking) != hipSuccess) return fail(GUBER_E_HIP, "hipStreamCreate");
        const size_t in_bytes = (size_t)(s->h_out - s->h_in);
        if (s->dmem.ensure(in_bytes)) return GUBER_E_NOMEM;
        uint8_t* d_in = s->dmem.p;
        // two copies: the fixed-width columns up to the last one present, then the keys
        const void* last = b.created_at ? (const void*)(b.created_at + b.n) : b.burst ? (const void*)(b.burst + b.n) : b.is_owner ? (const void*)(b.is_owner + b.n) : (const void*)(b.algorithm + b.n);
        const size_t fixed = (size_t)((const uint8_t*)last - s->h_in);
        HIPCHK(hipMemcpyAsync(d_in, s->h_in, fixed, hipMemcpyHostToDevice, e->copy_in));
        HIPCHK(hipMemcpyAsync(d_in + s->in_fixed, s->h_in + s->in_fixed, (size_t)b.key_off[b.n] + 16, hipMemcpyHostToDevice, e->copy_in));
        HIPCHK(hipEventRecord(s->ev_in, e->copy_in));
        HIPCHK(hipStreamWaitEvent(e->stream, s->ev_in, 0));
        auto dev = [&](const void* hp) { return hp ? d_in + ((const uint8_t*)hp - s->h_in) : nullptr; };
        B = BatchView{b.n, 0, dev(b.key_bytes), (const uint32_t*)dev(b.key_off), (const int64_t*)dev(b.hits), (const int64_t*)dev(b.limit),
                      (const int64_t*)dev(b.duration), (const int64_t*)dev(b.burst), (const int64_t*)dev(b.created_at), dev(b.algorithm),
                      (const uint32_t*)dev(b.behavior), dev(b.is_owner), nullptr, nullptr, b.now_ms};
    }
    hipLaunchKernelGGL(k_ctr_snapshot, dim3(1), dim3(256), 0, e->stream, e->ctr.p, e->bctr.p, e->n_bctr, s->rb0_ctr, s->rb0_bctr, (uint32_t*)nullptr, 0u);
    rc = launch_batch(e, B, R, !dma);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ctr_snapshot, dim3(1), dim3(256), 0, e->stream, e->ctr.p, e->bctr.p, e->n_bctr, s->rb_ctr, s->rb_bctr, (uint32_t*)nullptr, 0u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev, e->stream));
    s->gev = nullptr; s->mode = 2;
    return GUBER_OK;
}

extern "C" int guber_stage_wait(guber_stage_t* s) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    guber_engine* e = s->e;
    guber_result_t& r = s->result;
    r.over_limit_count = r.cache_hits = r.cache_misses = r.unexpired_evictions = 0;
    if (s->mode == 0) return GUBER_OK;
    if (s->mode == 3) {                                      // launched by guber_stages_submit: look at the outcome now
        const int rc3 = resolve_small(s, true);
        if (rc3 < 0) return rc3;
    }
    if (s->mode == 4) {                                      // a routed stage on the one-launch path: every share's outcome
        const int rc4 = resolve_routed_small(s, true);
        if (rc4 < 0) { forget_small_pending(s); s->mode = 0; s->routed.clear(); s->parts.clear(); return rc4; }
    }
    bool general = s->mode == 2;
    if (s->mode == 1) {                                      // answered by the one-launch path, already complete (guber_stage_submit)
        s->mode = 0;
        if (s->parts.empty()) {
            std::lock_guard<std::mutex> lk(e->mu);
            r.over_limit_count = s->sout->over; r.cache_hits = s->sout->hits; r.cache_misses = s->sout->misses; r.cache_size = e->last_ctr.size;
            return GUBER_OK;
        }
        s->parts.clear();                                    // (a routed stage: no aggregates; a re-run share may have left internal retries)
    }
    if (general) {
        if (s->gev) {
            if (s->gev->seq.load(std::memory_order_acquire) == s->gev_seq && hipEventSynchronize(s->gev->ev) != hipSuccess) return fail(GUBER_E_HIP, "hipEventSynchronize");
            s->gev = nullptr;
        } else if (hipEventSynchronize(s->ev) != hipSuccess) return fail(GUBER_E_HIP, "hipEventSynchronize");
        s->mode = 0;
        std::lock_guard<std::mutex> lk(e->mu);
        if (s->no_agg) general = false;                       // no read-backs were taken: the aggregates stay 0 (guber_stats has the totals)
        // the read-backs taken right before and right after this batch's kernels: their difference is this batch alone
        auto fold = [&](const DevCounters* c0, const BlockCounters* b0) {
            DevCounters c = *c0;
            for (uint32_t k = 0; k < e->n_bctr; ++k) { c.over += b0[k].over; c.hits += b0[k].hits; c.misses += b0[k].misses; c.size += b0[k].size_delta; }
            return c;
        };
        if (general) {
            const DevCounters c1 = fold(s->rb_ctr, s->rb_bctr), c0 = fold(s->rb0_ctr, s->rb0_bctr);
            r.over_limit_count = c1.over - c0.over; r.cache_hits = c1.hits - c0.hits; r.cache_misses = c1.misses - c0.misses;
            r.cache_size = c1.size;
        }
    }
    // two new keys sharing one 64-bit hash (or one claim fingerprint) inside the batch: re-submit those items on the host
    // path, which runs the careful rounds
    if (memchr(r.err, GUBER_ITEM_E_RETRY, s->n)) {
        std::vector<guber_engine*> engs = s->routed;
        if (engs.empty()) engs.push_back(e);
        for (size_t j = 0; j < engs.size(); ++j) {
            guber_engine* ej = engs[j];
            std::vector<uint32_t> again;
            for (uint32_t i = 0; i < s->n; ++i)
                if (r.err[i] == GUBER_ITEM_E_RETRY && (s->routed.empty() || (s->h_dest[i] >> 24) == j)) again.push_back(i);
            if (again.empty()) continue;
            std::lock_guard<std::mutex> lk(ej->mu);
            if (ej->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
            guber_batch_t hb = s->batch;
            int rc0 = engine_refresh_counters(ej);
            if (rc0) return rc0;
            const DevCounters t0 = ej->last_ctr;
            for (int round = 0; round < 64 && !again.empty(); ++round) {
                ej->careful = true;
                const int rc = eval_host_once(ej, &hb, &r, again.data(), (uint32_t)again.size(), nullptr);
                ej->careful = false;
                if (rc) return rc;
                std::vector<uint32_t> next;
                for (uint32_t i : again) if (r.err[i] == GUBER_ITEM_E_RETRY) next.push_back(i);
                again.swap(next);
            }
            r.over_limit_count += ej->last_ctr.over - t0.over; r.cache_hits += ej->last_ctr.hits - t0.hits; r.cache_misses += ej->last_ctr.misses - t0.misses;
            r.cache_size = ej->last_ctr.size;
        }
    }
    s->routed.clear();
    return GUBER_OK;
}

// ---- several stages in one submission: what the dispatcher of a GPUWorkerPool calls (worker_pool.cpp).  Never waits for the GPU.
// A <= 256-request stage launched here is in mode 3 until somebody looks at its outcome (guber_stage_poll / guber_stage_wait,
// or the next submission on its engine): the one-launch path may decline a batch (requests of one key that differ, a hash
// collision), and then the general pipeline has to run it before anything later of the same engine.
// the shares of a routed small stage that belong to `holder` (its mutex held): outcome looked at, a declined share re-run through
// the general pipeline — synchronously, on the host-pointer path, picking the share out of the stage by its ranks
static int resolve_routed_parts_locked(guber_stage* s, guber_engine* holder, bool block) {
    for (auto& part : s->parts) {
        if (part.e != holder || !part.pending) continue;
        volatile unsigned int* flag = &part.out->done;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != part.seq) {
            if (!block) return 0;
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(holder->stream)); break; }
        }
        part.pending = false;
        if (holder->small_pending == s) holder->small_pending = nullptr;
        if (!part.out->fallback) {
            holder->last_ctr.over += part.out->over; holder->last_ctr.hits += part.out->hits; holder->last_ctr.misses += part.out->misses; holder->last_ctr.size += part.out->size_delta;
            continue;
        }
        holder->small_fallbacks++;
        if (holder->set_device()) return part.rc = fail(GUBER_E_HIP, "hipSetDevice");
        // the kernel also declines a share whose ranks in dest are not a permutation of 0..n-1 (the caller wrote dest): that is a
        // caller error, not a batch for the general pipeline
        std::vector<uint32_t> idx(part.n, 0xffffffffu);
        uint32_t placed = 0;
        for (uint32_t i = 0; i < s->n; ++i) {
            if ((s->h_dest[i] >> 24) != part.engine) continue;
            const uint32_t rk = s->h_dest[i] & 0xffffffu;
            if (rk >= part.n || idx[rk] != 0xffffffffu) { placed = 0xffffffffu; break; }
            idx[rk] = i; ++placed;
        }
        if (placed != part.n) return part.rc = fail(GUBER_E_INVALID_ARG, "dest: the ranks of an engine's share are not a permutation of 0 .. count-1");
        guber_batch_t hb = s->batch;
        part.rc = eval_host_once(holder, &hb, &s->result, idx.data(), part.n, nullptr);
        if (part.rc) return part.rc;
    }
    return 1;
}
static int resolve_small_locked(guber_stage* s, bool block, guber_engine* holder) {   // engine mutex held; 1 = resolved, 0 = still running
    guber_engine* e = s->e;
    if (s->mode == 4) return resolve_routed_parts_locked(s, holder ? holder : e, block);
    if (s->mode != 3) return 1;
    volatile unsigned int* flag = &s->sout->done;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != s->seq) {
        if (!block) return 0;
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(e->stream)); break; }
    }
    if (e->small_pending == s) e->small_pending = nullptr;
    if (!s->sout->fallback) {
        e->last_ctr.over += s->sout->over; e->last_ctr.hits += s->sout->hits; e->last_ctr.misses += s->sout->misses; e->last_ctr.size += s->sout->size_delta;
        s->mode = 1;
        return 1;
    }
    e->small_fallbacks++;
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const guber_batch_t& b = s->batch;
    BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                b.greg_expire, b.greg_duration, b.now_ms};
    ResultView R{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
    int rc = launch_batch(e, B, R, true);
    if (rc) { s->mode = 0; return rc; }
    HIPCHK(hipEventRecord(s->ev, e->stream));
    s->gev = nullptr; s->mode = 2;
    return 1;
}
static int resolve_small(guber_stage* s, bool block) {
    std::lock_guard<std::mutex> lk(s->e->mu);
    return resolve_small_locked(s, block);
}
// every share of a routed small stage (mode 4), each under its engine's mutex; all resolved: the stage is complete (mode 1)
static int resolve_routed_small(guber_stage* s, bool block) {
    for (size_t k = 0; k < s->parts.size(); ++k) {
        guber_engine* e = s->parts[k].e;
        std::lock_guard<std::mutex> lk(e->mu);
        if (s->parts[k].rc) return s->parts[k].rc;
        if (!s->parts[k].pending) continue;
        const int rc = resolve_routed_parts_locked(s, e, block);
        if (rc <= 0) return rc;
    }
    s->mode = 1;
    return 1;
}

extern "C" int guber_stage_poll(guber_stage_t* s) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    if (s->mode == 3) {
        const int rc = resolve_small(s, false);
        if (rc <= 0) return rc;
    }
    if (s->mode == 4) {
        const int rc = resolve_routed_small(s, false);
        if (rc <= 0) return rc;
    }
    if (s->mode == 2) {
        if (s->gev && s->gev->seq.load(std::memory_order_acquire) != s->gev_seq) return 1;   // the group's event slot has moved on: complete
        const hipError_t q = hipEventQuery(s->gev ? s->gev->ev : s->ev);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return fail(GUBER_E_HIP, "hipEventQuery", q);
    }
    return 1;
}

namespace {
struct StagePlan { guber_stage* s; BatchView B; ResultView R; bool copy; StageIn in; };
}
// the views of a stage batch: it reaches HBM through the copy kernel (the stage's device mirror; one PCIe round trip for all of
// it, where k_front reading host memory in place pays one per dependent load: key offset, key bytes, fields) unless the engine
// was told otherwise (GUBER_STAGE_COPY_MIN / GUBER_NO_STAGE_DMA: k_front then keeps a copy of the request columns for k_eval2)
static int stage_views(guber_stage* s, StagePlan& P) {
    guber_engine* e = s->e;
    const guber_batch_t& b = s->batch;
    P.s = s;
    P.B = BatchView{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                    b.greg_expire, b.greg_duration, b.now_ms};
    P.R = ResultView{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
    P.copy = e->stage_dma && b.n >= e->stage_copy_min && !b.greg_expire && !b.greg_duration;
    P.in = StageIn{};
    if (!P.copy) return 0;
    const size_t in_bytes = (size_t)(s->h_out - s->h_in);
    if (s->dmem.ensure(in_bytes)) return GUBER_E_NOMEM;
    uint8_t* d_in = s->dmem.p;
    P.in.src = (const uint4*)s->h_in; P.in.dst = (uint4*)d_in;
    auto seg = [&](const void* col, size_t bytes) {                 // the first `bytes` of a column that is present
        if (!col || !bytes) return;
        P.in.off16[P.in.nseg] = (uint32_t)(((const uint8_t*)col - s->h_in) / 16);
        P.in.n16[P.in.nseg] = (uint32_t)((bytes + 15) / 16);
        P.in.nseg++;
    };
    const size_t n = b.n;
    seg(b.key_off, (n + 1) * 4); seg(b.hits, n * 8); seg(b.limit, n * 8); seg(b.duration, n * 8); seg(b.behavior, n * 4);
    seg(b.algorithm, n); seg(b.is_owner, n); seg(b.burst, n * 8); seg(b.created_at, n * 8); seg(b.key_bytes, (size_t)b.key_off[b.n] + 16);
    auto dev = [&](const void* hp) { return hp ? d_in + ((const uint8_t*)hp - s->h_in) : nullptr; };
    P.B = BatchView{b.n, 0, dev(b.key_bytes), (const uint32_t*)dev(b.key_off), (const int64_t*)dev(b.hits), (const int64_t*)dev(b.limit),
                    (const int64_t*)dev(b.duration), (const int64_t*)dev(b.burst), (const int64_t*)dev(b.created_at), dev(b.algorithm),
                    (const uint32_t*)dev(b.behavior), dev(b.is_owner), nullptr, nullptr, b.now_ms};
    return 0;
}

// one group: <= MULTI_MEM_MAX large stages of engines that share device and stream (engine mutexes held by the caller).  Up to
// MULTI_MAX of them take the launches whose arguments travel by value; more (a pool dispatcher's generation over 8, 12 shards)
// take the same launches with their argument blocks in device memory: written into the leading stage's host block here,
// brought over by the copy kernel that also moves the request columns.  Three launches and one event record per group.
static int launch_stage_group(StagePlan* P, int g) {
    guber_engine* e0 = P[0].s->e;
    const bool mem_args = g > MULTI_MAX;
    MultiStageIn MI{}; MultiFront MF{}; MultiEval ME{};
    MultiArgsMem* HA = P[0].s->h_margs;
    uint32_t tiles = 0; int planned = 0, rc = 0; bool any_copy = false;
    FastPlan FP[MULTI_MEM_MAX];
    for (int i = 0; i < g; ++i) {
        guber_engine* e = P[i].s->e;
        Work W;
        rc = batch_prelude(e, P[i].B, W);
        if (!rc) rc = plan_fast(e, P[i].B, !P[i].copy, W, FP[i]);
        if (rc) break;
        tiles += FP[i].ftiles;
        if (mem_args) {
            HA->F.end_tile[planned] = HA->E.end_tile[planned] = tiles;
            HA->F.sub[planned] = FrontArgs{e->T, FP[i].B2, FP[i].W};
            HA->E.sub[planned] = EvalArgs{e->T, FP[i].B3, P[i].R, FP[i].W};
        } else {
            MF.end_tile[planned] = ME.end_tile[planned] = tiles;
            MF.sub[planned] = FrontArgs{e->T, FP[i].B2, FP[i].W};
            ME.sub[planned] = EvalArgs{e->T, FP[i].B3, P[i].R, FP[i].W};
        }
        MI.sub[planned] = P[i].in;
        any_copy = any_copy || P[i].copy;
        ++planned;
    }
    if (!planned) return rc;
    hipStream_t st = e0->stream;
    MultiArgsMem* DA = nullptr;
    MI.nb = (uint32_t)planned;
    if (mem_args) {
        if (e0->d_margs.ensure(sizeof(MultiArgsMem))) return GUBER_E_NOMEM;
        DA = (MultiArgsMem*)e0->d_margs.p;
        HA->F.nb = HA->E.nb = (uint32_t)planned;
        StageIn& a = MI.sub[MI.nb++];                                // the argument blocks: one more segment list of the copy kernel
        a = StageIn{};
        a.src = (const uint4*)HA; a.dst = (uint4*)DA; a.nseg = 2;
        a.off16[0] = 0; a.n16[0] = (uint32_t)((offsetof(MultiFrontMem, sub) + (size_t)planned * sizeof(FrontArgs) + 15) / 16);
        a.off16[1] = (uint32_t)(offsetof(MultiArgsMem, E) / 16); a.n16[1] = (uint32_t)((offsetof(MultiEvalMem, sub) + (size_t)planned * sizeof(EvalArgs) + 15) / 16);
    }
    if (any_copy || mem_args) {
        MI.wg_per = 64;
        hipLaunchKernelGGL(k_stage_in_multi, dim3(MI.nb * MI.wg_per), dim3(256), 0, st, MI);
    }
    uint64_t units = 0;
    for (int i = 0; i < planned; ++i) units += P[i].B.n;
    if (planned == 1) {
        e0->span_begin(KT_FRONT, units);
        hipLaunchKernelGGL(k_front, dim3(FP[0].ftiles), dim3(FT), 0, st, e0->T, FP[0].B2, FP[0].W);
        e0->span_end();
        e0->span_begin(KT_EVAL2, units);
        hipLaunchKernelGGL(k_eval2, dim3(FP[0].ftiles), dim3(256), 0, st, EvalArgs{e0->T, FP[0].B3, P[0].R, FP[0].W});
        e0->span_end();
    } else if (mem_args) {
        e0->span_begin(KT_FRONT_MULTI, units);
        hipLaunchKernelGGL(k_front_multi_mem, dim3(tiles), dim3(FT), 0, st, (const MultiFrontMem*)&DA->F);
        e0->span_end();
        e0->span_begin(KT_EVAL2_MULTI, units);
        hipLaunchKernelGGL(k_eval2_multi_mem, dim3(tiles), dim3(256), 0, st, (const MultiEvalMem*)&DA->E);
        e0->span_end();
    } else {
        MF.nb = ME.nb = (uint32_t)planned;
        e0->span_begin(KT_FRONT_MULTI, units);
        hipLaunchKernelGGL(k_front_multi, dim3(tiles), dim3(FT), 0, st, MF);
        e0->span_end();
        e0->span_begin(KT_EVAL2_MULTI, units);
        hipLaunchKernelGGL(k_eval2_multi, dim3(tiles), dim3(256), 0, st, ME);
        e0->span_end();
    }
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    guber_engine::GroupEv* G = nullptr; uint32_t gseq = 0;
    if (planned > 1) {
        G = &e0->gev[e0->gev_next++ % guber_engine::kGroupEvs];
        if (!G->ev) { if (hipEventCreateWithFlags(&G->ev, hipEventDisableTiming) != hipSuccess) return fail(GUBER_E_HIP, "hipEventCreate"); }
        else if (hipEventSynchronize(G->ev) != hipSuccess) return fail(GUBER_E_HIP, "hipEventSynchronize");   // (kGroupEvs groups back: long complete)
        gseq = G->seq.load(std::memory_order_relaxed) + 1;
        if (hipEventRecord(G->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
        G->seq.store(gseq, std::memory_order_release);
    }
    for (int i = 0; i < planned; ++i) {
        guber_stage* s = P[i].s;
        finish_fast(s->e, P[i].B.n);
        if (planned > 1) s->e->fused_batches++;
        s->gev = G; s->gev_seq = gseq;
        if (!G && hipEventRecord(s->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
        s->mode = 2;
    }
    return rc;
}

extern "C" int guber_stages_submit(guber_stage_t* const* stages, uint32_t n, uint32_t flags, uint32_t* done) {
    if (done) *done = 0;
    if (!stages && n) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (!(flags & GUBER_STAGES_NO_AGGREGATES)) {                  // per-batch aggregates wanted: the stages go one by one
        for (uint32_t k = 0; k < n; ++k) {
            const int rc = guber_stage_submit(stages[k]);
            if (rc) return rc;
            if (done) *done = k + 1;
        }
        return GUBER_OK;
    }
    for (uint32_t k = 0; k < n; ++k) {
        guber_stage* s = stages[k];
        if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
        if (s->mode) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
        const guber_batch_t& b = s->batch;
        if (b.n > s->max_n || (b.n && b.key_off[b.n] > s->key_cap)) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled");
        for (uint32_t q = 0; q < k; ++q) if (stages[q]->e == s->e) return fail(GUBER_E_INVALID_ARG, "two stages of one engine in one submission");
    }
    uint32_t enq = 0;
    StagePlan grp[MULTI_MEM_MAX]; int g = 0;
    const int group_max = MULTI_MEM_MAX;
    auto flush = [&]() -> int {
        if (!g) return 0;
        guber_engine* order[MULTI_MEM_MAX];
        for (int i = 0; i < g; ++i) order[i] = grp[i].s->e;
        std::sort(order, order + g);                               // engine locks in address order (launch_group's rule)
        for (int i = 0; i < g; ++i) { order[i]->mu.lock(); ep_flush_held(order[i]); }
        int rc = 0;
        if (grp[0].s->e->set_device()) rc = fail(GUBER_E_HIP, "hipSetDevice");
        for (int i = 0; i < g && !rc; ++i) {
            guber_engine* e = grp[i].s->e;
            if (e->small_pending) { const int r2 = resolve_small_locked(e->small_pending, true, e); if (r2 < 0) rc = r2; }
            if (!rc) rc = stage_views(grp[i].s, grp[i]);
        }
        if (!rc) rc = launch_stage_group(grp, g);
        for (int i = g - 1; i >= 0; --i) order[i]->mu.unlock();
        if (!rc) enq += (uint32_t)g;
        g = 0;
        return rc;
    };
    // batches of <= 256 requests of engines that share device and stream: ONE k_small_multi, one workgroup per batch
    guber_stage* sgrp[SMALL_MULTI_MAX]; int sg = 0;
    auto flush_small = [&]() -> int {
        if (!sg) return 0;
        guber_engine* order[SMALL_MULTI_MAX];
        for (int i = 0; i < sg; ++i) order[i] = sgrp[i]->e;
        std::sort(order, order + sg);
        for (int i = 0; i < sg; ++i) order[i]->mu.lock();
        int rc = 0, planned = 0;
        MultiSmall MS{};
        guber_engine* e0 = sgrp[0]->e;
        if (e0->set_device()) rc = fail(GUBER_E_HIP, "hipSetDevice");
        for (int i = 0; i < sg && !rc; ++i) {
            guber_stage* s = sgrp[i]; guber_engine* e = s->e;
            if (e->small_pending) { const int r2 = resolve_small_locked(e->small_pending, true, e); if (r2 < 0) { rc = r2; break; } }
            const guber_batch_t& b = s->batch;
            BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                        b.greg_expire, b.greg_duration, b.now_ms};
            rc = small_prelude(e, B);
            if (rc) break;
            s->seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
            s->sout->done = 0;
            MS.sub[planned] = SmallArgs{e->T, B, ResultView{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err},
                                        s->sout, e->touch, s->seq};
            ++planned;
        }
        if (planned) {
            MS.nb = (uint32_t)planned;
            if (planned == 1) hipLaunchKernelGGL(k_small, dim3(1), dim3(FT), 0, e0->stream, MS.sub[0].T, MS.sub[0].B, MS.sub[0].R, MS.sub[0].out, MS.sub[0].seq, MS.sub[0].touch);
            else hipLaunchKernelGGL(k_small_multi, dim3(planned), dim3(FT), 0, e0->stream, MS);
            if (hipGetLastError() != hipSuccess && !rc) rc = fail(GUBER_E_HIP, "kernel launch");
            for (int i = 0; i < planned; ++i) { sgrp[i]->mode = 3; sgrp[i]->e->small_pending = sgrp[i]; }
            enq += (uint32_t)planned;
        }
        for (int i = sg - 1; i >= 0; --i) order[i]->mu.unlock();
        sg = 0;
        return rc;
    };
    int rc = 0;
    for (uint32_t k = 0; k < n && !rc; ++k) {
        guber_stage* s = stages[k];
        guber_engine* e = s->e;
        const guber_batch_t& b = s->batch;
        s->n = b.n; s->now_ms = b.now_ms; s->no_agg = true;
        if (b.n == 0) { s->mode = 0; ++enq; continue; }
        memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);        // the kernels read keys as 8-byte words
        const bool small = b.n <= FT && !e->no_small && !lru_may_bind_unlocked(e, b.n);
        const bool fusable = !small && can_fuse(e, b.n);
        if (small) {
            if (sg && (sg == SMALL_MULTI_MAX || e->stream != sgrp[0]->e->stream || e->device != sgrp[0]->e->device)) rc = flush_small();
            if (rc) break;
            sgrp[sg++] = s;
            continue;
        }
        if (g && (!fusable || g == group_max || e->stream != grp[0].s->e->stream || e->device != grp[0].s->e->device)) rc = flush();
        if (rc) break;
        if (fusable) { grp[g++].s = s; continue; }
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->set_device()) { rc = fail(GUBER_E_HIP, "hipSetDevice"); break; }
        if (e->small_pending) { const int r2 = resolve_small_locked(e->small_pending, true, e); if (r2 < 0) { rc = r2; break; } }
        BatchView B{b.n, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner,
                    b.greg_expire, b.greg_duration, b.now_ms};
        ResultView R{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
        // the radix pipeline (n > 65 536) or a test configuration
        rc = launch_batch(e, B, R, true);
        if (rc) break;
        if (hipEventRecord(s->ev, e->stream) != hipSuccess) { rc = fail(GUBER_E_HIP, "hipEventRecord"); break; }
        s->gev = nullptr; s->mode = 2;
        ++enq;
    }
    if (!rc) rc = flush();
    if (!rc) rc = flush_small();
    if (done) {                                                      // the leading stages (array order) that were enqueued; after an
        uint32_t lead = 0;                                           // error the caller settles the others with guber_stage_wait
        while (lead < n && (stages[lead]->mode != 0 || stages[lead]->batch.n == 0)) ++lead;
        *done = rc ? lead : n;
    }
    (void)enq;
    return rc;
}

// ---- ONE stage for several engines: the device-level stage of a pool.  Callers fill it in arrival order and say, per request,
// which engine it belongs to and which place it has in that engine's share (guber_stage_dest); the copy kernel scatters the
// request columns into HBM so that every share is contiguous (k_stage_in_routed), the shares then run as the batches of ONE
// k_front_multi_mem + ONE k_eval2_multi_mem, and a last launch takes the answers back to the callers' slots.  Four launches
// and one event for a whole generation, whatever the number of shards; nothing on the host is proportional to the requests.
extern "C" uint32_t* guber_stage_dest(guber_stage_t* s) { return s ? s->h_dest : nullptr; }
// bytes of a routed stage's HBM mirror before the key bytes (guber_stage_submit_routed lays the request and answer columns out there)
static size_t routed_mirror_fixed(size_t cap) {
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    return 3 * col(cap * 4 + 4) + 5 * col(cap * 8) + col(cap * 4) + 2 * col(cap) + 3 * col(cap * 8) + 2 * col(cap);
}
// The routing of a front stage done by the device (k_route_count + k_route_dest): the callers wrote their requests in arrival
// order and nothing else; afterwards guber_stage_dest(s) holds what they would have written and *counts the shares' sizes —
// exactly the inputs of guber_stage_submit_routed.  The rule is the placement's (guber_placement_export); it is copied to the
// device when given (NULL = the one given last).  Never waits for the GPU except when a rule is uploaded (a placement change).
extern "C" int guber_stage_route(guber_stage_t* s, const guber_route_rule_t* rule, uint32_t n_engines) {
    if (!s) return fail(GUBER_E_INVALID_ARG, "null stage");
    if (n_engines == 0 || n_engines > (uint32_t)MULTI_MEM_MAX) return fail(GUBER_E_INVALID_ARG, "1 .. 16 engines per routed stage");
    if (s->mode || s->route_pending) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
    const guber_batch_t& b = s->batch;
    if (b.n > s->max_n || b.n > 65536u || (b.n && b.key_off[b.n] > s->key_cap)) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled (a routed stage holds at most 65 536 requests)");
    if (!b.behavior) return fail(GUBER_E_INVALID_ARG, "a routed stage carries every request column");
    guber_engine* e = s->e;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    if (rule) {
        if (rule->n_shards == 0 || rule->per == 0 || !rule->table || rule->n_shards > 4096 || (rule->ex_cells & (rule->ex_cells - 1)) ||
            (rule->ex_n && (!rule->ex_hash || !rule->ex_shard || rule->ex_n >= rule->ex_cells)))
            return fail(GUBER_E_INVALID_ARG, "malformed route rule");
        const size_t slots = (size_t)rule->n_shards * rule->per, cells = rule->ex_cells ? rule->ex_cells : 1;
        if (e->d_rt_table.ensure(slots) || e->d_rt_exh.ensure(cells) || e->d_rt_exs.ensure(cells)) return GUBER_E_NOMEM;
        hipError_t he = hipStreamSynchronize(e->stream);                // (launches still reading the previous rule)
        if (he == hipSuccess) he = hipMemcpy(e->d_rt_table.p, rule->table, slots * 2, hipMemcpyHostToDevice);
        if (he == hipSuccess && rule->ex_n) he = hipMemcpy(e->d_rt_exh.p, rule->ex_hash, cells * 8, hipMemcpyHostToDevice);
        if (he == hipSuccess && rule->ex_n) he = hipMemcpy(e->d_rt_exs.p, rule->ex_shard, cells * 2, hipMemcpyHostToDevice);
        if (he != hipSuccess) { e->have_rule = false; return fail(GUBER_E_HIP, "guber_stage_route: rule upload", he); }
        e->rule = RouteRule{rule->n_shards, rule->per, rule->ex_cells, rule->ex_n, rule->global_engine, rule->step, rule->inv_step, rule->inv_sub,
                            e->d_rt_table.p, (const unsigned long long*)e->d_rt_exh.p, e->d_rt_exs.p};
        e->have_rule = true;
    }
    if (!e->have_rule) return fail(GUBER_E_INVALID_ARG, "guber_stage_route: no rule given yet");
    s->route_engines = n_engines;
    for (uint32_t j = 0; j < (uint32_t)MULTI_MEM_MAX; ++j) s->h_route[j] = 0;
    if (b.n == 0) { s->route_pending = false; return GUBER_OK; }
    const uint32_t tiles = (b.n + 255u) / 256u;
    const size_t tab = (size_t)256 * MULTI_MEM_MAX * 4;
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    const size_t cap = s->max_n, off_bytes = col(cap * 4 + 4), beh_bytes = col(cap * 4);
    const bool fresh = s->d_route.p == nullptr;
    if (s->d_route.ensure(2 * tab + 64 + col(cap) + off_bytes + beh_bytes) || s->dmem.ensure(routed_mirror_fixed(cap) + col((size_t)s->key_cap + 64))) return GUBER_E_NOMEM;
    if (fresh && hipMemsetAsync(s->d_route.p + 2 * tab, 0, 64, e->stream) != hipSuccess) return fail(GUBER_E_HIP, "hipMemsetAsync");
    memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);            // the kernels read keys as 8-byte words
    uint8_t* d_keys = s->dmem.p + routed_mirror_fixed(cap);           // (where guber_stage_submit_routed expects them)
    uint32_t* d_off = (uint32_t*)(s->d_route.p + 2 * tab + 64 + col(cap)); uint32_t* d_beh = (uint32_t*)((uint8_t*)d_off + off_bytes);
    RouteIn I{};
    I.src[0] = (const uint4*)b.key_bytes; I.dst[0] = (uint4*)d_keys; I.n16[0] = (uint32_t)(((size_t)b.key_off[b.n] + 16 + 15) / 16);
    I.src[1] = (const uint4*)b.key_off; I.dst[1] = (uint4*)d_off; I.n16[1] = (uint32_t)(((size_t)b.n * 4 + 4 + 15) / 16);
    I.src[2] = (const uint4*)b.behavior; I.dst[2] = (uint4*)d_beh; I.n16[2] = (uint32_t)(((size_t)b.n * 4 + 15) / 16);
    for (int k = 0; k < 3; ++k) I.nb[k] = std::max<uint32_t>(1u, std::min<uint32_t>(256u, (I.n16[k] + 1023) / 1024));
    hipLaunchKernelGGL(k_route_in, dim3(I.nb[0] + I.nb[1] + I.nb[2]), dim3(256), 0, e->stream, I);
    RouteArgs A{};
    A.n = b.n; A.n_engines = n_engines; A.max_key = e->max_key; A.seq = ++s->route_seq ? s->route_seq : ++s->route_seq;
    A.key_bytes = d_keys; A.key_off = d_off; A.behavior = d_beh;
    A.tile_cnt = (uint32_t*)s->d_route.p; A.tile_base = (uint32_t*)(s->d_route.p + tab); A.ticket = (uint32_t*)(s->d_route.p + 2 * tab);
    A.eng = s->d_route.p + 2 * tab + 64;
    A.dest = s->h_dest; A.counts = s->h_route; A.done = (unsigned int*)(s->h_route + MULTI_MEM_MAX);
    A.R = e->rule;
    hipLaunchKernelGGL(k_route_count, dim3(tiles), dim3(256), 0, e->stream, A);
    hipLaunchKernelGGL(k_route_dest, dim3(tiles), dim3(256), 0, e->stream, A);
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    s->route_pending = true; s->keys_resident = true;
    return GUBER_OK;
}
// 1 = the shares' sizes are in counts[0 .. n_engines) (guber_stage_dest is complete by the time anything enqueued later on the
// engines' stream runs: guber_stage_submit_routed may follow at once), 0 = still running
extern "C" int guber_stage_route_poll(guber_stage_t* s, uint32_t* counts) {
    if (!s || !counts) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (s->route_pending) {
        if (__atomic_load_n((volatile unsigned int*)(s->h_route + MULTI_MEM_MAX), __ATOMIC_ACQUIRE) != s->route_seq) return 0;
        s->route_pending = false;
    }
    for (uint32_t j = 0; j < s->route_engines; ++j) counts[j] = s->h_route[j];
    return 1;
}
extern "C" int guber_stage_submit_routed(guber_stage_t* s, guber_engine_t* const* engines, uint32_t n_engines, const uint32_t* counts) {
    if (!s || !engines || !counts) return fail(GUBER_E_INVALID_ARG, "null argument");
    if (n_engines == 0 || n_engines > (uint32_t)MULTI_MEM_MAX) return fail(GUBER_E_INVALID_ARG, "1 .. 16 engines per routed stage");
    if (s->mode) return fail(GUBER_E_INVALID_ARG, "stage already in flight");
    const guber_batch_t& b = s->batch;
    const bool keys_there = s->keys_resident;                        // (guber_stage_route brought this batch's key bytes to the HBM mirror: same bytes, same place)
    s->keys_resident = false;
    s->n = b.n; s->now_ms = b.now_ms; s->no_agg = true; s->routed.clear();
    if (b.n > s->max_n || (b.n && b.key_off[b.n] > s->key_cap)) return fail(GUBER_E_BATCH_TOO_LARGE, "stage overfilled");
    if (b.greg_expire || b.greg_duration) return fail(GUBER_E_INVALID_ARG, "a routed stage takes its calendar intervals from the device");
    if (!b.burst || !b.created_at || !b.behavior || !b.algorithm || !b.is_owner) return fail(GUBER_E_INVALID_ARG, "a routed stage carries every request column");
    uint64_t total = 0; bool own = false;
    for (uint32_t j = 0; j < n_engines; ++j) {
        guber_engine* e = engines[j];
        if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
        if (e->device != s->e->device || e->stream != s->e->stream) return fail(GUBER_E_INVALID_ARG, "the engines of a routed stage share device and stream");
        for (uint32_t q = 0; q < j; ++q) if (engines[q] == e) return fail(GUBER_E_INVALID_ARG, "an engine twice in one routed stage");
        if (counts[j] && !fits_fused(e, counts[j])) return fail(GUBER_E_BATCH_TOO_LARGE, "an engine's share is larger than its two-launch pipeline takes");
        own = own || e == s->e;
        total += counts[j];
    }
    if (!own) return fail(GUBER_E_INVALID_ARG, "the stage's own engine is one of the engines");
    if (total != b.n) return fail(GUBER_E_INVALID_ARG, "the shares do not add up to the batch");
    if (b.n == 0) { s->mode = 0; return GUBER_OK; }
    memset((uint8_t*)b.key_bytes + b.key_off[b.n], 0, 16);            // the kernels read keys as 8-byte words
    guber_engine* order[MULTI_MEM_MAX];
    for (uint32_t j = 0; j < n_engines; ++j) order[j] = engines[j];
    std::sort(order, order + n_engines);                             // engine locks in address order (launch_group's rule)
    for (uint32_t j = 0; j < n_engines; ++j) { order[j]->mu.lock(); ep_flush_held(order[j]); }
    struct Unlock { guber_engine** o; uint32_t n; ~Unlock() { for (uint32_t j = n; j-- > 0;) o[j]->mu.unlock(); } } unlock{order, n_engines};
    guber_engine* e0 = s->e;
    if (e0->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    for (uint32_t j = 0; j < n_engines; ++j)
        if (engines[j]->small_pending) { const int r2 = resolve_small_locked(engines[j]->small_pending, true, engines[j]); if (r2 < 0) return r2; }
    if (b.n <= FT) {                                                 // a handful of requests: ONE launch, a workgroup per share, in place
        bool small_ok = true;
        for (uint32_t j = 0; j < n_engines; ++j) small_ok = small_ok && !engines[j]->no_small && !(counts[j] && lru_may_bind(engines[j], counts[j]));
        if (small_ok) {
            MultiSmallRouted MS{};
            s->parts.clear();
            BatchView BH{0, 0, b.key_bytes, b.key_off, b.hits, b.limit, b.duration, b.burst, b.created_at, b.algorithm, b.behavior, b.is_owner, nullptr, nullptr, b.now_ms};
            for (uint32_t j = 0; j < n_engines; ++j) {
                if (!counts[j]) continue;
                guber_engine* e = engines[j];
                BatchView Bj = BH; Bj.n = counts[j];
                const int rc = small_prelude(e, Bj);
                if (rc) { s->parts.clear(); return rc; }             // (nothing has been launched; the stage stays idle)
                const uint32_t seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
                SmallOut* out = (SmallOut*)(s->h_parts_out + 64 * s->parts.size());
                out->done = 0;
                MS.sub[s->parts.size()] = SmallRoutedSub{e->T, out, e->touch, seq, counts[j], j};
                s->parts.push_back(guber_stage::RoutedPart{e, j, counts[j], seq, out, true, 0});
            }
            MS.nb = (uint32_t)s->parts.size(); MS.n_total = b.n; MS.dest = s->h_dest; MS.B = BH;
            MS.R = ResultView{s->result.status, s->result.limit, s->result.remaining, s->result.reset_time, s->result.err};
            hipLaunchKernelGGL(k_small_routed, dim3(MS.nb), dim3(FT), 0, e0->stream, MS);
            if (hipGetLastError() != hipSuccess) { s->parts.clear(); return fail(GUBER_E_HIP, "kernel launch"); }
            for (auto& part : s->parts) part.e->small_pending = s;
            s->routed.assign(engines, engines + n_engines);
            s->gev = nullptr; s->mode = 4;
            return GUBER_OK;
        }
    }
    // the HBM mirror: every fixed-width column for max_n requests (each 64-byte aligned), where the request came from, the keys
    const size_t n = b.n, cap = s->max_n;
    auto col = [](size_t bytes) { return (bytes + 63) & ~(size_t)63; };
    const size_t fixed = routed_mirror_fixed(cap);
    if (s->dmem.ensure(fixed + col((size_t)s->key_cap + 64)) || e0->d_margs.ensure(sizeof(MultiArgsMem))) return GUBER_E_NOMEM;

    uint8_t* p = s->dmem.p;
    RoutedIn A{};
    A.d_key_off = (uint32_t*)p; p += col(cap * 4 + 4); A.d_key_len = (uint32_t*)p; p += col(cap * 4 + 4); A.d_fwd = (uint32_t*)p; p += col(cap * 4 + 4);
    A.d_hits = (int64_t*)p; p += col(cap * 8); A.d_limit = (int64_t*)p; p += col(cap * 8); A.d_duration = (int64_t*)p; p += col(cap * 8);
    A.d_burst = (int64_t*)p; p += col(cap * 8); A.d_created_at = (int64_t*)p; p += col(cap * 8);
    A.d_behavior = (uint32_t*)p; p += col(cap * 4); A.d_algorithm = p; p += col(cap); A.d_is_owner = p; p += col(cap);
    RoutedOut O{};                                                   // the answers: HBM in the shares' order, then home in arrival order
    O.n = (uint32_t)n; O.fwd = A.d_fwd;
    int64_t* o_limit = (int64_t*)p; p += col(cap * 8); int64_t* o_remaining = (int64_t*)p; p += col(cap * 8); int64_t* o_reset = (int64_t*)p; p += col(cap * 8);
    uint8_t* o_status = p; p += col(cap); uint8_t* o_err = p; p += col(cap);
    O.d_status = o_status; O.d_err = o_err; O.d_limit = o_limit; O.d_remaining = o_remaining; O.d_reset_time = o_reset;
    O.status = s->result.status; O.err = s->result.err; O.limit = s->result.limit; O.remaining = s->result.remaining; O.reset_time = s->result.reset_time;
    uint8_t* d_keys = p;
    A.n = (uint32_t)n; A.dest = s->h_dest;
    A.key_off = b.key_off; A.hits = b.hits; A.limit = b.limit; A.duration = b.duration; A.burst = b.burst; A.created_at = b.created_at;
    A.behavior = b.behavior; A.algorithm = b.algorithm; A.is_owner = b.is_owner;
    A.key_src = (const uint4*)b.key_bytes; A.key_dst = (uint4*)d_keys; A.key_n16 = (uint32_t)(((size_t)b.key_off[b.n] + 16 + 15) / 16);
    MultiArgsMem* HA = s->h_margs; MultiArgsMem* DA = (MultiArgsMem*)e0->d_margs.p;
    uint32_t tiles = 0, base = 0; int planned = 0;
    guber_engine* took[MULTI_MEM_MAX]; uint32_t took_n[MULTI_MEM_MAX];
    // a share that may overflow its engine's cache needs the eviction pre-pass (launch_batch), which reads the share's keys: then the
    // shares are brought to HBM first and evaluated engine by engine
    bool exact = false;
    for (uint32_t j = 0; j < n_engines; ++j) exact = exact || (counts[j] && lru_may_bind(engines[j], counts[j]));
    BatchView XB[MULTI_MEM_MAX]; ResultView XR[MULTI_MEM_MAX];
    for (uint32_t j = 0; j < n_engines; ++j) {
        A.base[j] = base;
        const uint32_t nj = counts[j];
        if (!nj) continue;
        guber_engine* e = engines[j];
        BatchView B{nj, 0, d_keys, A.d_key_off + base, A.d_hits + base, A.d_limit + base, A.d_duration + base, A.d_burst + base, A.d_created_at + base,
                    A.d_algorithm + base, A.d_behavior + base, A.d_is_owner + base, nullptr, nullptr, b.now_ms, 0, A.d_key_len + base};
        ResultView R{o_status + base, o_limit + base, o_remaining + base, o_reset + base, o_err + base};
        if (exact) { XB[planned] = B; XR[planned] = R; took[planned] = e; took_n[planned] = nj; ++planned; base += nj; continue; }
        Work W; FastPlan FP;
        int rc = batch_prelude(e, B, W);
        if (!rc) rc = plan_fast(e, B, false, W, FP);
        if (rc) return rc;                                           // (nothing has been launched; the stage stays idle)
        tiles += FP.ftiles;
        HA->F.end_tile[planned] = HA->E.end_tile[planned] = tiles;
        HA->F.sub[planned] = FrontArgs{e->T, FP.B2, FP.W};
        HA->E.sub[planned] = EvalArgs{e->T, FP.B3, R, FP.W};
        took[planned] = e; took_n[planned] = nj;
        ++planned;
        base += nj;
    }
    HA->F.nb = HA->E.nb = (uint32_t)planned;
    A.arg_src = (const uint4*)HA; A.arg_dst = (uint4*)DA;
    A.arg_off16[0] = 0; A.arg_n16[0] = (uint32_t)((offsetof(MultiFrontMem, sub) + (size_t)planned * sizeof(FrontArgs) + 15) / 16);
    A.arg_off16[1] = (uint32_t)(offsetof(MultiArgsMem, E) / 16); A.arg_n16[1] = (uint32_t)((offsetof(MultiEvalMem, sub) + (size_t)planned * sizeof(EvalArgs) + 15) / 16);
    A.nb_req = (uint32_t)((n + 255) / 256);
    A.nb_key = keys_there ? 0u : std::max<uint32_t>(1u, std::min<uint32_t>(256u, (A.key_n16 + 1023) / 1024));
    A.nb_arg = 4;
    hipStream_t st = e0->stream;
    hipLaunchKernelGGL(k_stage_in_routed, dim3(A.nb_req + A.nb_key + A.nb_arg), dim3(256), 0, st, A);
    if (exact) {
        for (int i = 0; i < planned; ++i) { const int rc = launch_batch(took[i], XB[i], XR[i]); if (rc) return rc; }
        hipLaunchKernelGGL(k_stage_out_routed, dim3(A.nb_req), dim3(256), 0, st, O);
        if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
        if (hipEventRecord(s->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
        s->routed.assign(engines, engines + n_engines);
        s->gev = nullptr; s->mode = 2;
        return GUBER_OK;
    }
    e0->span_begin(KT_FRONT_MULTI, n);
    hipLaunchKernelGGL(k_front_multi_mem, dim3(tiles), dim3(FT), 0, st, (const MultiFrontMem*)&DA->F);
    e0->span_end();
    e0->span_begin(KT_EVAL2_MULTI, n);
    hipLaunchKernelGGL(k_eval2_multi_mem, dim3(tiles), dim3(256), 0, st, (const MultiEvalMem*)&DA->E);
    e0->span_end();
    hipLaunchKernelGGL(k_stage_out_routed, dim3(A.nb_req), dim3(256), 0, st, O);
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    for (int i = 0; i < planned; ++i) { finish_fast(took[i], took_n[i]); if (planned > 1) took[i]->fused_batches++; }
    if (hipEventRecord(s->ev, st) != hipSuccess) return fail(GUBER_E_HIP, "hipEventRecord");
    s->routed.assign(engines, engines + n_engines);
    s->gev = nullptr; s->mode = 2;
    return GUBER_OK;
}
