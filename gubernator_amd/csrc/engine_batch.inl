// engine_batch.inl — part of guber_engine.hip's translation unit (included there, in this order; not a header of its own):
// the bounded cache's pre-pass (lru_admit), a batch's prelude and plans, launch_batch and the device-pointer entry points of ONE engine.
// ---- the bounded cache's exact victim order (guber_kernels_lru.h) --------------------------------------------------------------
// May this call make the cache longer than cache_size?  size_upper is the host's upper bound of the live items (every request
// might create one); lru_admit reads the exact figure when it matters.
// When the bound says yes, the bound is first brought up to date: it counts every request in flight as a new item, so the host
// looks at the counter snapshots that ride on the batches (maintain) and, as long as one is on its way, waits for the GPU to get
// there — a wait for PROGRESS, not a drain: the queue stays as deep as the cache's headroom allows.  Only when nothing is left to
// wait for is the answer yes (the pre-pass then synchronises and sees the exact figure).  Engine mutex held.
static bool lru_may_bind(guber_engine* e, uint64_t n) {
    if (e->size_upper + n <= e->cache_size) return false;
    bool waited = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (rb_fold_newest(e) && e->size_upper + n <= e->cache_size) { e->settle_waits += waited; return false; }
        if (!rb_any_launched(e)) break;
        waited = true;
        if ((spins & 0xff) == 0xff) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
            std::this_thread::yield();                           // (the engine mutex is held: whoever else needs this CPU gets it)
        }
        __builtin_ia32_pause();
    }
    e->settle_waits += waited;
    return true;
}
static bool lru_may_bind_unlocked(guber_engine* e, uint64_t n) { std::lock_guard<std::mutex> lk(e->mu); return lru_may_bind(e, n); }
static LruKeys lru_keys_of(const BatchView& B) {
    LruKeys K{};
    K.bytes = B.key_bytes; K.algorithm = B.algorithm; K.key_stride = B.key_stride;
    K.behavior = B.behavior; K.duration = B.duration; K.greg_duration = (B.greg_expire && B.greg_duration) ? B.greg_duration : nullptr;
    if (!B.key_stride) { K.off_p = (const uint8_t*)B.key_off; K.off_stride = 4; }
    if (B.key_stride || B.key_len) { K.len_p = (const uint8_t*)B.key_len; K.len_stride = 4; }
    return K;
}
// The tail list: every live item's (stamp, slot), sorted by stamp — one table scan and one radix sort, then good for as many
// batches as it has valid entries left (an entry is valid while its bucket still carries that stamp).
static int lru_rebuild(guber_engine* e) {
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    const uint64_t live = (uint64_t)std::max<long long>(e->last_ctr.size, 0);
    const uint64_t cap = live + 64;
    if (e->lru_ctl.ensure(1) || e->lru_hctl.ensure(1) || e->lru_cnt.ensure(1) || e->lru_tstamp.ensure(cap) || e->lru_tstamp_in.ensure(cap) ||
        e->lru_tslot.ensure(cap) || e->lru_tslot_in.ensure(cap)) return GUBER_E_NOMEM;
    hipStream_t st = e->stream;
    HIPCHK(hipMemsetAsync(e->lru_cnt.p, 0, 8, st));
    hipLaunchKernelGGL(k_lru_gather, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, st, e->T, e->slots, e->lru_tstamp_in.p, e->lru_tslot_in.p, cap, e->lru_cnt.p);
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, e->lru_cnt.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (cnt > cap) return fail(GUBER_E_HIP, "the table holds more live items than its counters say");
    if (cnt) {
        size_t tmp = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, e->lru_tstamp_in.p, e->lru_tstamp.p, e->lru_tslot_in.p, e->lru_tslot.p, (int)cnt, 0, 53, st));
        if (e->lru_sort_tmp.ensure(tmp + 16)) return GUBER_E_NOMEM;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(e->lru_sort_tmp.p, tmp, e->lru_tstamp_in.p, e->lru_tstamp.p, e->lru_tslot_in.p, e->lru_tslot.p, (int)cnt, 0, 53, st));
    }
    LruCtl c{}; c.cursor = 0; c.tail_n = cnt;
    HIPCHK(hipMemcpyAsync(e->lru_ctl.p, &c, sizeof(c), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    e->lru_tail_ok = true; e->lru_rebuilds++;
    return 0;
}
// The pre-pass of a call that may overflow the cache: n requests (or n = 0: only bring the cache down to cache_size).  On return
// *status is LRU_NONE / LRU_APPLIED (the buckets that leave are absent, the counters adjusted: evaluate the batch) or LRU_CUT
// (nothing done: the batch is larger than the cache and evictions are due — the caller evaluates it in pieces of cache_size).
static int lru_admit(guber_engine* e, const LruKeys& K, uint32_t n, int64_t now_ms, uint32_t* status) {
    hipStream_t st = e->stream;
    if (e->lru_ctl.ensure(1) || e->lru_hctl.ensure(1)) return GUBER_E_NOMEM;
    if (!e->lru_tstamp.p) {                                          // first use: an empty list (the first check asks for a real one)
        LruCtl c{};
        HIPCHK(hipMemcpyAsync(e->lru_ctl.p, &c, sizeof(c), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        if (e->lru_tstamp.ensure(16) || e->lru_tslot.ensure(16)) return GUBER_E_NOMEM;
        e->lru_tail_ok = true;                                       // (valid and empty)
    }
    uint32_t cells = 1024; while (cells < 2 * (uint64_t)n) cells <<= 1;
    e->lru_admits++;
    uint64_t w_len = std::max<uint64_t>(2 * (uint64_t)n, 4096);
    for (int round = 0; round < 64; ++round) {
        if (!e->lru_tail_ok) { const int rc = lru_rebuild(e); if (rc) return rc; }
        const uint64_t live_cap = e->lru_tstamp.cap;
        if (w_len > live_cap) w_len = live_cap;
        const uint32_t W = (uint32_t)std::min<uint64_t>(w_len, 1u << 30), wblocks = (W + 255) / 256;
        // scratch: u64 [cells gid | n rstamp | W zstamp], u32 [cells gfirst | cells gfirst_ok | cells gfirst_reset | n rfirst | n rslot | W zslot | W zwidx | n qfirst | n qrank | n qslot |
        //               wblocks + 1 blockcnt | n + 1 new_before | W + 1 touched_before | 4 n_risk], u8 [n + 1 isnew_at | W wflag | W ztouched]
        const size_t nn = (size_t)n + 1;
        if (e->lru_u64.ensure((size_t)cells + nn + W + 8) || e->lru_u32.ensure(3 * (size_t)cells + 6 * nn + 3 * ((size_t)W + 1) + wblocks + 16) ||
            e->lru_u8.ensure(nn + 2 * ((size_t)W + 1) + 64)) return GUBER_E_NOMEM;
        unsigned long long* p64 = e->lru_u64.p; uint32_t* p32 = e->lru_u32.p; uint8_t* p8 = e->lru_u8.p;
        LruGroups G{p64, p32, p32 + cells, p32 + 2 * (size_t)cells, cells - 1}; p64 += cells; p32 += 3 * (size_t)cells;
        LruRes R{p32, p32 + nn, p64}; p32 += 2 * nn; p64 += nn;
        LruWin Z{p64, p32, p32 + W + 1}; p64 += W; p32 += 2 * ((size_t)W + 1);
        LruRisk Q{p32, p32 + nn, p32 + 2 * nn}; p32 += 3 * nn;
        uint32_t* blockcnt = p32; p32 += wblocks + 1;
        uint32_t* new_before = p32; p32 += nn;
        uint32_t* touched_before = p32; p32 += (size_t)W + 1;
        uint32_t* n_risk = p32;
        uint8_t* isnew_at = p8; uint8_t* wflag = p8 + nn; uint8_t* ztouched = wflag + W + 1;
        LruCtl* C = e->lru_ctl.p;
        hipLaunchKernelGGL(k_lru_begin, dim3(1), dim3(256), 0, st, e->T, C, e->n_bctr);
        if (n) {
            HIPCHK(hipMemsetAsync(G.id, 0xff, (size_t)cells * 8, st));
            HIPCHK(hipMemsetAsync(G.first, 0xff, (size_t)cells * 12, st));                  // (first, first_ok and first_reset)
            HIPCHK(hipMemsetAsync(isnew_at, 0, nn, st));
            hipLaunchKernelGGL(k_lru_probe, dim3((n + 255) / 256), dim3(256), 0, st, e->T, K, n, G);
            hipLaunchKernelGGL(k_lru_keys, dim3(cells / 256), dim3(256), 0, st, e->T, G, C, isnew_at, R);
        }
        HIPCHK(hipMemsetAsync(ztouched, 0, (size_t)W + 1, st));
        HIPCHK(hipMemsetAsync(n_risk, 0, 4, st));
        if (W) {
            hipLaunchKernelGGL(k_lru_win_flag, dim3(wblocks), dim3(256), 0, st, e->T, e->lru_tstamp.p, e->lru_tslot.p, C, W, wflag, blockcnt);
            hipLaunchKernelGGL(k_lru_scan_u32, dim3(1), dim3(1024), 0, st, blockcnt, wblocks, &C->win_valid);
            hipLaunchKernelGGL(k_lru_win_emit, dim3(wblocks), dim3(256), 0, st, e->lru_tstamp.p, e->lru_tslot.p, C, W, wflag, blockcnt, Z);
        }
        hipLaunchKernelGGL(k_lru_check, dim3(1), dim3(1), 0, st, C, W, n, e->cache_size);
        if (W) {
            if (n) {
                hipLaunchKernelGGL(k_lru_risk, dim3((n + 255) / 256), dim3(256), 0, st, C, R, Z, ztouched, Q, n_risk);
                hipLaunchKernelGGL(k_lru_scan_u8, dim3(1), dim3(1024), 0, st, isnew_at, n, new_before);
            }
            hipLaunchKernelGGL(k_lru_scan_u8, dim3(1), dim3(1024), 0, st, ztouched, W, touched_before);
            if (n) hipLaunchKernelGGL(k_lru_decide, dim3((n + 255) / 256), dim3(256), 0, st, e->T, C, e->cache_size, Q, n_risk, new_before, now_ms);
            hipLaunchKernelGGL(k_lru_evict, dim3(wblocks), dim3(256), 0, st, e->T, C, Z, ztouched, touched_before, now_ms);
            hipLaunchKernelGGL(k_lru_end, dim3(1), dim3(1), 0, st, e->T, C, Z);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(e->lru_hctl.p, C, sizeof(LruCtl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const LruCtl& c = *e->lru_hctl.p;
        e->lru_passes++;
        if (c.status == LRU_NONE || c.status == LRU_APPLIED) {
            const long long left = c.len0 - (long long)c.evicted;
            e->size_upper = (uint64_t)std::max<long long>(left, 0);
            rb_disarm_all(e);                                        // (the stream has drained: every snapshot on its way is older news)
            e->last_ctr.size = left; e->last_ctr.evictions += c.unexpired;
            if (c.status == LRU_APPLIED) e->lru_applied++;
            *status = c.status;
            return 0;
        }
        if (c.status == LRU_CUT) { e->lru_cuts++; *status = LRU_CUT; return 0; }
        if (c.status == LRU_SPLIT) { e->lru_cuts++; e->lru_split_at = c.split_at; *status = LRU_SPLIT; return 0; }
        if (c.status == LRU_MORE) { w_len *= 4; continue; }
        if (c.status == LRU_REBUILD) { e->lru_tail_ok = false; w_len = std::max<uint64_t>(w_len, 2 * (uint64_t)n + c.zone); continue; }
        return fail(GUBER_E_HIP, "the eviction pre-pass left no verdict");
    }
    return fail(GUBER_E_HIP, "the eviction pre-pass did not converge");
}

static int batch_prelude(guber_engine* e, const BatchView& B, Work& W, bool* defer_hard = nullptr) {
    const uint32_t n = B.n;
    if (n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    if (defer_hard && e->epoch + 1 >= 0x7fffffffu) { *defer_hard = true; return 0; }   // (the wrap below enqueues a launch)
    if (B.now_ms > e->clock_ms) e->clock_ms = B.now_ms;
    // Bounded cache and directory load.  size_upper / tags_upper are host-side upper bounds (every request might create a
    // new item); only when one crosses its limit are the real counters read back, the least recently used items evicted
    // (lrucache.go:98-100) and, if the directory is above its load limit, the table rebuilt without its dead entries.  A
    // batch that still finds no room gets per-item GUBER_ITEM_E_TABLE_FULL from the bounded probe, for NEW keys only.
    {
        const int rc = maintain(e, n, B.now_ms, takes_fast_path(e, n), defer_hard);
        if (rc) return rc;
        if (defer_hard && *defer_hard) return 0;             // (nothing has been done: the caller comes back)
    }
    note_enqueued(e, n);
    if (++e->epoch >= 0x7fffffffu) {   // 31-bit epoch wrapped: drop all dense-id claims
        hipLaunchKernelGGL(k_clear_claims, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, e->stream, e->T, e->slots);
        e->epoch = 1;
    }
    W = e->W;
    W.epoch = e->epoch;
    W.touch = take_stamps(e, n);
    W.tiles = (n + TILE - 1) / TILE;
    return 0;
}

static int plan_fast(guber_engine* e, const BatchView& B, bool host_resident, Work& W, FastPlan& P) {
    const uint32_t n = B.n;
    BatchView B2 = B;
    B2.n_cap = e->fast_cap;
    W.careful = (e->careful || e->always_careful) ? 1u : 0u;
    W.snap_seq = 0;
    if (e->rb_ride >= 0) attach_counter_readback(e, W);
    if (++e->fast_epoch16 > 0xffffu) {   // 16-bit claim epoch wrapped: forget every cell
        HIPCHK(hipMemsetAsync(e->w_claims.p, 0, (size_t)e->claims_cells * 8, e->stream));
        HIPCHK(hipMemset2DAsync(&e->w_srec.p[0].flags, sizeof(SegRec), 0, sizeof(unsigned long long), e->fast_cap, e->stream));   // epoch-tagged flag words
        e->fast_epoch16 = 1;
    }
    W.epoch16 = e->fast_epoch16;
    {   // the batch's share of the claim table: 4 cells per request (k_eval2 zeroes exactly that part again)
        uint32_t cells = 1024;
        while (cells < 4 * n && cells < e->claims_cells) cells <<= 1;
        W.cmask = cells - 1;
    }
    W.parity = e->fast_batches & 1u;
    W.did = e->w_did2.p + (size_t)W.parity * e->fast_cap;
    W.did_prev = e->w_did2.p + (size_t)(W.parity ^ 1u) * e->fast_cap;
    W.clear_n = e->fast_prev_n;
#ifdef GUBER_PHASE_TIMING
    W.dbg = e->dbg.p;
#endif
    BatchView B3 = B2;                     // what k_eval2 reads
    W.st_hits = nullptr;
    if (host_resident) {
        const size_t c = e->fast_cap;
        if (e->d_stash64.ensure(5 * c) || e->d_stash32.ensure(c) || e->d_stash8.ensure(2 * c)) return GUBER_E_NOMEM;
        int64_t* q = e->d_stash64.p;
        W.st_hits = q; W.st_limit = q + c; W.st_duration = q + 2 * c; W.st_burst = q + 3 * c; W.st_created = q + 4 * c;
        W.st_behavior = e->d_stash32.p; W.st_algorithm = e->d_stash8.p; W.st_owner = e->d_stash8.p + c;
        B3.hits = W.st_hits; B3.limit = W.st_limit; B3.duration = W.st_duration; B3.burst = W.st_burst; B3.created_at = W.st_created;
        B3.behavior = W.st_behavior; B3.algorithm = W.st_algorithm; B3.is_owner = W.st_owner;
    }
    P.B2 = B2; P.B3 = B3; P.W = W; P.ftiles = (n + FT - 1) / FT;
    return 0;
}
static int plan_part(guber_engine* e, const BatchView& B, Work& W, FastPlan& P) {
    BatchView B2 = B;
    B2.n_cap = e->cap256;
    W.careful = 0u;
    W.snap_seq = 0;
    if (e->rb_ride >= 0) attach_counter_readback(e, W);
    // (a GUBER_FUSE_EP engine: packed words and owner count per batch parity — this batch's k_part may run beside the previous
    // batch's k_eval3, k_evalpart_multi)
    W.did = e->w_did3.p + (e->fuse_ep ? (size_t)(e->part_batches & 1) * e->cap256 : 0);
    W.pmslot = e->fuse_ep ? 1u + (uint32_t)(e->part_batches & 1) : 0u;
    W.st_hits = nullptr;
#ifdef GUBER_PHASE_TIMING
    W.dbg = e->dbg.p;
#endif
    P.B2 = B2; P.B3 = B2; P.W = W; P.ftiles = (B.n + FT - 1) / FT;
    return 0;
}
static void finish_fast(guber_engine* e, uint32_t n) {
    e->fast_batches++;
    e->fast_prev_n = n;
    e->batches++;
}

static int launch_batch_inner(guber_engine* e, const BatchView& B, const ResultView& R, bool host_resident);
// requests [pos, pos + len) of a batch as a batch of their own
static BatchView batch_slice(const BatchView& B, uint32_t pos, uint32_t len) {
    BatchView S = B;
    S.n = len;
    if (B.key_stride) S.key_bytes = B.key_bytes + (size_t)pos * B.key_stride; else S.key_off = B.key_off + pos;
    if (B.key_len) S.key_len = B.key_len + pos;
    S.hits = B.hits + pos; S.limit = B.limit + pos; S.duration = B.duration + pos;
    if (B.burst) S.burst = B.burst + pos;
    if (B.created_at) S.created_at = B.created_at + pos;
    if (B.algorithm) S.algorithm = B.algorithm + pos;
    if (B.behavior) S.behavior = B.behavior + pos;
    if (B.is_owner) S.is_owner = B.is_owner + pos;
    if (B.greg_expire) S.greg_expire = B.greg_expire + pos;
    if (B.greg_duration) S.greg_duration = B.greg_duration + pos;
    return S;
}
// One batch through the engine.  A batch that may overflow the cache first goes through the eviction pre-pass (lru_admit: the
// reference evicts in the middle of a stream of requests, lrucache.go:98-100, and the pre-pass reproduces exactly that); a batch
// larger than the cache is then evaluated in pieces of cache_size requests, each with its own pre-pass.
static int launch_batch(guber_engine* e, const BatchView& B, const ResultView& R, bool host_resident = false) {
    if (B.n == 0) return 0;
    if (!lru_may_bind(e, B.n)) return launch_batch_inner(e, B, R, host_resident);
    if (B.n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    uint8_t* const sf0 = e->W.store_flags; Rec* const sa0 = e->W.store_after;
    int rc = 0;
    for (uint32_t pos = 0; pos < B.n && !rc;) {
        uint32_t len = std::min<uint32_t>(B.n - pos, 1u << 20);
        uint32_t st = 0;
        rc = lru_admit(e, lru_keys_of(batch_slice(B, pos, len)), len, B.now_ms, &st);
        if (!rc && st == LRU_CUT) {
            len = (uint32_t)std::min<uint64_t>(len, std::max<uint64_t>(e->cache_size, 1));
            rc = lru_admit(e, lru_keys_of(batch_slice(B, pos, len)), len, B.now_ms, &st);
        }
        // a resident key whose first request cannot insert (guber_kernels_lru.h "ISOLATED"): the requests before it, then it alone, then the rest
        if (!rc && st == LRU_SPLIT) {
            len = e->lru_split_at ? std::min(len, e->lru_split_at) : 1u;
            rc = lru_admit(e, lru_keys_of(batch_slice(B, pos, len)), len, B.now_ms, &st);
            if (!rc && st != LRU_NONE && st != LRU_APPLIED) rc = fail(GUBER_E_HIP, "the eviction pre-pass split a piece twice");
        }
        if (rc) break;
        if (sf0) { e->W.store_flags = sf0 + pos; e->W.store_after = sa0 + pos; }
        rc = launch_batch_inner(e, batch_slice(B, pos, len), ResultView{R.status + pos, R.limit + pos, R.remaining + pos, R.reset_time + pos, R.err + pos}, host_resident);
        pos += len;
    }
    e->W.store_flags = sf0; e->W.store_after = sa0;
    return rc;
}
static int launch_batch_inner(guber_engine* e, const BatchView& B, const ResultView& R, bool host_resident) {
    const uint32_t n = B.n;
    if (n == 0) return 0;
    Work W;
    {
        const int rc = batch_prelude(e, B, W);
        if (rc) return rc;
    }
    const uint32_t tiles = W.tiles;
    if (takes_part_path(e, n, host_resident, false)) {
        FastPlan P;
        {
            const int rc = plan_part(e, B, W, P);
            if (rc) return rc;
        }
        e->span_begin(KT_PART, n);
        hipLaunchKernelGGL(k_part, dim3(P.ftiles), dim3(FT), 0, e->stream, e->T, P.B2, P.W);
        e->span_end();
        e->span_begin(KT_OWN, n);
        hipLaunchKernelGGL(k_own, dim3(PT_PARTS), dim3(256), 0, e->stream, e->T, P.B2, P.W, P.ftiles);
        e->span_end();
        e->span_begin(KT_EVAL3, n);
        hipLaunchKernelGGL(k_eval3, dim3(P.ftiles), dim3(256), 0, e->stream, EvalArgs{e->T, P.B3, R, P.W});
        e->span_end();
        HIPCHK(hipGetLastError());
#ifdef GUBER_PHASE_TIMING
        if (n == e->fast_cap) {   // fold the stamps of full batches (as for the two-launch pipeline below)
            static unsigned long long hb[3 * 2048];
            (void)hipStreamSynchronize(e->stream);
            (void)hipMemcpy(hb, e->dbg.p + 4096, sizeof(hb), hipMemcpyDeviceToHost);
            static const int nst[3] = {6, 8, 4};
            for (int kern = 0; kern < 3; ++kern) {
                const unsigned long long* b = hb + kern * 2048;
                const uint32_t wgs = kern == 1 ? (uint32_t)PT_PARTS : P.ftiles;
                unsigned long long t0 = ~0ull;
                uint32_t ran = 0;                                        // (k_own workgroups beyond the batch's owner count return at once and stamp 0)
                for (uint32_t t = 0; t < wgs; ++t) if (b[t * 8]) { t0 = b[t * 8] < t0 ? b[t * 8] : t0; ran++; }
                for (int k = 0; k < nst[kern] && ran; ++k) {
                    double sum = 0, mx = 0;
                    for (uint32_t t = 0; t < wgs; ++t) { if (!b[t * 8]) continue; const double v = (double)(b[t * 8 + k] - t0) * 0.01; sum += v; mx = v > mx ? v : mx; }
                    e->dbg_avg[2 + kern][k] += sum / ran; e->dbg_max[2 + kern][k] += mx;
                }
            }
            e->dbg_pn++;
        }
#endif
        e->batches++; e->part_batches++;
        return 0;
    }
    if (takes_fast_path(e, n)) {
        // two launches: resolve + in-tile grouping, then evaluation
        FastPlan P;
        {
            const int rc = plan_fast(e, B, host_resident, W, P);
            if (rc) return rc;
        }
        const uint32_t ftiles = P.ftiles;
        e->span_begin(KT_FRONT, n);
        hipLaunchKernelGGL(k_front, dim3(ftiles), dim3(FT), 0, e->stream, e->T, P.B2, P.W);
        e->span_end();
        e->span_begin(KT_EVAL2, n);
        hipLaunchKernelGGL(k_eval2, dim3((n + 255) / 256), dim3(256), 0, e->stream, EvalArgs{e->T, P.B3, R, P.W});
        e->span_end();
        HIPCHK(hipGetLastError());
#ifdef GUBER_PHASE_TIMING
        if (n == e->fast_cap) {   // fold the stamps of full batches: avg and max over workgroups, relative to the first workgroup's entry
            static unsigned long long hb[4096];
            (void)hipStreamSynchronize(e->stream);
            (void)hipMemcpy(hb, e->dbg.p, sizeof(hb), hipMemcpyDeviceToHost);
            for (int kern = 0; kern < 2; ++kern) {
                const int ns = kern ? 5 : 8;
                const unsigned long long* b = hb + kern * 2048;
                unsigned long long t0 = ~0ull;
                for (uint32_t t = 0; t < ftiles; ++t) t0 = b[t * 8] < t0 ? b[t * 8] : t0;
                for (int k = 0; k < ns; ++k) {
                    double sum = 0, mx = 0;
                    for (uint32_t t = 0; t < ftiles; ++t) { const double v = (double)(b[t * 8 + k] - t0) * 0.01; sum += v; mx = v > mx ? v : mx; }
                    e->dbg_avg[kern][k] += sum / ftiles; e->dbg_max[kern][k] += mx;
                }
            }
            e->dbg_n++;
        }
#endif
        finish_fast(e, n);
        return 0;
    }
    int passes = 1;
    while (passes < MAX_PASSES && (1ull << (RADIX_BITS * passes)) < n) passes++;
    e->span_begin(KT_RESOLVE);
    hipLaunchKernelGGL(k_resolve, dim3(tiles), dim3(TILE), 0, e->stream, e->T, B, W);
    e->span_end();
    const uint32_t* kin = nullptr; const uint32_t* vin = nullptr;
    uint32_t* kout = W.keyA; uint32_t* vout = W.valA;
    for (int p = 0; p < passes; ++p) {
        if (p > 0) {
            e->span_begin(KT_HIST);
            hipLaunchKernelGGL(k_hist, dim3(tiles), dim3(TILE), 0, e->stream, W, n, p, kin);
            e->span_end();
        }
        e->span_begin(p == 0 ? KT_SCATTER0 : KT_SCATTER);
        hipLaunchKernelGGL(k_scatter, dim3(tiles), dim3(TILE), 0, e->stream, e->T, B, W, p, p == 0 ? 1 : 0,
                           p == passes - 1 ? 1 : 0, kin, vin, kout, vout);
        e->span_end();
        kin = kout; vin = vout;
        kout = (kout == W.keyA) ? W.keyB : W.keyA; vout = (vout == W.valA) ? W.valB : W.valA;
    }
    e->span_begin(KT_HEADS);
    hipLaunchKernelGGL(k_heads, dim3((n + 255) / 256), dim3(256), 0, e->stream, W, n);
    e->span_end();
    e->span_begin(KT_EVAL);
    hipLaunchKernelGGL(k_eval, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, B, R, W);
    e->span_end();
    HIPCHK(hipGetLastError());
    e->batches++;
    return 0;
}

static int check_batch_args(const guber_batch_t* b, const guber_result_t* r) {
    if (!b || !r) return fail(GUBER_E_INVALID_ARG, "null batch/result");
    if (b->n == 0) return 0;
    if (!b->key_bytes || !b->key_off || !b->hits || !b->limit || !b->duration)
        return fail(GUBER_E_INVALID_ARG, "batch is missing a mandatory array");
    if (!r->status || !r->limit || !r->remaining || !r->reset_time || !r->err)
        return fail(GUBER_E_INVALID_ARG, "result is missing an array");
    return 0;
}

extern "C" int guber_eval_batch_dev(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    int rc = check_batch_args(b, r);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    BatchView B{b->n, 0, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms};
    ResultView R{r->status, r->limit, r->remaining, r->reset_time, r->err};
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
    return launch_batch(e, B, R);
}

extern "C" int guber_eval_batches_dev(guber_engine_t* e, const guber_batch_t* batches, guber_result_t* results, uint32_t count,
                                      uint32_t* done) {
    if (done) *done = 0;
    if (!e || (count && (!batches || !results))) return fail(GUBER_E_INVALID_ARG, "null argument");
    for (uint32_t k = 0; k < count; ++k) {
        const int rc = check_batch_args(&batches[k], &results[k]);
        if (rc) return rc;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    for (uint32_t k = 0; k < count; ++k) {
        const guber_batch_t* b = &batches[k]; guber_result_t* r = &results[k];
        BatchView B{b->n, 0, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                    b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms};
        ResultView R{r->status, r->limit, r->remaining, r->reset_time, r->err};
        r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
        const int rc = launch_batch(e, B, R);
        if (rc) return rc;
        if (done) *done = k + 1;
    }
    return GUBER_OK;
}
