// engine_host.inl — part of guber_engine.hip's translation unit (included there, in this order; not a header of its own):
// the host-pointer entry (stage -> copies -> kernels -> copies back), the one-launch small path.

// Host-pointer evaluation: stage -> H2D -> kernels -> D2H.  `idx` (optional) selects a subset of
// the caller's batch (used to re-submit GUBER_ITEM_E_RETRY items).
static void item_from_rec(const Rec& s, guber_item_t* out);
// the same prelude launch_batch has, for the one-launch path
static int small_prelude(guber_engine* e, const BatchView& B) {
    if (B.now_ms > e->clock_ms) e->clock_ms = B.now_ms;
    const int rc = maintain(e, B.n, B.now_ms);
    if (rc) return rc;
    note_enqueued(e, B.n);
    take_stamps(e, B.n);
    e->batches++; e->small_batches++;
    return 0;
}
static int launch_small(guber_engine* e, const BatchView& B, const ResultView& R, SmallOut* out, uint32_t seq) {
    const int rc = small_prelude(e, B);
    if (rc) return rc;
    hipLaunchKernelGGL(k_small, dim3(1), dim3(FT), 0, e->stream, e->T, B, R, out, seq, e->touch);
    HIPCHK(hipGetLastError());
    return 0;
}

static int eval_host_once(guber_engine* e, const guber_batch_t* b, guber_result_t* r, const uint32_t* idx, uint32_t n,
                          guber_store_events_t* sev = nullptr) {
    const bool has_burst = b->burst, has_created = b->created_at, has_greg = b->greg_expire && b->greg_duration;
    // key bytes of the (sub)batch
    size_t kbytes = 0;
    for (uint32_t j = 0; j < n; ++j) { uint32_t i = idx ? idx[j] : j; kbytes += b->key_off[i + 1] - b->key_off[i]; }
    if (kbytes > 0xfffffff0ull) return fail(GUBER_E_BATCH_TOO_LARGE, "key bytes exceed 4 GiB");
    const size_t n64 = (size_t)n * 7;   // hits limit duration burst created greg_expire greg_duration
    const size_t stage_bytes = (kbytes + 16) + (size_t)(n + 1) * 4 + n64 * 8 + (size_t)n * 4 + (size_t)n * 2 + 64 +
                               (size_t)n * (3 * 8 + 2) + 64 + sizeof(SmallOut);
    const bool zc = e->zero_copy;
    int rc = 0;
    if (zc) rc |= e->z_stage.ensure(stage_bytes + 256);
    else {
        rc |= e->h_stage.ensure(stage_bytes + 256);
        rc |= e->d_keys.ensure(kbytes + 16); rc |= e->d_off.ensure(n + 1); rc |= e->d_i64.ensure(n64);
        rc |= e->d_beh.ensure(n); rc |= e->d_u8.ensure((size_t)n * 2);
        rc |= e->d_out64.ensure((size_t)n * 3); rc |= e->d_out8.ensure((size_t)n * 2);
    }
    if (rc) return GUBER_E_NOMEM;
    // carve the arena (8-byte aligned pieces first)
    uint8_t* base = zc ? e->z_stage.p : e->h_stage.p;
    SmallOut* sout = (SmallOut*)base; base += 64;
    int64_t* s64 = (int64_t*)base; base += n64 * 8;
    int64_t* o64 = (int64_t*)base; base += (size_t)n * 3 * 8;
    uint32_t* soff = (uint32_t*)base; base += (size_t)(n + 1) * 4;
    uint32_t* sbeh = (uint32_t*)base; base += (size_t)n * 4;
    uint8_t* su8 = base; base += (size_t)n * 2;
    uint8_t* o8 = base; base += (size_t)n * 2;
    uint8_t* skeys = base;
    uint32_t off = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t i = idx ? idx[j] : j;
        const uint32_t len = b->key_off[i + 1] - b->key_off[i];
        memcpy(skeys + off, b->key_bytes + b->key_off[i], len);
        soff[j] = off; off += len;
        s64[j] = b->hits[i]; s64[n + j] = b->limit[i]; s64[2 * (size_t)n + j] = b->duration[i];
        s64[3 * (size_t)n + j] = has_burst ? b->burst[i] : 0;
        s64[4 * (size_t)n + j] = has_created ? b->created_at[i] : b->now_ms;
        s64[5 * (size_t)n + j] = has_greg ? b->greg_expire[i] : 0;
        s64[6 * (size_t)n + j] = has_greg ? b->greg_duration[i] : 0;
        sbeh[j] = b->behavior ? b->behavior[i] : 0;
        su8[j] = b->algorithm ? b->algorithm[i] : 0;
        su8[n + j] = b->is_owner ? b->is_owner[i] : 1;
    }
    soff[n] = off;
    memset(skeys + off, 0, 16);
    hipStream_t st = e->stream;
    std::vector<uint8_t> h_sflags; std::vector<Rec> h_safter;
    if (sev) {
        if (e->d_sflags.ensure(n) || e->d_safter.ensure(n)) return GUBER_E_NOMEM;
        HIPCHK(hipMemsetAsync(e->d_sflags.p, 0, n, st));
        e->W.store_flags = e->d_sflags.p; e->W.store_after = e->d_safter.p;
    }
    if (zc) {
        // the kernels read the request arrays and write the responses in place, over PCIe: no copy launches
        BatchView B{n, 0, skeys, soff, s64, s64 + n, s64 + 2 * (size_t)n, s64 + 3 * (size_t)n, s64 + 4 * (size_t)n,
                    su8, sbeh, su8 + n, has_greg ? s64 + 5 * (size_t)n : nullptr, has_greg ? s64 + 6 * (size_t)n : nullptr, b->now_ms};
        ResultView R{o8, o64, o64 + n, o64 + 2 * (size_t)n, o8 + n};
        bool done = false;
        if (n <= FT && !sev && !e->no_small && !e->careful && !lru_may_bind(e, n)) {
            // one launch, one workgroup; completion = a sequence number in host memory, polled
            const uint32_t seq = ++e->small_seq ? e->small_seq : ++e->small_seq;
            sout->done = 0;
            rc = launch_small(e, B, R, sout, seq);
            if (rc) return rc;
            volatile unsigned int* flag = &sout->done;
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(st)); break; }
            }
            if (!sout->fallback) {
                e->last_ctr.over += sout->over; e->last_ctr.hits += sout->hits; e->last_ctr.misses += sout->misses; e->last_ctr.size += sout->size_delta;
                done = true;
            } else e->small_fallbacks++;
        }
        if (!done) {
            rc = launch_batch(e, B, R, true);
            e->W.store_flags = nullptr; e->W.store_after = nullptr;
            if (rc) return rc;
            if (sev) {
                try { h_sflags.resize(n); h_safter.resize(n); } catch (...) { return GUBER_E_NOMEM; }
                HIPCHK(hipMemcpyAsync(h_sflags.data(), e->d_sflags.p, n, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(h_safter.data(), e->d_safter.p, (size_t)n * sizeof(Rec), hipMemcpyDeviceToHost, st));
            }
            rc = enqueue_counter_readback(e);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(st));
            fold_counters(e);
        }
    } else {
        HIPCHK(hipMemcpyAsync(e->d_keys.p, skeys, kbytes + 16, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_off.p, soff, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_i64.p, s64, n64 * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_beh.p, sbeh, (size_t)n * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_u8.p, su8, (size_t)n * 2, hipMemcpyHostToDevice, st));
        int64_t* d64 = e->d_i64.p;
        BatchView B{n, 0, e->d_keys.p, e->d_off.p, d64, d64 + n, d64 + 2 * (size_t)n, d64 + 3 * (size_t)n, d64 + 4 * (size_t)n,
                    e->d_u8.p, e->d_beh.p, e->d_u8.p + n, has_greg ? d64 + 5 * (size_t)n : nullptr, has_greg ? d64 + 6 * (size_t)n : nullptr, b->now_ms};
        ResultView R{e->d_out8.p, e->d_out64.p, e->d_out64.p + n, e->d_out64.p + 2 * (size_t)n, e->d_out8.p + n};
        rc = launch_batch(e, B, R);
        e->W.store_flags = nullptr; e->W.store_after = nullptr;
        if (rc) return rc;
        if (sev) {
            try { h_sflags.resize(n); h_safter.resize(n); } catch (...) { return GUBER_E_NOMEM; }
            HIPCHK(hipMemcpyAsync(h_sflags.data(), e->d_sflags.p, n, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(h_safter.data(), e->d_safter.p, (size_t)n * sizeof(Rec), hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipMemcpyAsync(o64, e->d_out64.p, (size_t)n * 3 * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(o8, e->d_out8.p, (size_t)n * 2, hipMemcpyDeviceToHost, st));
        rc = enqueue_counter_readback(e);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(st));
        fold_counters(e);
    }
    e->W.store_flags = nullptr; e->W.store_after = nullptr;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t i = idx ? idx[j] : j;
        r->status[i] = o8[j]; r->err[i] = o8[n + j];
        r->limit[i] = o64[j]; r->remaining[i] = o64[n + j]; r->reset_time[i] = o64[2 * (size_t)n + j];
        if (sev && o8[n + j] != GUBER_ITEM_E_RETRY) {
            sev->flags[i] = h_sflags[j];
            if (h_sflags[j] & GUBER_STORE_ONCHANGE) {
                item_from_rec(h_safter[j], &sev->items[i]);
                sev->items[i].key = b->key_bytes + b->key_off[i];
                sev->items[i].key_len = b->key_off[i + 1] - b->key_off[i];
            }
        }
    }
    return 0;
}

// (engine mutex held by the caller: the GLOBAL exchange re-runs collided rows through here without letting go of its engines)
static int eval_batch_host_locked(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r, guber_store_events_t* sev) {
    int rc = 0;
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    const DevCounters before = e->last_ctr;
    r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0;
    if (b->n > e->max_batch) return fail(GUBER_E_BATCH_TOO_LARGE, "batch larger than guber_config_t.max_batch");
    if (b->n) {
        rc = eval_host_once(e, b, r, nullptr, b->n, sev);
        if (rc) return rc;
        // two new keys sharing one 64-bit hash inside one batch: re-submit the affected items; on the
        // second pass the first key is resident and the other one probes past it.
        for (int round = 0; round < 64; ++round) {
            std::vector<uint32_t> again;
            for (uint32_t i = 0; i < b->n; ++i) if (r->err[i] == GUBER_ITEM_E_RETRY) again.push_back(i);
            if (again.empty()) break;
            e->careful = true;
            rc = eval_host_once(e, b, r, again.data(), (uint32_t)again.size(), sev);
            e->careful = false;
            if (rc) return rc;
        }
        rc = maintain(e, 0, b->now_ms);
        if (rc) return rc;
    }
    r->over_limit_count = e->last_ctr.over - before.over;
    r->cache_hits = e->last_ctr.hits - before.hits;
    r->cache_misses = e->last_ctr.misses - before.misses;
    r->unexpired_evictions = e->last_ctr.evictions - before.evictions;
    r->cache_size = e->last_ctr.size;
    return GUBER_OK;
}
static int eval_batch_host(guber_engine_t* e, const guber_batch_t* b, guber_result_t* r, guber_store_events_t* sev) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    const int rc = check_batch_args(b, r);
    if (rc) return rc;
    if (sev && b->n && (!sev->flags || !sev->items)) return fail(GUBER_E_INVALID_ARG, "null store event arrays");
    if (sev && b->n) memset(sev->flags, 0, b->n);
    std::lock_guard<std::mutex> lk(e->mu);
    return eval_batch_host_locked(e, b, r, sev);
}
