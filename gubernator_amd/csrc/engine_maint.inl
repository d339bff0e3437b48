// engine_maint.inl — part of guber_engine.hip's translation unit (included there, in this order; not a header of its own):
// maintenance between batches (compaction, eviction, the counters' bounds), per-kernel timing, the time zone.
// Rebuild the table keeping only live buckets (and buckets with pending GLOBAL work).  Called explicitly (guber_compact) or
// by maintain() when the directory is above its load limit.
static int compact_table(guber_engine* e, int64_t now_ms) {
    quiesce_all(e);
    DevBuf<DirEntry> ndir; DevBuf<Bucket> nb; DevBuf<uint8_t> narena; DevBuf<GPend> ngp; DevBuf<CompactOut> d_out;
    int rc = ndir.ensure(e->slots) | nb.ensure(e->slots) | narena.ensure(e->T.arena_cap + 64) | d_out.ensure(1);
    if (e->T.gpend) rc |= ngp.ensure(e->slots);
    auto cleanup = [&]() { ndir.release(); nb.release(); narena.release(); ngp.release(); d_out.release(); };
    if (rc) { cleanup(); return GUBER_E_NOMEM; }
    hipError_t he;
    if ((he = hipMemsetAsync(ndir.p, 0, e->slots * sizeof(DirEntry), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(nb.p, 0, e->slots * sizeof(Bucket), e->stream)) != hipSuccess ||
        (he = hipMemsetAsync(d_out.p, 0, sizeof(CompactOut), e->stream)) != hipSuccess ||
        (ngp.p && (he = hipMemsetAsync(ngp.p, 0, e->slots * sizeof(GPend), e->stream)) != hipSuccess)) {
        cleanup();
        return fail(GUBER_E_HIP, "compaction", he);
    }
    Table N = e->T;
    N.dir = ndir.p; N.buckets = nb.p; N.arena = narena.p;
    if (e->T.gpend) { N.gpend = ngp.p; N.gdirty = e->gdirty2.p; }
    hipLaunchKernelGGL(k_compact, dim3((unsigned)((e->slots + 255) / 256)), dim3(256), 0, e->stream, e->T, e->slots, N, now_ms, d_out.p);
    CompactOut co{};
    if ((he = hipMemcpyAsync(&co, d_out.p, sizeof(co), hipMemcpyDeviceToHost, e->stream)) != hipSuccess ||
        (he = hipStreamSynchronize(e->stream)) != hipSuccess) {
        cleanup();
        return fail(GUBER_E_HIP, "compaction", he);
    }
    // tags_used = kept entries; the live count is unchanged except for the expired buckets that were dropped: recount it
    // from the kept entries that are live (kept - pending-but-dead is not tracked separately: size := live kept)
    DevCounters c;
    HIPCHK(hipMemcpy(&c, e->ctr.p, sizeof(c), hipMemcpyDeviceToHost));
    std::vector<BlockCounters> bc(e->n_bctr);
    HIPCHK(hipMemcpy(bc.data(), e->bctr.p, e->n_bctr * sizeof(BlockCounters), hipMemcpyDeviceToHost));
    for (auto& x : bc) x.size_delta = 0;
    c.size = (long long)co.live; c.tags_used = co.kept; c.arena_head = co.arena_head;
    if (e->T.gpend) c.gdirty_n = co.gdirty_n;
    HIPCHK(hipMemcpy(e->ctr.p, &c, sizeof(c), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->bctr.p, bc.data(), e->n_bctr * sizeof(BlockCounters), hipMemcpyHostToDevice));
    std::swap(e->dir.p, ndir.p); std::swap(e->buckets.p, nb.p); std::swap(e->arena.p, narena.p);
    std::swap(e->dir.cap, ndir.cap); std::swap(e->buckets.cap, nb.cap); std::swap(e->arena.cap, narena.cap);
    if (e->T.gpend) {
        std::swap(e->gpend.p, ngp.p); std::swap(e->gpend.cap, ngp.cap);
        std::swap(e->gdirty.p, e->gdirty2.p);
        e->T.gpend = e->gpend.p; e->T.gdirty = e->gdirty.p;
    }
    cleanup();
    e->T.dir = e->dir.p; e->T.buckets = e->buckets.p; e->T.arena = e->arena.p;
    e->tags_upper = co.kept; e->size_upper = co.live;
    e->last_ctr.size = (long long)co.live; e->last_ctr.tags_used = co.kept;
    rb_disarm_all(e);
    e->lru_tail_ok = false;                                          // the tail list names slots of the old table
    e->compactions++;
    return 0;
}

// Bring the cache down to cache_size: the least recently used items go, in the list's exact order (lrucache.go:98-100,138-149).
// Batches never leave the cache above its size (their pre-pass evicts as the reference does, in the middle of the batch); this is
// what Add / UpdatePeerGlobals / Load need — adding n items and then dropping the oldest leaves exactly the items the reference's
// item-by-item Add leaves — and the safety net behind everything else.
static int evict_to_size(guber_engine* e, int64_t now_ms) {
    uint32_t st = 0;
    const LruKeys none{};
    const int rc = lru_admit(e, none, 0, now_ms, &st);
    if (!rc) e->evict_passes++;
    return rc;
}

// Keep the cache within cache_size and the directory under its load limit before `incoming` more requests arrive.
// The bounds are upper bounds (every request in flight might create an item).  Near a limit, counter snapshots are kept on their
// way (one riding on every batch) and folded as they complete; the stream is drained only when an eviction / rebuild is really
// due or a HARD limit (physical room) is at stake.
// defer_hard (GUBER_FUSE_EP, launch_group): the caller is holding a k_eval3 back on this stream — anything that would enqueue, synchronise
// or read the counters must wait until that has been launched: *defer_hard = true and NOTHING is done; the caller launches it and calls again
static int maintain(guber_engine* e, uint64_t incoming, int64_t now_ms, bool batch_follows, bool* defer_hard) {
    const uint64_t tag_limit = e->slots - e->slots / 8;   // keep >= 1/8 of the entries free
    const uint64_t hard_size = e->cache_size + std::max<uint64_t>(e->cache_size / 2, 4ull * e->max_batch);
    if (e->rb_ride >= 0 && !batch_follows) {              // a snapshot that was to ride on a batch that never came: launch it now
        if (defer_hard) { *defer_hard = true; return 0; }
        const uint32_t i = (uint32_t)e->rb_ride;
        e->rb_ride = -1;
        rb_launch(e, i);
    }
    if (rb_any_armed(e)) rb_fold_newest(e);                // exact as of the newest completed snapshot + what was enqueued since
    // A call whose size bound reaches cache_size goes through the eviction pre-pass, which synchronises (lru_may_bind / lru_admit),
    // so the bound is tightened EARLY: from 16 calls' worth of requests below a limit on, every batch carries a snapshot, and the
    // bound a decision is taken on is the exact count a few batches ago plus what was enqueued since.
    const uint64_t early = std::min<uint64_t>(16 * incoming, e->cache_size / 2);
    const bool over = e->size_upper > e->cache_size, tags = e->tags_upper + incoming > tag_limit;
    const bool near = e->size_upper + incoming + early > e->cache_size || e->tags_upper + incoming + early > tag_limit;
    if (!over && !tags && !near) return 0;
    const bool sure_over = (uint64_t)std::max<long long>(e->last_ctr.size, 0) > e->cache_size && !rb_any_armed(e);
    const bool hard = (incoming == 0 && (over || tags)) || e->size_upper > hard_size || tags || sure_over;
    if (!hard) {
        if (batch_follows) (void)rb_arm(e, true);            // no launch of its own: the batch's first kernel carries it
        else if (defer_hard) { *defer_hard = true; return 0; }
        else if (!rb_any_armed(e)) { (void)rb_arm(e, false); HIPCHK(hipGetLastError()); }
        return 0;
    }
    if (defer_hard) { *defer_hard = true; return 0; }
    int rc = engine_refresh_counters(e);
    if (rc) return rc;
    if ((uint64_t)std::max<long long>(e->last_ctr.size, 0) > e->cache_size) {
        rc = evict_to_size(e, now_ms);
        if (rc) return rc;
    }
    if (e->tags_upper + incoming > tag_limit) {
        // dead entries (expired, removed, evicted) still hold their tags: rebuild without them
        rc = compact_table(e, now_ms);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int guber_compact(guber_engine_t* e, int64_t now_ms) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    if (now_ms > e->clock_ms) e->clock_ms = now_ms;
    return compact_table(e, now_ms);
}

// the engine has no clock of its own: `now` comes with every batch; maintenance between batches (eviction after Add) uses
// the latest value seen, which a caller with a frozen or external clock sets here (clock.Freeze / clock.Advance)
extern "C" int guber_set_clock(guber_engine_t* e, int64_t now_ms) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    e->clock_ms = now_ms;
    return GUBER_OK;
}

extern "C" int guber_profile_enable(guber_engine_t* e, int enable) {
    if (!e) return fail(GUBER_E_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    e->profiling = enable != 0;
    return GUBER_OK;
}
extern "C" int guber_profile_read(guber_engine_t* e, guber_kernel_time_t* out, uint32_t cap, uint32_t* n_out) {
    if (!e || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    HIPCHK(hipStreamSynchronize(e->stream));
    {   // one pipeline pass = the spans from a first-stage kernel up to the next first-stage kernel
        // (k_evalpart_multi both ends a pass — the previous group's k_eval3 — and begins the next: a pass that is followed by one
        // lasts until the end of that launch)
        auto first_stage = [](int k) { return k == KT_FRONT || k == KT_FRONT_MULTI || k == KT_PART || k == KT_PART_MULTI || k == KT_RESOLVE || k == KT_EVALPART_MULTI; };
        // (a front's routing kernels run on a stream of their own and belong to no pass: guber_front_latencies times a generation's way)
        std::vector<const guber_engine::Span*> sp;
        for (auto& x : e->spans) if (x.kernel < KT_FR_COUNT || x.kernel > KT_FR_OUT) sp.push_back(&x);
        size_t g0 = 0;
        for (size_t i = 0; i <= sp.size(); ++i) {
            if (i == sp.size() || (i > g0 && first_stage(sp[i]->kernel))) {
                if (i > g0 && first_stage(sp[g0]->kernel)) {
                    float ms = 0.f;
                    const size_t last = i < sp.size() && sp[i]->kernel == KT_EVALPART_MULTI ? i : i - 1;
                    if (hipEventElapsedTime(&ms, sp[g0]->a, sp[last]->b) == hipSuccess) e->group_us.push_back(ms * 1e3f);
                }
                g0 = i;
            }
        }
    }
    for (auto& s : e->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { e->prof_ms[s.kernel] += ms; e->prof_n[s.kernel]++; }
        e->event_pool.push_back(s.a); e->event_pool.push_back(s.b);
    }
    e->spans.clear();
    *n_out = KT_COUNT;
    for (uint32_t k = 0; k < KT_COUNT && k < cap && out; ++k) {
        memset(&out[k], 0, sizeof(out[k]));
        snprintf(out[k].name, sizeof(out[k].name), "%s", kKernelNames[k]);
        out[k].launches = e->prof_n[k]; out[k].total_ms = e->prof_ms[k]; out[k].units = e->prof_units[k];
    }
    if (out) for (int k = 0; k < KT_COUNT; ++k) { e->prof_ms[k] = 0; e->prof_n[k] = 0; e->prof_units[k] = 0; }
    return GUBER_OK;
}

// the process's zone: the host helpers' copy and, on every visible device, the kernels' (guber_table.h g_tz)
extern "C" int guber_set_timezone(const guber_tz_t* tz) {
    // validate first; then every device; the host helpers' copy LAST, and only when every device has the zone — a failure part of the
    // way leaves the devices that were reached in the new zone and says so, the host (and with it guber_gregorian_*) in the old one
    // never ahead of them; the caller's current device is restored on every path
    guber::TzTable t{};
    const int rc = guber_host_build_tz(tz, &t);
    if (rc != GUBER_OK) return fail(rc, "time zone: at most 16 transitions, ascending");
    int ndev = 0, cur = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { guber_host_publish_tz(t); return GUBER_OK; }   // (no device: the helpers still follow the zone)
    (void)hipGetDevice(&cur);
    int failed = -1;
    for (int d = 0; d < ndev && failed < 0; ++d) {
        if (hipSetDevice(d) != hipSuccess) continue;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(guber::g_tz), &t, sizeof(guber::TzTable)) != hipSuccess) failed = d;
    }
    (void)hipSetDevice(cur);
    if (failed >= 0) { (void)hipGetLastError(); return fail(GUBER_E_HIP, "time zone: a device did not take the table (the host helpers keep the zone they had)"); }
    guber_host_publish_tz(t);
    return GUBER_OK;
}

extern "C" int guber_profile_passes(guber_engine_t* e, float* us, uint32_t cap, uint32_t* n_out) {
    if (!e || !n_out) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    *n_out = (uint32_t)e->group_us.size();
    if (us) {
        for (uint32_t k = 0; k < cap && k < e->group_us.size(); ++k) us[k] = e->group_us[k];
        e->group_us.clear();
    }
    return GUBER_OK;
}

extern "C" const char* guber_last_error(void) { return g_last_error.c_str(); }

#include "guber_global_sync.h"
#include "guber_wire_dev.h"
#include "guber_wire_pool.h"    // guber_wire_pool_*: the payload stage (caller threads hand over serialized RPCs)
