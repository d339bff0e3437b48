// engine_dispatch.inl — part of guber_engine.hip's translation unit (included there, in this order; not a header of its own):
// one dispatcher for several engines: fused launches of up to four tables, the held-back k_eval3 (GUBER_FUSE_EP), guber_eval_batches_routed_dev.
// One dispatcher for several engines (the logical shards of a GPU, each a table of its own): batch k goes to
// engines[which[k]]; per engine the array order is kept, between engines there is nothing to order (disjoint keys, no
// shared state).  Each round takes the next batch of every engine that has one and, where the engines share device and
// stream and the batches take the two-launch pipeline, enqueues up to MULTI_MAX of them as ONE k_front_multi + ONE
// k_eval2_multi (guber_kernels.h): the batches' dependent memory trips then overlap inside a launch, without the
// per-stream kernel boundaries that throttle shards running on separate streams (profiles/archive/r02_m_shard_streams.txt).
static bool fits_fused(const guber_engine* e, uint32_t n) {
#ifdef GUBER_PHASE_TIMING
    return false;
#else
    return e->fuse && takes_fast_path(e, n);
#endif
}
// (a batch that may overflow the cache goes alone, through launch_batch and its eviction pre-pass)
static bool can_fuse(guber_engine* e, uint32_t n) { return fits_fused(e, n) && !lru_may_bind_unlocked(e, n); }   // (takes the engine mutex for the look)

// GUBER_FUSE_EP: a group's k_eval3_multi that has not been launched yet — held back until the same tables' next group comes (then
// it shares that group's first launch: k_evalpart_multi) or until anything else is about to be enqueued on its stream / the call ends
// (then it goes on its own).  Lives inside ONE guber_eval_batches_routed_dev call, one per stream the call uses.
// guber_front: "every evaluation of generation g on this stream has been launched" as an event the answers' way home waits for.
// A held-back evaluation carries the hook of its generation; whoever launches it — the dispatcher's next group (k_evalpart_multi), a
// flush, another thread's entry point — counts it off, and the last one records the event behind the launch.
struct EvalHook {
    hipEvent_t ev = nullptr; hipStream_t st = nullptr;
    std::atomic<int> outstanding{0}; std::atomic<bool> recorded{false};
    void launched() { if (outstanding.fetch_sub(1) == 1) { (void)hipEventRecord(ev, st); recorded.store(true, std::memory_order_release); } }
};
struct PendingEval {
    std::mutex pm;                                                 // two threads that each hold ONE of the group's engines may both come to launch it
    EvalHook* hook = nullptr;                                      // (written under pm)
    std::atomic<bool> valid{false};                                // (written under pm; the dispatcher also looks before it has taken the engines' locks, and again after)
    int n = 0; uint32_t tiles = 0; uint64_t units = 0;
    guber_engine* eng[MULTI_MAX]; MultiEval ME;
};
static thread_local int tl_ep_dispatcher = 0;                      // this thread is inside a routed call that holds evaluations back: it launches them itself
// launch it (if it has not been launched).  The caller holds the mutex of at least one of its engines: nothing can be enqueued on
// any of them by the dispatcher meanwhile (it takes them all), and the launch lands on their stream before whatever the caller enqueues next.
static int launch_held(PendingEval& p, bool spans) {
    std::lock_guard<std::mutex> lk(p.pm);
    if (!p.valid) return 0;
    p.valid = false;
    guber_engine* e0 = p.eng[0];
    if (hipSetDevice(e0->device) != hipSuccess) return fail(GUBER_E_HIP, "hipSetDevice");
    if (spans) e0->span_begin(KT_EVAL3_MULTI, p.units);            // (per-kernel timing belongs to the group's first engine: only under its mutex)
    hipLaunchKernelGGL(k_eval3_multi, dim3(p.tiles), dim3(256), 0, e0->stream, p.ME);
    if (spans) e0->span_end();
    if (p.hook) { p.hook->launched(); p.hook = nullptr; }
    if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    return 0;
}
// an entry point other than the dispatcher that holds it back, with e's mutex held
static void ep_flush_held(const guber_engine* e) {
    if (!e->held || tl_ep_dispatcher) return;
    (void)launch_held(*e->held, false);
    e->held = nullptr;
}
// the dispatcher's own: with all of its engines locked (engines_locked) or locking them here
static int flush_pending(PendingEval& p, bool engines_locked = false) {
    guber_engine* order[MULTI_MAX];
    for (int i = 0; i < p.n; ++i) order[i] = p.eng[i];
    std::sort(order, order + p.n);
    if (!engines_locked) for (int i = 0; i < p.n; ++i) order[i]->mu.lock();
    struct Unlock { guber_engine** o; int g; ~Unlock() { for (int i = g - 1; i >= 0; --i) o[i]->mu.unlock(); } } unlock{order, engines_locked ? 0 : p.n};
    const int rc = launch_held(p, true);
    for (int i = 0; i < p.n; ++i) if (p.eng[i]->held == &p) p.eng[i]->held = nullptr;
    return rc;
}
// every k_eval3 a call is holding back: at most one per set of engines (sets are disjoint: one that overlaps a new group without
// being it is launched before the group is)
struct PendSet {
    std::vector<std::unique_ptr<PendingEval>> items;
    // (a slot only ever serves ONE set of tables: an engine's `held` may outlive a foreign launch and must not come to mean another group)
    PendingEval* slot_for(guber_engine* const* grp, int g) {
        for (auto& q : items) {
            if (q->valid || q->n != g) continue;
            bool same = true;
            for (int i = 0; i < g && same; ++i) same = q->eng[i] == grp[i];
            if (same) return q.get();
        }
        items.emplace_back(new PendingEval());
        items.back()->n = g;
        for (int i = 0; i < g; ++i) items.back()->eng[i] = grp[i];
        return items.back().get();
    }
    int flush_touching(guber_engine* const* grp, int g, const PendingEval* keep = nullptr) {
        for (auto& q : items) {
            if (q.get() == keep) continue;                          // (also the ones a foreign thread launched: their engines' `held` is cleared here)
            bool overlap = false;
            for (int i = 0; i < q->n && !overlap; ++i) for (int j = 0; j < g && !overlap; ++j) overlap = q->eng[i] == grp[j];
            if (overlap) { const int rc = flush_pending(*q); if (rc) return rc; }
        }
        return 0;
    }
    int flush_all() { int r = 0; for (auto& q : items) { const int rc = flush_pending(*q); if (!r) r = rc; } return r; }
};
// GUBER_DISPATCH_PROFILE=1: where the dispatcher's time goes (printed at the end of every guber_eval_batches_routed_dev call):
// [0] waiting for the GPU's progress before a batch may be enqueued (can_fuse -> lru_may_bind), [1] locks + held-back launches,
// [2] preludes + plans, [3] argument blocks, [4] inside hipLaunchKernelGGL, [5] groups, [6] batches
static const bool g_dprof = guber_lab_env("GUBER_DISPATCH_PROFILE") != nullptr;
static thread_local uint64_t tl_dp[8];
static inline uint64_t dp_now() { return g_dprof ? (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0; }
struct DpSpan { int k; uint64_t t0; explicit DpSpan(int kk) : k(kk), t0(dp_now()) {} ~DpSpan() { if (g_dprof) tl_dp[k] += dp_now() - t0; } };
// (the batches as views: a caller's guber_batch_t, or an engine's share of a front's generation — guber_front.h; hook: the front's
// "this generation's evaluations on this stream have all been launched" — a held-back evaluation takes it along)
struct GroupItem { BatchView B; ResultView R; };
static int launch_group(guber_engine* const* grp, const GroupItem* it, int g, uint32_t* enqueued, PendSet* ps = nullptr, EvalHook* hook = nullptr) {
    if (g_dprof) { tl_dp[5]++; tl_dp[6] += (uint64_t)g; }
    auto views = [&](int i, BatchView& B, ResultView& R) { B = it[i].B; R = it[i].R; };
    // (one batch: launch_batch.  Measured in round 5 and not kept: a sequence of ONE table's batches through these fused launches —
    // 1.50 against 2.40 G decisions/s: 128 k_own workgroups for the whole chip take 36 us, profiles/r05_g_one_table_fused.txt)
    if (g == 1) {
        guber_engine* e = grp[0];
        if (ps) { const int rc = ps->flush_touching(grp, 1); if (rc) return rc; }
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->held) { (void)launch_held(*e->held, false); e->held = nullptr; }      // (another call's)
        if (e->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
        BatchView B; ResultView R; views(0, B, R);
        const int rc = launch_batch(e, B, R);
        if (rc == 0) ++*enqueued;
        return rc;
    }
    // GUBER_FUSE_EP: the k_eval3 held back on this stream shares this group's first launch (k_evalpart_multi) if the group is the same
    // tables again, in the same order, all taking the owner-partitioned pipeline, and no prelude has anything to enqueue; otherwise it
    // goes first, on its own.  What needs no lock is decided here, before the group's locks are taken (flush_pending takes its own).
    bool same_set = false;
    PendingEval* pend = nullptr;                                   // the k_eval3 held back for exactly these tables, if there is one
    if (ps) {
        same_set = g <= EP_MAX;
        for (int i = 0; i < g && same_set; ++i) same_set = grp[i]->fuse_ep && takes_part_path(grp[i], it[i].B.n, false, true);
        for (auto& q : ps->items) {
            if (!q->valid || !same_set || q->n != g) continue;
            bool same = true;
            for (int i = 0; i < g && same; ++i) same = q->eng[i] == grp[i];
            if (same) { pend = q.get(); break; }
        }
        const int rc0 = ps->flush_touching(grp, g, pend);         // (one that holds some of these engines in another combination: first)
        if (rc0) return rc0;
    }
    // lock the group's engines in address order (any other caller holds at most one engine lock, or locks in this order)
    guber_engine* order[MULTI_MAX];
    uint64_t dp_t = dp_now();
    auto dp_lap = [&](int k) { if (g_dprof) { const uint64_t t = dp_now(); tl_dp[k] += t - dp_t; dp_t = t; } };
    for (int i = 0; i < g; ++i) order[i] = grp[i];
    std::sort(order, order + g);
    for (int i = 0; i < g; ++i) order[i]->mu.lock();
    struct Unlock { guber_engine** o; int g; ~Unlock() { for (int i = g - 1; i >= 0; --i) o[i]->mu.unlock(); } } unlock{order, g};
    for (int i = 0; i < g; ++i)                                    // a k_eval3 ANOTHER call holds back for one of these tables goes first
        if (grp[i]->held && grp[i]->held != pend) { (void)launch_held(*grp[i]->held, false); grp[i]->held = nullptr; }
    if (grp[0]->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    {   // can_fuse() looked at the cache bound BEFORE these locks were taken (it takes and drops each engine's mutex): another thread's
        // AddCacheItem / eval on one of the tables may have used the headroom since.  Looked at again here, under the locks, with the
        // cheap form of the bound; a table that is tight now leaves the group and goes through launch_batch and its eviction pre-pass,
        // one by one — the cache never grows past cache_size and the victims stay lrucache.go's
        bool tight = false;
        for (int i = 0; i < g; ++i) tight = tight || grp[i]->size_upper + it[i].B.n > grp[i]->cache_size;
        if (tight) {
            if (pend && pend->valid) { const int rcf = flush_pending(*pend, true); if (rcf) return rcf; }
            for (int i = g - 1; i >= 0; --i) order[i]->mu.unlock();
            unlock.g = 0;
            int rc1 = 0;
            for (int i = 0; i < g && !rc1; ++i) rc1 = launch_group(&grp[i], &it[i], 1, enqueued, ps, hook);
            return rc1;
        }
    }
    MultiFront MF{}; MultiEval ME{};
    uint32_t tiles = 0, ns[MULTI_MAX];
    int planned = 0, rc = 0;
    bool part = true;                                              // the group takes the owner-partitioned pipeline if all its batches do
    for (int i = 0; i < g; ++i) part = part && takes_part_path(grp[i], it[i].B.n, false, true);
    // GUBER_FUSE_EP: the k_eval3 held back on this stream shares this group's first launch if the group is the same tables again, in
    // the same order, and no prelude has anything to enqueue; otherwise it goes first, on its own
    const bool ep = ps && same_set && part;                       // (same_set, pend: decided before the locks were taken, below the g == 1 case)
    bool join = ep && pend && pend->valid;
    if (pend && pend->valid && !join) { rc = flush_pending(*pend, true); if (rc) return rc; }   // (pend => the same engines: locked)
    dp_lap(1);
    for (int i = 0; i < g; ++i) {
        guber_engine* e = grp[i];
        BatchView B; ResultView R; views(i, B, R);
        Work W; FastPlan P;
        bool defer = false;
        rc = batch_prelude(e, B, W, join ? &defer : nullptr);
        if (!rc && defer) {                                       // this prelude has something to enqueue or to read: the k_eval3 held back goes first
            join = false;
            rc = flush_pending(*pend, true);
            if (!rc) rc = batch_prelude(e, B, W);
        }
        if (!rc) rc = part ? plan_part(e, B, W, P) : plan_fast(e, B, false, W, P);
        if (rc) break;                                            // enqueue what is planned, then report
        tiles += P.ftiles;
        MF.end_tile[planned] = ME.end_tile[planned] = tiles;
        MF.sub[planned] = FrontArgs{e->T, P.B2, P.W};
        ME.sub[planned] = EvalArgs{e->T, P.B3, R, P.W};
        ns[planned++] = B.n;
    }
    dp_lap(2);
    if (planned) {
        static_assert(FT == 256, "k_eval2's workgroup is k_front's tile");
        MF.nb = ME.nb = (uint32_t)planned;
        uint64_t units = 0;
        for (int i = 0; i < planned; ++i) units += ns[i];
        if (part) {
            // (a prelude that was not quiet after all — a counter read-back now rides on this k_part — or a group cut short by an
            // error: the k_eval3 held back goes first)
            bool joined = join && pend->valid && planned == g;
            for (int i = 0; i < planned && joined; ++i) joined = MF.sub[i].T.buckets == pend->ME.sub[i].T.buckets;
            if (pend && pend->valid && !joined) { const int rcf = flush_pending(*pend, true); if (rcf) return rcf; }
            // a counter read-back riding on this k_part runs beside the held-back k_eval3: what it reads lies between the counters
            // before and after that batch, so the host counts that batch's requests among "enqueued since" as well (rb_fold_slot)
            for (int i = 0; i < planned && joined; ++i) {
                if (!MF.sub[i].W.snap_seq) continue;
                for (auto& slot : grp[i]->rb)
                    if (slot.armed && slot.seq == MF.sub[i].W.snap_seq) slot.mark -= std::min<uint64_t>(slot.mark, pend->ME.sub[i].B.n);
            }
            if (joined) {
                // ONE launch: workgroups [0, pending tiles) are the held-back k_eval3, the rest this group's k_part
                MultiEP EP{};
                EP.nb = (uint32_t)planned;
                for (int i = 0; i < planned; ++i) {
                    EP.end_e[i] = pend->ME.end_tile[i]; EP.end_p[i] = MF.end_tile[i];
                    EP.sub[i].E = pend->ME.sub[i]; EP.sub[i].Bp = MF.sub[i].B; EP.sub[i].did_p = MF.sub[i].W.did; EP.sub[i].pmslot_p = MF.sub[i].W.pmslot;
                    const Work& Wp = MF.sub[i].W;
                    EP.sub[i].snap_seq = Wp.snap_seq; EP.sub[i].snap_n = Wp.snap_n; EP.sub[i].snap_c = Wp.snap_c; EP.sub[i].snap_b = Wp.snap_b; EP.sub[i].snap_stamp = Wp.snap_stamp;
                }
                EvalHook* joined_hook;
                { std::lock_guard<std::mutex> pl(pend->pm); pend->valid = false; joined_hook = pend->hook; pend->hook = nullptr; }
                for (int i = 0; i < planned; ++i) grp[i]->held = nullptr;
                dp_lap(3);
                grp[0]->span_begin(KT_EVALPART_MULTI, pend->units);
                hipLaunchKernelGGL(k_evalpart_multi, dim3(pend->tiles + tiles), dim3(256), 0, grp[0]->stream, EP);
                grp[0]->span_end();
                if (joined_hook) joined_hook->launched();
                grp[0]->ep_launches++;
            } else {
                dp_lap(3);
                grp[0]->span_begin(KT_PART_MULTI, units);
                hipLaunchKernelGGL(k_part_multi, dim3(tiles), dim3(FT), 0, grp[0]->stream, MF);
                grp[0]->span_end();
            }
            grp[0]->span_begin(KT_OWN_MULTI, units);
            hipLaunchKernelGGL(k_own_multi, dim3((unsigned)planned * PT_PARTS), dim3(256), 0, grp[0]->stream, MF);
            grp[0]->span_end();
            dp_lap(4);
            if (ep && planned == g) {                             // held back: the same tables' next group, or flush_pending, launches it
                if (!pend) pend = ps->slot_for(grp, planned);
                {
                    std::lock_guard<std::mutex> pl(pend->pm);
                    pend->valid = true; pend->n = planned; pend->tiles = tiles; pend->units = units; pend->ME = ME;
                    pend->hook = hook;
                    if (hook) hook->outstanding.fetch_add(1);
                }
                for (int i = 0; i < planned; ++i) { pend->eng[i] = grp[i]; grp[i]->held = pend; grp[i]->batches++; grp[i]->part_batches++; grp[i]->fused_batches++; }
                *enqueued += (uint32_t)planned;
                if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
                dp_lap(3);
                return rc;
            }
            grp[0]->span_begin(KT_EVAL3_MULTI, units);
            hipLaunchKernelGGL(k_eval3_multi, dim3(tiles), dim3(256), 0, grp[0]->stream, ME);
            grp[0]->span_end();
            for (int i = 0; i < planned; ++i) { grp[i]->batches++; grp[i]->part_batches++; grp[i]->fused_batches++; }
            *enqueued += (uint32_t)planned;
            if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
            return rc;
        }
        grp[0]->span_begin(KT_FRONT_MULTI, units);                    // (per-kernel timing, when enabled, is kept by the group's first engine)
        hipLaunchKernelGGL(k_front_multi, dim3(tiles), dim3(FT), 0, grp[0]->stream, MF);
        grp[0]->span_end();
        grp[0]->span_begin(KT_EVAL2_MULTI, units);
        hipLaunchKernelGGL(k_eval2_multi, dim3(tiles), dim3(256), 0, grp[0]->stream, ME);
        grp[0]->span_end();
        for (int i = 0; i < planned; ++i) { finish_fast(grp[i], ns[i]); grp[i]->fused_batches++; }
        *enqueued += (uint32_t)planned;
        if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    }
    return rc;
}

// Up to 16 tables of ONE device and stream in ONE pair of launches (k_front_multi_mem / k_eval2_multi_mem: the argument blocks travel through
// device memory, one copy command ahead of the pair): what a payload stage's generation is — a few thousand requests per table, where the
// owner-partitioned pipeline's three launches per group of four would be six or nine launches for eight or twelve tables.  The caller has
// checked that every share fits the two-launch pipeline and cannot make its cache bind (can_fuse) and that nothing is held back on these
// tables by its own call.  HA: device-visible host memory that stays untouched until the copy has run; DA: its place in HBM.
// Returns 1 when a table turned out tight under the locks (nothing was enqueued: the caller takes the ordinary way), 0 / < 0 otherwise.
static int launch_group_mem(guber_engine* const* grp, const GroupItem* it, int g, uint32_t* enqueued, MultiArgsMem* HA, MultiArgsMem* DA) {
    guber_engine* order[MULTI_MEM_MAX];
    for (int i = 0; i < g; ++i) order[i] = grp[i];
    std::sort(order, order + g);
    for (int i = 0; i < g; ++i) order[i]->mu.lock();
    struct Unlock { guber_engine** o; int g; ~Unlock() { for (int i = g - 1; i >= 0; --i) o[i]->mu.unlock(); } } unlock{order, g};
    for (int i = 0; i < g; ++i)                                    // a k_eval3 ANOTHER call holds back for one of these tables goes first
        if (grp[i]->held) { (void)launch_held(*grp[i]->held, false); grp[i]->held = nullptr; }
    if (grp[0]->set_device()) return fail(GUBER_E_HIP, "hipSetDevice");
    for (int i = 0; i < g; ++i)
        if (grp[i]->size_upper + it[i].B.n > grp[i]->cache_size || grp[i]->small_pending) return 1;
    uint32_t tiles = 0, ns[MULTI_MEM_MAX];
    int planned = 0, rc = 0;
    for (int i = 0; i < g; ++i) {
        guber_engine* e = grp[i];
        Work W; FastPlan P;
        rc = batch_prelude(e, it[i].B, W);
        if (!rc) rc = plan_fast(e, it[i].B, false, W, P);
        if (rc) break;                                            // enqueue what is planned, then report
        tiles += P.ftiles;
        HA->F.end_tile[planned] = HA->E.end_tile[planned] = tiles;
        HA->F.sub[planned] = FrontArgs{e->T, P.B2, P.W};
        HA->E.sub[planned] = EvalArgs{e->T, P.B3, it[i].R, P.W};
        ns[planned++] = it[i].B.n;
    }
    if (planned) {
        HA->F.nb = HA->E.nb = (uint32_t)planned;
        uint64_t units = 0;
        for (int i = 0; i < planned; ++i) units += ns[i];
        hipStream_t st = grp[0]->stream;
        if (hipMemcpyAsync(DA, HA, sizeof(MultiArgsMem), hipMemcpyHostToDevice, st) != hipSuccess) return fail(GUBER_E_HIP, "hipMemcpyAsync");
        grp[0]->span_begin(KT_FRONT_MULTI, units);
        hipLaunchKernelGGL(k_front_multi_mem, dim3(tiles), dim3(FT), 0, st, (const MultiFrontMem*)&DA->F);
        grp[0]->span_end();
        grp[0]->span_begin(KT_EVAL2_MULTI, units);
        hipLaunchKernelGGL(k_eval2_multi_mem, dim3(tiles), dim3(256), 0, st, (const MultiEvalMem*)&DA->E);
        grp[0]->span_end();
        for (int i = 0; i < planned; ++i) { finish_fast(grp[i], ns[i]); grp[i]->fused_batches++; }
        *enqueued += (uint32_t)planned;
        if (hipGetLastError() != hipSuccess) return fail(GUBER_E_HIP, "kernel launch");
    }
    return rc;
}

// One round after the other: the next item of every engine that has one; engines that share device and stream share launches.
// fifo[j] = the items of engines[j] in their order.  *enqueued counts items.  The caller owns `ps` (and flushes it).
static int dispatch_rounds(guber_engine_t* const* engines, uint32_t n_engines, const std::vector<std::vector<GroupItem>>& fifo, PendSet* ps,
                           uint32_t* enqueued, EvalHook* const* hook_of_engine = nullptr) {
    std::vector<size_t> pos(n_engines, 0);
    for (;;) {
        guber_engine* grp[MULTI_MAX]; GroupItem git[MULTI_MAX]; int g = 0;
        EvalHook* hk = nullptr;
        bool any = false;
        int rc = 0;
        for (uint32_t j = 0; j < n_engines && !rc; ++j) {
            if (pos[j] >= fifo[j].size()) continue;
            any = true;
            guber_engine* e = engines[j];
            const GroupItem& item = fifo[j][pos[j]++];
            bool fits;
            { DpSpan sp(0); fits = can_fuse(e, item.B.n); }
            for (int i = 0; i < g && fits; ++i) fits = grp[i] != e;
            if (g && (!fits || g == MULTI_MAX || e->stream != grp[0]->stream || e->device != grp[0]->device)) {
                rc = launch_group(grp, git, g, enqueued, ps, hk);
                g = 0;
                if (rc) break;
            }
            grp[g] = e; git[g] = item; ++g;
            hk = hook_of_engine ? hook_of_engine[j] : nullptr;      // (engines of one stream share their generation's hook)
            if (!fits) { rc = launch_group(grp, git, g, enqueued, ps, hk); g = 0; }
        }
        if (!rc && g) rc = launch_group(grp, git, g, enqueued, ps, hk);
        if (rc) return rc;
        if (!any) break;
    }
    return 0;
}

extern "C" int guber_eval_batches_routed_dev(guber_engine_t* const* engines, uint32_t n_engines, const uint32_t* which,
                                             const guber_batch_t* batches, guber_result_t* results, uint32_t count, uint32_t* done) {
    if (done) *done = 0;
    if (!engines || !n_engines || (count && (!which || !batches || !results))) return fail(GUBER_E_INVALID_ARG, "null argument");
    std::vector<std::vector<GroupItem>> fifo(n_engines);
    uint32_t empty = 0;
    for (uint32_t k = 0; k < count; ++k) {
        if (which[k] >= n_engines || !engines[which[k]]) return fail(GUBER_E_INVALID_ARG, "which[k] names no engine");
        const int rc = check_batch_args(&batches[k], &results[k]);
        if (rc) return rc;
        const guber_batch_t* b = &batches[k]; guber_result_t* r = &results[k];
        r->over_limit_count = r->cache_hits = r->cache_misses = r->unexpired_evictions = 0; r->cache_size = 0;
        if (!b->n) { ++empty; continue; }
        fifo[which[k]].push_back(GroupItem{BatchView{b->n, 0, b->key_bytes, b->key_off, b->hits, b->limit, b->duration, b->burst, b->created_at,
                                                     b->algorithm, b->behavior, b->is_owner, b->greg_expire, b->greg_duration, b->now_ms},
                                           ResultView{r->status, r->limit, r->remaining, r->reset_time, r->err}});
    }
    uint32_t enqueued = 0;
    // GUBER_FUSE_EP engines: the k_eval3 of a group of tables is held back for the same tables' next group (launch_group)
    PendSet pendset;
    bool any_ep = false;
    for (uint32_t j = 0; j < n_engines; ++j) any_ep = any_ep || (engines[j] && engines[j]->fuse_ep);
    PendSet* const ps = any_ep ? &pendset : nullptr;
    struct Dispatching { bool on; Dispatching(bool o) : on(o) { if (on) ++tl_ep_dispatcher; } ~Dispatching() { if (on) --tl_ep_dispatcher; } } dispatching(any_ep);
    const int rc = dispatch_rounds(engines, n_engines, fifo, ps, &enqueued);
    const int rcf = pendset.flush_all();                            // (what was enqueued is completed: its k_eval3 goes now)
    if (done) *done = enqueued;
    if (rc) return rc;
    if (rcf) return rcf;
    if (g_dprof && tl_dp[6]) {
        fprintf(stderr, "[dispatch] %llu batches in %llu groups; per batch: wait-for-progress %.2f us, locks %.2f, preludes+plans %.2f, argument blocks %.2f, launches %.2f\n",
                (unsigned long long)tl_dp[6], (unsigned long long)tl_dp[5], tl_dp[0] / 1e3 / tl_dp[6], tl_dp[1] / 1e3 / tl_dp[6], tl_dp[2] / 1e3 / tl_dp[6],
                tl_dp[3] / 1e3 / tl_dp[6], tl_dp[4] / 1e3 / tl_dp[6]);
        for (auto& v : tl_dp) v = 0;
    }
    if (done) *done = enqueued + empty;
    return GUBER_OK;
}
