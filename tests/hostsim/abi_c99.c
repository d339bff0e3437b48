/* TEST-ONLY.  cgo compiles the preamble of go/gpu_worker_pool.go as C: this file is that preamble and the call sequence the Go
 * binding makes, as plain C99 (gcc -std=c99 -pedantic -Wall -Werror), so that include/guber_gpu.h and include/guber_wire.h are
 * checked to be usable from C — they are otherwise only ever included from C++ and mirrored by hand in ctypes.  Linked against the
 * product library and run without a GPU: pool creation must fail with GUBER_E_NO_DEVICE and nothing else may be reached. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "guber_gpu.h"
#include "guber_wire.h"

/* the //export trampolines of the Go side */
static int goStoreGet(void* user, guber_store_req_t* r, guber_item_t* out) { (void)user; (void)r; (void)out; return 0; }
static void goStoreOnChange(void* user, guber_store_req_t* r, guber_item_t* item) { (void)user; (void)r; (void)item; }
static void goStoreRemove(void* user, uint8_t* key, uint32_t key_len) { (void)user; (void)key; (void)key_len; }
static void goStoreSave(void* user, guber_item_t* item) { (void)user; (void)item; }

/* the cgo preamble's helpers, verbatim in shape */
static void guber_go_set_store(guber_pool_t* p, void* user) {
    guber_store_callbacks_t cb;
    cb.get = (int (*)(void*, const guber_store_req_t*, guber_item_t*))goStoreGet;
    cb.on_change = (void (*)(void*, const guber_store_req_t*, const guber_item_t*))goStoreOnChange;
    cb.remove = (void (*)(void*, const uint8_t*, uint32_t))goStoreRemove;
    cb.user = user;
    guber_pool_set_store(p, &cb);
}
static int guber_go_store_all(guber_pool_t* p, void* user) { return guber_pool_store(p, (void (*)(void*, const guber_item_t*))goStoreSave, user); }

/* NewGPUWorkerPool -> GetRateLimits -> AddCacheItem -> GetCacheItem -> Load -> Store -> GlobalSync -> Close, as the binding calls them */
static int call_sequence(int really) {
    guber_config_t cfg;
    int32_t devs[1] = {0};
    guber_pool_t* pool = NULL;
    int rc;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg;
    cfg.cache_size = 50000;
    cfg.max_batch = 1000;
    cfg.flags = GUBER_FLAG_GLOBAL;
    cfg.device = devs[0];
    rc = guber_pool_create_multi(&cfg, devs, 1, 8, 1000, 500, &pool);
    if (rc != GUBER_OK) {
        printf("create: %s (%s)\n", guber_strerror(rc), guber_last_error());
        return rc;
    }
    if (!really) return 0;
    {   /* setEngineTimezone: the daemon's zone (here: UTC+1 until a transition to UTC+2, then back), and UTC again for what follows */
        const int64_t when_s[2] = {1900000000, 1915000000};
        const int32_t offset_s[2] = {7200, 3600};
        guber_tz_t tz;
        tz.n = 2; tz.offset0_s = 3600; tz.when_s = when_s; tz.offset_s = offset_s;
        rc = guber_set_timezone(&tz);
        if (rc == GUBER_OK) rc = guber_set_timezone(NULL);
        if (rc != GUBER_OK) { printf("set_timezone: %s (%s)\n", guber_strerror(rc), guber_last_error()); guber_pool_destroy(pool); return rc; }
    }
    {
        enum { N = 2, STRIDE = 200 };
        const char names[] = "ab", ukeys[] = "xy";
        uint32_t name_off[N + 1] = {0, 1, 2}, ukey_off[N + 1] = {0, 1, 2};
        int64_t hits[N] = {1, 1}, limit[N] = {10, 10}, duration[N] = {1000, 1000}, burst[N] = {0, 0}, created[N] = {0, 0};
        int64_t o_limit[N], o_rem[N], o_reset[N];
        int32_t algo[N] = {0, 1};
        uint32_t behavior[N] = {0, GUBER_BEHAVIOR_GLOBAL};
        uint8_t owner[N] = {1, 0}, status[N], er[N];
        char text[N * STRIDE];
        guber_result_t out;
        guber_item_t item, got;
        int found = 0;
        guber_global_sync_stats_t st;
        memset(&out, 0, sizeof out);
        out.status = status; out.limit = o_limit; out.remaining = o_rem; out.reset_time = o_reset; out.err = er;
        guber_go_set_store(pool, NULL);
        rc = guber_pool_get_rate_limits_owner(pool, N, (const uint8_t*)names, name_off, (const uint8_t*)ukeys, ukey_off, hits, limit, duration, burst,
                                              created, algo, behavior, owner, &out, text, STRIDE);
        memset(&item, 0, sizeof item);
        item.key = (const uint8_t*)"a_x"; item.key_len = 3; item.limit = 10; item.duration = 1000; item.remaining = 5;
        rc |= guber_pool_add_item_for(pool, &item, GUBER_BEHAVIOR_GLOBAL);
        rc |= guber_pool_get_item(pool, (const uint8_t*)"a_x", 3, &got, &found);
        rc |= guber_pool_load(pool, &item, 1);
        rc |= guber_go_store_all(pool, NULL);
        rc |= guber_pool_global_sync(pool, &st);
    }
    guber_pool_destroy(pool);
    return rc;
}

/* the front of a GPU's shards from C — two engines on one stream, the placement's rule exported, one generation of
 * requests in arrival order in device-visible memory (guber_alloc_pinned: what a binding without a HIP allocator of its own has), the
 * answers in arrival order.  Two requests of one key must be answered in their order whatever engine the key lives on. */
static int front_sequence(void) {
    enum { N = 6, L = 4 };
    guber_config_t cfg;
    guber_engine_t* eng[2] = {NULL, NULL};
    guber_placement_t* place = NULL;
    guber_front_t* front = NULL;
    guber_front_stats_t fst;
    struct guber_route_rule rule;
    guber_batch_t gen;
    guber_result_t res;
    uint8_t* mem;
    int rc, i;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg; cfg.cache_size = 1000; cfg.max_batch = 1024;
    rc = guber_engine_create(&cfg, &eng[0]);
    if (rc != GUBER_OK) return rc;
    cfg.stream = guber_engine_stream(eng[0]);
    rc = guber_engine_create(&cfg, &eng[1]);
    if (rc == GUBER_OK) rc = guber_placement_create(2, 0, &place);
    if (rc == GUBER_OK) rc = guber_placement_export(place, &rule);
    if (rc == GUBER_OK) { rule.global_engine = -1; rc = guber_front_create(eng, 2, &rule, 1024, 0, &front); }
    mem = (uint8_t*)guber_alloc_pinned(4096);
    if (rc == GUBER_OK && mem) {
        uint8_t* keys = mem; uint32_t* off = (uint32_t*)(mem + 256);
        int64_t* hits = (int64_t*)(mem + 512); int64_t* limit = hits + N; int64_t* duration = limit + N;
        uint32_t* beh = (uint32_t*)(mem + 1024); uint8_t* algo = mem + 1280;
        int64_t* o_limit = (int64_t*)(mem + 2048); int64_t* o_rem = o_limit + N; int64_t* o_reset = o_rem + N;
        uint8_t* o_status = mem + 3072; uint8_t* o_err = mem + 3200;
        memset(mem, 0, 4096);
        memcpy(keys, "k_aak_bbk_aak_cck_bbk_aa", N * L);                   /* a b a c b a */
        for (i = 0; i <= N; ++i) off[i] = (uint32_t)(i * L);
        for (i = 0; i < N; ++i) { hits[i] = 1; limit[i] = 2; duration[i] = 60000; }
        memset(&gen, 0, sizeof gen); memset(&res, 0, sizeof res);
        gen.n = N; gen.key_bytes = keys; gen.key_off = off; gen.hits = hits; gen.limit = limit; gen.duration = duration;
        gen.behavior = beh; gen.algorithm = algo; gen.now_ms = 1700000000000LL;
        res.status = o_status; res.limit = o_limit; res.remaining = o_rem; res.reset_time = o_reset; res.err = o_err;
        rc = guber_front_eval_dev(front, &gen, &res, 1, NULL);
        if (rc == GUBER_OK) rc = guber_front_synchronize(front);
        if (rc == GUBER_OK) rc = guber_front_stats(front, &fst);
        /* key a: 1, 0 left, then refused; b: 1, 0; c: 1 — in request order */
        if (rc == GUBER_OK && !(o_rem[0] == 1 && o_rem[2] == 0 && o_status[5] == GUBER_STATUS_OVER_LIMIT && o_rem[1] == 1 && o_rem[4] == 0 && o_rem[3] == 1 &&
                                o_status[0] == GUBER_STATUS_UNDER_LIMIT && fst.generations == 1)) {
            printf("front: answers out of order: %lld %lld %lld %lld %lld status[5] %d\n", (long long)o_rem[0], (long long)o_rem[1], (long long)o_rem[2], (long long)o_rem[3], (long long)o_rem[4], o_status[5]);
            rc = GUBER_E_INVALID_ARG;
        }
    }
    if (rc != GUBER_OK) printf("front: %s (%s)\n", guber_strerror(rc), guber_last_error());
    guber_front_destroy(front);
    guber_free_pinned(mem);
    guber_placement_destroy(place);
    if (eng[1]) guber_engine_destroy(eng[1]);
    if (eng[0]) guber_engine_destroy(eng[0]);
    return rc;
}

/* go/wire_server.go's calls: the payload stage over two tables; one GetRateLimitsReq of three requests (a, b, a: limit 2) handed over as bytes, the
 * GetRateLimitsResp bytes checked field by field */
static int wire_pool_sequence(void) {
    /* RateLimitReq{name "n", unique_key "a"|"b", hits 1, limit 2, duration 60000}: 0a <len> { 0a 01 6e  12 01 61  18 01  20 02  28 e0 d4 03 } */
    static const uint8_t req[] = {0x0a, 14, 0x0a, 1, 'n', 0x12, 1, 'a', 0x18, 1, 0x20, 2, 0x28, 0xe0, 0xd4, 0x03,
                                  0x0a, 14, 0x0a, 1, 'n', 0x12, 1, 'b', 0x18, 1, 0x20, 2, 0x28, 0xe0, 0xd4, 0x03,
                                  0x0a, 14, 0x0a, 1, 'n', 0x12, 1, 'a', 0x18, 1, 0x20, 2, 0x28, 0xe0, 0xd4, 0x03};
    guber_config_t cfg;
    guber_engine_t* eng[2] = {NULL, NULL};
    guber_placement_t* place = NULL;
    guber_wire_pool_t* pool = NULL;
    guber_wire_pool_config_t wc;
    guber_wire_pool_stats_t st;
    struct guber_route_rule rule;
    uint8_t resp[2048];
    size_t n = 0, bound;
    int rc;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg; cfg.cache_size = 1000; cfg.max_batch = 1024;
    rc = guber_engine_create(&cfg, &eng[0]);
    if (rc != GUBER_OK) return rc;
    cfg.stream = guber_engine_stream(eng[0]);
    rc = guber_engine_create(&cfg, &eng[1]);
    if (rc == GUBER_OK) rc = guber_placement_create(2, 0, &place);
    if (rc == GUBER_OK) rc = guber_placement_export(place, &rule);
    memset(&wc, 0, sizeof wc);
    wc.stages = 2; wc.max_items = 1024; wc.max_payload_bytes = 1u << 16; wc.max_rpcs = 8;
    if (rc == GUBER_OK) { rule.global_engine = -1; rc = guber_wire_pool_create(eng, 2, &rule, &wc, &pool); }
    if (rc == GUBER_OK) rc = guber_wire_pool_set_clock(pool, 1700000000000LL);
    bound = guber_wire_pool_response_bound(req, sizeof req);
    if (rc == GUBER_OK && bound > sizeof resp) rc = GUBER_E_NOMEM;
    if (rc == GUBER_OK) rc = guber_wire_pool_get_rate_limits(pool, req, sizeof req, 1, 1, resp, sizeof resp, &n);
    if (rc == GUBER_OK) rc = guber_wire_pool_stats(pool, &st);
    if (rc == GUBER_OK) {
        /* a: limit 2, remaining 1 | b: limit 2, remaining 1 | a: limit 2 (remaining 0 is not written): reset_time 1700000060000 = e0 a4 99 ff bc 31 */
        static const uint8_t want[] = {0x0a, 11, 0x10, 2, 0x18, 1, 0x20, 0xe0, 0xa4, 0x99, 0xff, 0xbc, 0x31,
                                       0x0a, 11, 0x10, 2, 0x18, 1, 0x20, 0xe0, 0xa4, 0x99, 0xff, 0xbc, 0x31,
                                       0x0a, 9, 0x10, 2, 0x20, 0xe0, 0xa4, 0x99, 0xff, 0xbc, 0x31};
        if (n != sizeof want || memcmp(resp, want, n) != 0 || st.rpcs != 1 || st.items != 3) {
            size_t i;
            printf("wire pool: unexpected response (%u bytes, rpcs %llu items %llu):", (unsigned)n, (unsigned long long)st.rpcs, (unsigned long long)st.items);
            for (i = 0; i < n; ++i) printf(" %02x", resp[i]);
            printf("\n");
            rc = GUBER_E_INVALID_ARG;
        }
    }
    if (rc != GUBER_OK) printf("wire pool: %s (%s)\n", guber_strerror(rc), guber_last_error());
    guber_wire_pool_destroy(pool);
    guber_placement_destroy(place);
    if (eng[1]) guber_engine_destroy(eng[1]);
    if (eng[0]) guber_engine_destroy(eng[0]);
    return rc;
}

int main(int argc, char** argv) {
    int rc = call_sequence(argc > 1 && !strcmp(argv[1], "--gpu"));
    if (rc == GUBER_OK && argc > 1 && !strcmp(argv[1], "--gpu")) rc = front_sequence();
    if (rc == GUBER_OK && argc > 1 && !strcmp(argv[1], "--gpu")) rc = wire_pool_sequence();
    (void)argv;
    if (argc > 1) return rc == GUBER_OK ? 0 : 1;
    return rc == GUBER_E_NO_DEVICE ? 0 : 2;       /* no GPU: the product fails loudly, it has no CPU path */
}
