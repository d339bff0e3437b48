/*
 * guber_oracle.c — CPU ORACLE for the gubernator rate-limit hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load it; the product path (gubernator_amd/, the C ABI in
 * include/guber_gpu.h) never links, imports or executes anything in oracle/.
 *
 * It restates, in plain C and in the reference's own (sequential, per-request) structure, the
 * algorithm of mailgun/gubernator v2:
 *     algorithms.go:37-493   tokenBucket / tokenBucketNewItem / leakyBucket / leakyBucketNewItem
 *     cache.go:43-57         CacheItem.IsExpired
 *     lrucache.go:88-171     LRUCache Add / GetItem / Remove / removeOldest / UpdateExpiration / Size
 *     workers.go:153-184     ComputeHash63 / getWorker (key -> worker shard)
 *     workers.go:293-324     handleGetRateLimit (algorithm switch, invalid algorithm error)
 *     store.go:29-43         TokenBucketItem / LeakyBucketItem
 *     gubernator.go:425-459  UpdatePeerGlobals item construction (see oracle_add_item callers)
 *     interval.go:84-148     GregorianDuration / GregorianExpiration (UTC)
 *     replicated_hash.go:78-119  ring construction and lookup
 * Each function cites the lines it follows.
 *
 * Parity pinning: the Go toolchain is absent in this image, so the reference itself cannot be
 * run.  The oracle is pinned instead against every golden vector the reference's own tests hold
 * for this path (the JSON files under tests/golden, transcribed from functional_test.go, store_test.go,
 * interval_test.go, replicated_hash_test.go, workers_internal_test.go; see
 * tests/test_oracle_golden.py).  Third-party arithmetic restated from the published algorithms:
 * XXH64 (github.com/OneOfOne/xxhash v1.2.8), FNV-1/FNV-1a 64 (github.com/segmentio/fasthash
 * v1.0.2), MD5 (Go crypto/md5, RFC 1321).
 *
 * Go semantics emulated: int64 wrap-around (built with -fwrapv), float64->int64 conversion as
 * amd64 CVTTSD2SQ (NaN / +-Inf / out of range -> INT64_MIN), int64->float64 round-to-nearest-even.
 */
#define _GNU_SOURCE   /* sched_getaffinity / pthread_setaffinity_np: the CPU-baseline harness pins its threads */
#include "guber_oracle.h"

#include <math.h>
#include <pthread.h>
#include <limits.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <sched.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Go numeric semantics
 * ---------------------------------------------------------------------------------------- */
static inline int64_t go_f2i(double d) {
    /* amd64 CVTTSD2SQ: out of range or NaN gives the "integer indefinite" 0x8000000000000000 */
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)d;
}

/* ------------------------------------------------------------------------------------------
 * XXH64 — restated from the published xxHash specification (seeded, little-endian lanes).
 * Reference call site: xxhash.ChecksumString64S(input, 0) >> 1, workers.go:153-155.
 * ---------------------------------------------------------------------------------------- */
#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) { return rotl64(acc + in * XP2, 31) * XP1; }
static inline uint64_t xmerge(uint64_t acc, uint64_t v) { return (acc ^ xround(0, v)) * XP1 + XP4; }

uint64_t oracle_xxhash64(const uint8_t* p, size_t len, uint64_t seed) {
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t* lim = end - 32;
        do {
            v1 = xround(v1, rd64(p));
            v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16));
            v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl64(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * XP1; h = rotl64(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * XP5; h = rotl64(h, 11) * XP1; p++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

/* FNV-1 / FNV-1a 64 — segmentio/fasthash: offset 0xcbf29ce484222325, prime 0x100000001b3.
 * Reference call sites: replicated_hash.go:33,81-84,108; config.go:429-433. */
uint64_t oracle_fnv1_64(const uint8_t* p, size_t len) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < len; i++) { h *= 0x100000001b3ULL; h ^= p[i]; }
    return h;
}
uint64_t oracle_fnv1a_64(const uint8_t* p, size_t len) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < len; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

/* MD5 (RFC 1321) — Go crypto/md5, used for the vnode labels at replicated_hash.go:81. */
static const uint32_t MD5_K[64] = {
    0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,
    0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
    0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,
    0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
    0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,
    0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
    0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,
    0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391};
static const uint8_t MD5_S[64] = {7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22,
                                  5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
                                  4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23,
                                  6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21};
void oracle_md5(const uint8_t* msg, size_t len, uint8_t out[16]) {
    uint32_t a0 = 0x67452301, b0 = 0xefcdab89, c0 = 0x98badcfe, d0 = 0x10325476;
    size_t padded = ((len + 8) / 64 + 1) * 64;
    uint8_t* buf = (uint8_t*)calloc(padded, 1);
    memcpy(buf, msg, len);
    buf[len] = 0x80;
    uint64_t bits = (uint64_t)len * 8;
    memcpy(buf + padded - 8, &bits, 8);
    for (size_t off = 0; off < padded; off += 64) {
        uint32_t M[16];
        memcpy(M, buf + off, 64);
        uint32_t A = a0, B = b0, C = c0, D = d0;
        for (int i = 0; i < 64; i++) {
            uint32_t F; int g;
            if (i < 16) { F = (B & C) | (~B & D); g = i; }
            else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { F = C ^ (B | ~D); g = (7 * i) & 15; }
            F = F + A + MD5_K[i] + M[g];
            A = D; D = C; C = B;
            B = B + ((F << MD5_S[i]) | (F >> (32 - MD5_S[i])));
        }
        a0 += A; b0 += B; c0 += C; d0 += D;
    }
    free(buf);
    memcpy(out, &a0, 4); memcpy(out + 4, &b0, 4); memcpy(out + 8, &c0, 4); memcpy(out + 12, &d0, 4);
}

/* ------------------------------------------------------------------------------------------
 * Cache item + LRU cache (lrucache.go:32-171): a string-keyed hash map plus a doubly linked
 * recency list, exactly the reference's structure.  Not thread-safe (lrucache.go:30-31); one
 * cache per worker (workers.go:163-177).
 * ---------------------------------------------------------------------------------------- */
enum { VK_NIL = 0, VK_TOKEN = 1, VK_LEAKY = 2 };

typedef struct citem {
    /* CacheItem (cache.go:29-41) */
    int32_t algorithm;
    char* key; uint32_t klen;
    int vkind;          /* dynamic type of CacheItem.Value */
    int64_t expire_at, invalid_at;
    /* TokenBucketItem (store.go:37-43) */
    int32_t t_status; int64_t t_limit, t_duration, t_remaining, t_created_at;
    /* LeakyBucketItem (store.go:29-35) */
    int64_t l_limit, l_duration; double l_remaining; int64_t l_updated_at, l_burst;
    /* map chain + list links */
    uint64_t h;
    struct citem* hnext;
    struct citem *prev, *next; /* list: head = most recently used */
} citem_t;

typedef struct lru {
    citem_t** buckets; uint64_t nbuckets; /* power of two */
    citem_t *head, *tail;
    int64_t len;
    int64_t cache_size;
} lru_t;

struct oracle {
    lru_t* workers; uint32_t nworkers; uint64_t ring_step;
    uint64_t over_limit, hits, misses, unexpired_evictions;
    /* Config.Store (store.go:49-65) for oracle_eval_batch_store; NULL = no store (the `s != nil` tests) */
    const oracle_store_t* store; uint32_t cur_req;
    struct mt_pool* mt;   /* persistent threads of oracle_eval_batch_mt (CPU baseline harness) */
};

static void lru_init(lru_t* c, int64_t cache_size) {
    memset(c, 0, sizeof(*c));
    c->nbuckets = 1024;
    c->buckets = (citem_t**)calloc(c->nbuckets, sizeof(citem_t*));
    c->cache_size = cache_size;
}
static void lru_grow(lru_t* c) {
    uint64_t nb = c->nbuckets * 2;
    citem_t** b = (citem_t**)calloc(nb, sizeof(citem_t*));
    for (uint64_t i = 0; i < c->nbuckets; i++) {
        citem_t* e = c->buckets[i];
        while (e) { citem_t* n = e->hnext; uint64_t j = e->h & (nb - 1); e->hnext = b[j]; b[j] = e; e = n; }
    }
    free(c->buckets); c->buckets = b; c->nbuckets = nb;
}
static citem_t* lru_find(lru_t* c, const char* key, uint32_t klen, uint64_t h) {
    for (citem_t* e = c->buckets[h & (c->nbuckets - 1)]; e; e = e->hnext)
        if (e->h == h && e->klen == klen && memcmp(e->key, key, klen) == 0) return e;
    return NULL;
}
static void list_unlink(lru_t* c, citem_t* e) {
    if (e->prev) e->prev->next = e->next; else c->head = e->next;
    if (e->next) e->next->prev = e->prev; else c->tail = e->prev;
    e->prev = e->next = NULL;
}
static void list_push_front(lru_t* c, citem_t* e) {
    e->prev = NULL; e->next = c->head;
    if (c->head) c->head->prev = e; else c->tail = e;
    c->head = e;
}
static void list_move_front(lru_t* c, citem_t* e) { if (c->head != e) { list_unlink(c, e); list_push_front(c, e); } }

/* lrucache.go:151-156 removeElement */
static void lru_remove_element(lru_t* c, citem_t* e) {
    list_unlink(c, e);
    citem_t** pp = &c->buckets[e->h & (c->nbuckets - 1)];
    while (*pp != e) pp = &(*pp)->hnext;
    *pp = e->hnext;
    c->len--;
    free(e->key); free(e);
}
/* lrucache.go:138-149 removeOldest */
static void lru_remove_oldest(oracle_t* o, lru_t* c, int64_t now) {
    citem_t* e = c->tail;
    if (e) {
        if (now < e->expire_at) o->unexpired_evictions++;
        lru_remove_element(c, e);
    }
}
/* cache.go:43-57 IsExpired */
static int item_is_expired(const citem_t* it, int64_t now) {
    if (it->invalid_at != 0 && it->invalid_at < now) return 1;
    if (it->expire_at < now) return 1;
    return 0;
}
/* lrucache.go:111-128 GetItem */
static citem_t* lru_get_item(oracle_t* o, lru_t* c, const char* key, uint32_t klen, uint64_t h, int64_t now) {
    citem_t* e = lru_find(c, key, klen, h);
    if (e) {
        if (item_is_expired(e, now)) { lru_remove_element(c, e); o->misses++; return NULL; }
        o->hits++;
        list_move_front(c, e);
        return e;
    }
    o->misses++;
    return NULL;
}
/* lrucache.go:88-103 Add: the new item REPLACES the element's value if the key exists (returns 1),
 * else pushes front and evicts the oldest when over cacheSize.  `src` is copied. */
static int lru_add(oracle_t* o, lru_t* c, const citem_t* src, const char* key, uint32_t klen, uint64_t h,
                   int64_t now, citem_t** out) {
    citem_t* e = lru_find(c, key, klen, h);
    if (e) {
        list_move_front(c, e);
        char* k = e->key; citem_t *hn = e->hnext, *p = e->prev, *n = e->next;
        *e = *src; e->key = k; e->klen = klen; e->h = h; e->hnext = hn; e->prev = p; e->next = n;
        if (out) *out = e;
        return 1;
    }
    e = (citem_t*)malloc(sizeof(citem_t));
    *e = *src;
    e->key = (char*)malloc(klen ? klen : 1); memcpy(e->key, key, klen); e->klen = klen; e->h = h;
    if ((uint64_t)c->len * 2 > c->nbuckets) lru_grow(c);
    uint64_t j = h & (c->nbuckets - 1);
    e->hnext = c->buckets[j]; c->buckets[j] = e;
    list_push_front(c, e);
    c->len++;
    if (out) *out = e;
    if (c->cache_size != 0 && c->len > c->cache_size) lru_remove_oldest(o, c, now);
    return 0;
}
/* lrucache.go:131-135 Remove */
static void lru_remove(lru_t* c, const char* key, uint32_t klen, uint64_t h) {
    citem_t* e = lru_find(c, key, klen, h);
    if (e) lru_remove_element(c, e);
}

/* ------------------------------------------------------------------------------------------
 * Requests / responses
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const char* key; uint32_t klen; uint64_t h;
    int64_t hits, limit, duration, burst, created_at;
    uint32_t algorithm, behavior; int is_owner;
    int64_t greg_expire, greg_duration; /* host-precomputed interval.go values; greg_duration<0 = error */
} req_t;
typedef struct { uint8_t status; int64_t limit, remaining, reset_time; uint8_t err; } resp_t;

#define HAS(b, f) (((b) & (f)) != 0) /* gubernator.go:776-779 HasBehavior */

static int greg_error(const req_t* r) { return r->greg_duration < 0 ? (int)(-r->greg_duration) : 0; }

static void item_from_abi(const guber_item_t* in, citem_t* it);
static void item_to_abi(const citem_t* it, guber_item_t* out);
/* `s.OnChange(ctx, r, item)`: the store sees the item as it is when the call is made */
static void store_on_change(oracle_t* o, const req_t* r, const citem_t* item) {
    if (!o->store || !r->is_owner || !o->store->on_change) return;
    guber_item_t out; item_to_abi(item, &out);
    out.key = (const uint8_t*)r->key; out.key_len = r->klen;
    o->store->on_change(o->store->user, o->cur_req, &out);
}
static void store_remove(oracle_t* o, const req_t* r) {
    if (o->store && o->store->remove) o->store->remove(o->store->user, o->cur_req, (const uint8_t*)r->key, r->klen);
}
/* algorithms.go:45-51 / :274-280: on a cache miss ask the store; a returned item is added to the cache */
static citem_t* store_get(oracle_t* o, lru_t* c, const req_t* r, int64_t now) {
    if (!o->store || !o->store->get) return NULL;
    guber_item_t in; memset(&in, 0, sizeof(in));
    if (!o->store->get(o->store->user, o->cur_req, &in)) return NULL;
    citem_t it; item_from_abi(&in, &it);
    citem_t* e = NULL;
    lru_add(o, c, &it, r->key, r->klen, r->h, now, &e);
    return e;
}

/* algorithms.go:206-257 tokenBucketNewItem */
static int token_bucket_new_item(oracle_t* o, lru_t* c, const req_t* r, int64_t now, resp_t* rl) {
    int64_t created_at = r->created_at;
    int64_t expire = created_at + r->duration;                      /* :208 */
    citem_t it; memset(&it, 0, sizeof(it));
    it.vkind = VK_TOKEN;
    it.t_limit = r->limit; it.t_duration = r->duration;             /* :210-215 */
    it.t_remaining = r->limit - r->hits; it.t_created_at = created_at;
    if (HAS(r->behavior, GUBER_BEHAVIOR_DURATION_IS_GREGORIAN)) {   /* :218-223 */
        int ge = greg_error(r);
        if (ge) { rl->err = (uint8_t)ge; return -1; }
        expire = r->greg_expire;
    }
    it.algorithm = GUBER_ALGO_TOKEN_BUCKET; it.expire_at = expire;  /* :225-230 */
    rl->status = GUBER_STATUS_UNDER_LIMIT; rl->limit = r->limit;    /* :232-237 */
    rl->remaining = it.t_remaining; rl->reset_time = expire;
    if (r->hits > r->limit) {                                       /* :240-248 */
        if (r->is_owner) o->over_limit++;
        rl->status = GUBER_STATUS_OVER_LIMIT;
        rl->remaining = r->limit;
        it.t_remaining = r->limit;
    }
    citem_t* added = NULL;
    lru_add(o, c, &it, r->key, r->klen, r->h, now, &added);          /* :250 */
    store_on_change(o, r, added);                                   /* :252-254 */
    return 0;
}

/* algorithms.go:37-203 tokenBucket (o->store == NULL is the reference's `s == nil`) */
static int token_bucket(oracle_t* o, lru_t* c, const req_t* r, int64_t now, resp_t* rl) {
    citem_t* item = lru_get_item(o, c, r->key, r->klen, r->h, now); /* :43 */
    int ok = item != NULL;
    if (!ok && (item = store_get(o, c, r, now)) != NULL) ok = 1;    /* :45-51 */
    if (ok && item->vkind == VK_NIL) ok = 0;                        /* :55-63 Value is nil */
    if (ok) {
        if (HAS(r->behavior, GUBER_BEHAVIOR_RESET_REMAINING)) {     /* :78-90 */
            lru_remove(c, r->key, r->klen, r->h);
            store_remove(o, r);                                     /* :81-83 */
            rl->status = GUBER_STATUS_UNDER_LIMIT; rl->limit = r->limit;
            rl->remaining = r->limit; rl->reset_time = 0;
            return 0;
        }
        if (item->vkind != VK_TOKEN) {                              /* :91-103 switched algorithms */
            lru_remove(c, r->key, r->klen, r->h);
            store_remove(o, r);                                     /* :98-100 */
            return token_bucket_new_item(o, c, r, now, rl);
        }
        citem_t* t = item;
        if (t->t_limit != r->limit) {                               /* :106-113 */
            t->t_remaining += r->limit - t->t_limit;
            if (t->t_remaining < 0) t->t_remaining = 0;
            t->t_limit = r->limit;
        }
        rl->status = (uint8_t)t->t_status; rl->limit = r->limit;    /* :115-120 */
        rl->remaining = t->t_remaining; rl->reset_time = item->expire_at;
        if (t->t_duration != r->duration) {                         /* :123-147 */
            int64_t expire = t->t_created_at + r->duration;
            if (HAS(r->behavior, GUBER_BEHAVIOR_DURATION_IS_GREGORIAN)) {
                int ge = greg_error(r);
                if (ge) { memset(rl, 0, sizeof(*rl)); rl->err = (uint8_t)ge; return -1; }
                expire = r->greg_expire;
            }
            int64_t created_at = r->created_at;
            if (expire <= created_at) {                             /* :136-142 renew */
                expire = created_at + r->duration;
                t->t_created_at = created_at;
                t->t_remaining = t->t_limit;
            }
            item->expire_at = expire;
            t->t_duration = r->duration;
            rl->reset_time = expire;
        }
        /* :149-153 `defer s.OnChange(ctx, r, item)`: runs at every return below */
        if (r->hits == 0) goto found_done;                                 /* :157-159 */
        if (rl->remaining == 0 && r->hits > 0) {                    /* :162-170 */
            if (r->is_owner) o->over_limit++;
            rl->status = GUBER_STATUS_OVER_LIMIT;
            t->t_status = rl->status;
            goto found_done;
        }
        if (t->t_remaining == r->hits) {                            /* :173-178 */
            t->t_remaining = 0; rl->remaining = 0;
            goto found_done;
        }
        if (r->hits > t->t_remaining) {                             /* :182-194 */
            if (r->is_owner) o->over_limit++;
            rl->status = GUBER_STATUS_OVER_LIMIT;
            if (HAS(r->behavior, GUBER_BEHAVIOR_DRAIN_OVER_LIMIT)) { t->t_remaining = 0; rl->remaining = 0; }
            goto found_done;
        }
        t->t_remaining -= r->hits;                                  /* :196-198 */
        rl->remaining = t->t_remaining;
        goto found_done;
    found_done:
        store_on_change(o, r, item);
        return 0;
    }
    return token_bucket_new_item(o, c, r, now, rl);                 /* :202 */
}

/* algorithms.go:437-493 leakyBucketNewItem.  `burst` is r.Burst after the :264 defaulting. */
static int leaky_bucket_new_item(oracle_t* o, lru_t* c, const req_t* r, int64_t burst, int64_t now, resp_t* rl) {
    int64_t created_at = r->created_at;
    int64_t duration = r->duration;
    double rate = (double)duration / (double)r->limit;              /* :440 */
    if (HAS(r->behavior, GUBER_BEHAVIOR_DURATION_IS_GREGORIAN)) {   /* :441-450 */
        int ge = greg_error(r);
        if (ge) { rl->err = (uint8_t)ge; return -1; }
        duration = r->greg_expire - now;   /* expire - n.UnixNano()/1e6; n = clock.Now() = the batch clock */
    }
    citem_t it; memset(&it, 0, sizeof(it));
    it.vkind = VK_LEAKY;
    it.l_remaining = (double)(burst - r->hits);                     /* :453-459 */
    it.l_limit = r->limit; it.l_duration = duration; it.l_updated_at = created_at; it.l_burst = burst;
    rl->status = GUBER_STATUS_UNDER_LIMIT; rl->limit = it.l_limit;  /* :461-466 */
    rl->remaining = burst - r->hits;
    rl->reset_time = created_at + (it.l_limit - (burst - r->hits)) * go_f2i(rate);
    if (r->hits > burst) {                                          /* :469-477 */
        if (r->is_owner) o->over_limit++;
        rl->status = GUBER_STATUS_OVER_LIMIT;
        rl->remaining = 0;
        rl->reset_time = created_at + (rl->limit - rl->remaining) * go_f2i(rate);
        it.l_remaining = 0;
    }
    it.expire_at = created_at + duration;                           /* :479-484 */
    it.algorithm = (int32_t)r->algorithm;
    citem_t* added = NULL;
    lru_add(o, c, &it, r->key, r->klen, r->h, now, &added);          /* :486 */
    store_on_change(o, r, added);                                   /* :488-490 */
    return 0;
}

/* algorithms.go:260-434 leakyBucket */
static int leaky_bucket(oracle_t* o, lru_t* c, const req_t* r, int64_t now, resp_t* rl) {
    int64_t burst = r->burst;
    if (burst == 0) burst = r->limit;                               /* :264-266 */
    int64_t created_at = r->created_at;
    citem_t* item = lru_get_item(o, c, r->key, r->klen, r->h, now); /* :272 */
    int ok = item != NULL;
    if (!ok && (item = store_get(o, c, r, now)) != NULL) ok = 1;    /* :274-280 */
    if (ok && item->vkind == VK_NIL) ok = 0;                        /* :284-292 */
    if (ok) {
        if (item->vkind != VK_LEAKY) {                              /* :308-318 */
            lru_remove(c, r->key, r->klen, r->h);
            store_remove(o, r);                                     /* :313-315 */
            return leaky_bucket_new_item(o, c, r, burst, now, rl);
        }
        citem_t* b = item;
        if (HAS(r->behavior, GUBER_BEHAVIOR_RESET_REMAINING)) b->l_remaining = (double)burst; /* :320-322 */
        if (b->l_burst != burst) {                                  /* :325-330 */
            if (burst > go_f2i(b->l_remaining)) b->l_remaining = (double)burst;
            b->l_burst = burst;
        }
        b->l_limit = r->limit; b->l_duration = r->duration;         /* :332-333 */
        int64_t duration = r->duration;
        double rate = (double)duration / (double)r->limit;          /* :336 */
        if (HAS(r->behavior, GUBER_BEHAVIOR_DURATION_IS_GREGORIAN)) { /* :338-354 */
            int ge = greg_error(r);
            if (ge) { rl->err = (uint8_t)ge; return -1; }
            rate = (double)r->greg_duration / (double)r->limit;
            duration = r->greg_expire - now;
        }
        if (r->hits != 0) item->expire_at = created_at + duration;  /* :356-358 UpdateExpiration */
        int64_t elapsed = created_at - b->l_updated_at;             /* :361-367 */
        double leak = (double)elapsed / rate;
        if (go_f2i(leak) > 0) { b->l_remaining += leak; b->l_updated_at = created_at; }
        if (go_f2i(b->l_remaining) > b->l_burst) b->l_remaining = (double)b->l_burst; /* :369-371 */
        rl->limit = b->l_limit; rl->remaining = go_f2i(b->l_remaining);             /* :373-378 */
        rl->status = GUBER_STATUS_UNDER_LIMIT;
        rl->reset_time = created_at + (b->l_limit - go_f2i(b->l_remaining)) * go_f2i(rate);
        /* :382-386 `defer s.OnChange(ctx, r, item)`: runs at every return below */
        if (go_f2i(b->l_remaining) == 0 && r->hits > 0) {           /* :389-395 */
            if (r->is_owner) o->over_limit++;
            rl->status = GUBER_STATUS_OVER_LIMIT;
            goto found_done;
        }
        if (go_f2i(b->l_remaining) == r->hits) {                    /* :398-403 */
            b->l_remaining = 0;
            rl->remaining = go_f2i(b->l_remaining);
            rl->reset_time = created_at + (rl->limit - rl->remaining) * go_f2i(rate);
            goto found_done;
        }
        if (r->hits > go_f2i(b->l_remaining)) {                     /* :407-420 */
            if (r->is_owner) o->over_limit++;
            rl->status = GUBER_STATUS_OVER_LIMIT;
            if (HAS(r->behavior, GUBER_BEHAVIOR_DRAIN_OVER_LIMIT)) { b->l_remaining = 0; rl->remaining = 0; }
            goto found_done;
        }
        if (r->hits == 0) goto found_done;                                 /* :423-425 */
        b->l_remaining -= (double)r->hits;                          /* :427-430 */
        rl->remaining = go_f2i(b->l_remaining);
        rl->reset_time = created_at + (rl->limit - rl->remaining) * go_f2i(rate);
        goto found_done;
    found_done:
        store_on_change(o, r, item);
        return 0;
    }
    return leaky_bucket_new_item(o, c, r, burst, now, rl);          /* :433 */
}

/* workers.go:293-324 handleGetRateLimit.  On error the reference returns a nil response. */
static void handle_get_rate_limit(oracle_t* o, lru_t* c, const req_t* r, int64_t now, resp_t* rl) {
    memset(rl, 0, sizeof(*rl));
    int rc;
    switch (r->algorithm) {
    case GUBER_ALGO_TOKEN_BUCKET: rc = token_bucket(o, c, r, now, rl); break;
    case GUBER_ALGO_LEAKY_BUCKET: rc = leaky_bucket(o, c, r, now, rl); break;
    default: rl->err = GUBER_ITEM_E_INVALID_ALGORITHM; rc = -1; break; /* :317-321 */
    }
    if (rc != 0) { uint8_t e = rl->err; memset(rl, 0, sizeof(*rl)); rl->err = e; }
}

/* ------------------------------------------------------------------------------------------
 * Worker pool (workers.go:125-184)
 * ---------------------------------------------------------------------------------------- */
oracle_t* oracle_create(uint64_t cache_size, uint32_t workers) {
    if (workers == 0) workers = 1;
    if (cache_size == 0) cache_size = 50000;                        /* workers.go:126 */
    oracle_t* o = (oracle_t*)calloc(1, sizeof(oracle_t));
    o->nworkers = workers;
    o->ring_step = (1ULL << 63) / workers;                          /* workers.go:134 */
    o->workers = (lru_t*)calloc(workers, sizeof(lru_t));
    for (uint32_t i = 0; i < workers; i++) lru_init(&o->workers[i], (int64_t)(cache_size / workers)); /* :132 */
    return o;
}
struct mt_pool;
static void mt_destroy(struct mt_pool* p);
void oracle_destroy(oracle_t* o) {
    if (!o) return;
    mt_destroy(o->mt);
    for (uint32_t i = 0; i < o->nworkers; i++) {
        lru_t* c = &o->workers[i];
        citem_t* e = c->head;
        while (e) { citem_t* n = e->next; free(e->key); free(e); e = n; }
        free(c->buckets);
    }
    free(o->workers); free(o);
}
/* workers.go:180-184 getWorker: idx = (xxhash64(key) >> 1) / hashRingStep */
static inline uint32_t worker_index(const oracle_t* o, uint64_t h) {
    const uint32_t w = (uint32_t)((h >> 1) / o->ring_step);
    return w < o->nworkers ? w : o->nworkers - 1;   /* (2^63 / W) * W < 2^63 when W is not a power of two: the last worker takes the remainder */
}
uint32_t oracle_worker_index_for_hash63(uint32_t workers, uint64_t hash63) {
    return (uint32_t)(hash63 / ((1ULL << 63) / workers));
}

static void load_req_h(const guber_batch_t* b, uint32_t i, uint64_t h, req_t* r) {
    r->key = (const char*)b->key_bytes + b->key_off[i];
    r->klen = b->key_off[i + 1] - b->key_off[i];
    r->h = h;
    r->hits = b->hits[i]; r->limit = b->limit[i]; r->duration = b->duration[i];
    r->burst = b->burst ? b->burst[i] : 0;
    r->created_at = b->created_at ? b->created_at[i] : b->now_ms;
    r->algorithm = b->algorithm ? b->algorithm[i] : 0;
    r->behavior = b->behavior ? b->behavior[i] : 0;
    r->is_owner = b->is_owner ? b->is_owner[i] : 1;
    r->greg_expire = b->greg_expire ? b->greg_expire[i] : 0;
    r->greg_duration = b->greg_duration ? b->greg_duration[i] : 0;
}
static void load_req(const guber_batch_t* b, uint32_t i, req_t* r) {
    const uint8_t* k = b->key_bytes + b->key_off[i];
    load_req_h(b, i, oracle_xxhash64(k, b->key_off[i + 1] - b->key_off[i], 0), r);
}
static void store_resp(guber_result_t* res, uint32_t i, const resp_t* rl) {
    res->status[i] = rl->status; res->limit[i] = rl->limit; res->remaining[i] = rl->remaining;
    res->reset_time[i] = rl->reset_time; res->err[i] = rl->err;
}

/* The reference applies a batch's requests one by one in request order (gubernator.go:203 is a
 * serial loop and each worker is a single goroutine, workers.go:190-258). */
int oracle_eval_batch(oracle_t* o, const guber_batch_t* b, guber_result_t* res) {
    uint64_t ol0 = o->over_limit, h0 = o->hits, m0 = o->misses, ev0 = o->unexpired_evictions;
    for (uint32_t i = 0; i < b->n; i++) {
        req_t r; resp_t rl;
        load_req(b, i, &r);
        handle_get_rate_limit(o, &o->workers[worker_index(o, r.h)], &r, b->now_ms, &rl);
        store_resp(res, i, &rl);
    }
    res->over_limit_count = o->over_limit - ol0; res->cache_hits = o->hits - h0;
    res->cache_misses = o->misses - m0; res->unexpired_evictions = o->unexpired_evictions - ev0;
    res->cache_size = oracle_size(o);
    return 0;
}

/* The same loop with Config.Store set (store.go:49-65): Get on a cache miss, OnChange after an evaluated
 * request when the node owns the key, Remove when an item is dropped because of RESET_REMAINING (token) or an
 * algorithm switch.  The callbacks receive the index of the request that caused them. */
int oracle_eval_batch_store(oracle_t* o, const guber_batch_t* b, guber_result_t* res, const oracle_store_t* st) {
    o->store = st;
    uint64_t ol0 = o->over_limit, h0 = o->hits, m0 = o->misses, ev0 = o->unexpired_evictions;
    for (uint32_t i = 0; i < b->n; i++) {
        req_t r; resp_t rl;
        load_req(b, i, &r);
        o->cur_req = i;
        handle_get_rate_limit(o, &o->workers[worker_index(o, r.h)], &r, b->now_ms, &rl);
        store_resp(res, i, &rl);
    }
    o->store = NULL;
    res->over_limit_count = o->over_limit - ol0; res->cache_hits = o->hits - h0;
    res->cache_misses = o->misses - m0; res->unexpired_evictions = o->unexpired_evictions - ev0;
    res->cache_size = oracle_size(o);
    return 0;
}

/* "Reference design" multi-core evaluation for the CPU baseline: requests are routed to their
 * worker shard (workers.go:180-184) and every worker applies its own requests in request order on
 * its own thread, as the reference's worker goroutines do.  Results are identical to
 * oracle_eval_batch because shards are disjoint.
 *
 * The harness keeps `threads` persistent worker threads per oracle (created on first use, pinned round-robin to the
 * CPUs the process may run on when there are enough of them), hands a batch over with one generation counter and runs
 * it in phases separated by barriers (a waiter looks briefly, then sleeps on a futex): (1) every thread hashes a contiguous slice of the batch and
 * counts its requests per worker; (2) per-worker totals and the slices' offsets (a parallel stable counting sort:
 * inside a worker the request order is kept); (3) every thread applies the queues of the workers it owns (worker w
 * belongs to thread w mod T).  Nothing is allocated per batch once the buffers have grown to the batch size. */
struct mt_arg { struct mt_pool* p; int t; };
struct mt_pool {
    oracle_t* o; int T;
    pthread_t* th;
    volatile uint32_t gen, stop;               /* gen: a batch is posted (threads spin, then nap) */
    volatile uint32_t bar_count, bar_gen, done;
    uint32_t spin_limit;                        /* how long a waiter looks before it sleeps (0 with more threads than usable CPUs) */
    const guber_batch_t* b; guber_result_t* res;
    uint64_t* hashes; uint32_t *widx, *order, *hist /* [T][W+1] */, *start /* [W+1] */; uint32_t cap;
    uint64_t* ctr;                              /* [W][4] per-batch counters */
    struct mt_arg* args;
};
/* Waiting: look at the word for a moment, then sleep on it (futex).  With more threads than the CPUs this process may really
 * use — a cgroup CPU quota counts — spinning waiters only get the whole group throttled, so they sleep at once. */
static void mt_fwait(volatile uint32_t* w, uint32_t seen) { syscall(SYS_futex, w, FUTEX_WAIT_PRIVATE, seen, NULL, NULL, 0); }
static void mt_fwake(volatile uint32_t* w) { syscall(SYS_futex, w, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0); }
static void mt_wait_change(struct mt_pool* p, volatile uint32_t* w, uint32_t seen) {
    for (uint32_t spins = 0; spins < p->spin_limit; spins++) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) != seen) return;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    while (__atomic_load_n(w, __ATOMIC_ACQUIRE) == seen) mt_fwait(w, seen);
}
static void mt_barrier(struct mt_pool* p) {
    const uint32_t g = __atomic_load_n(&p->bar_gen, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&p->bar_count, 1, __ATOMIC_ACQ_REL) == (uint32_t)p->T) {
        __atomic_store_n(&p->bar_count, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&p->bar_gen, g + 1, __ATOMIC_SEQ_CST);
        mt_fwake(&p->bar_gen);
        return;
    }
    mt_wait_change(p, &p->bar_gen, g);
}
static void mt_run_batch(struct mt_pool* p, int t) {
    oracle_t* o = p->o;
    const guber_batch_t* b = p->b; guber_result_t* res = p->res;
    const uint32_t W = o->nworkers, n = b->n, T = (uint32_t)p->T;
    const uint32_t lo = (uint32_t)((uint64_t)n * (uint32_t)t / T), hi = (uint32_t)((uint64_t)n * ((uint32_t)t + 1) / T);
    uint32_t* myhist = p->hist + (size_t)t * (W + 1);
    memset(myhist, 0, sizeof(uint32_t) * (W + 1));
    for (uint32_t i = lo; i < hi; i++) {
        const uint8_t* k = b->key_bytes + b->key_off[i];
        const uint64_t h = oracle_xxhash64(k, b->key_off[i + 1] - b->key_off[i], 0);
        const uint32_t w = worker_index(o, h);
        p->hashes[i] = h; p->widx[i] = w;
        myhist[w]++;
    }
    mt_barrier(p);
    /* for the workers this thread owns: the slices' offsets inside the worker's queue (exclusive scan over the threads) */
    for (uint32_t w = (uint32_t)t; w < W; w += T) {
        uint32_t acc = 0;
        for (uint32_t q = 0; q < T; q++) { uint32_t* hq = p->hist + (size_t)q * (W + 1) + w; const uint32_t c = *hq; *hq = acc; acc += c; }
        p->start[w + 1] = acc;                                      /* total of worker w; made a prefix by thread 0 below */
    }
    mt_barrier(p);
    if (t == 0) { p->start[0] = 0; for (uint32_t w = 0; w < W; w++) p->start[w + 1] += p->start[w]; }
    mt_barrier(p);
    for (uint32_t i = lo; i < hi; i++) { const uint32_t w = p->widx[i]; p->order[p->start[w] + myhist[w]++] = i; }
    mt_barrier(p);
    for (uint32_t w = (uint32_t)t; w < W; w += T) {
        oracle_t local = *o; /* private counters; caches are disjoint per worker */
        local.over_limit = local.hits = local.misses = local.unexpired_evictions = 0;
        for (uint32_t q = p->start[w]; q < p->start[w + 1]; q++) {
            const uint32_t i = p->order[q];
            req_t r; resp_t rl;
            load_req_h(b, i, p->hashes[i], &r);
            handle_get_rate_limit(&local, &o->workers[w], &r, b->now_ms, &rl);
            store_resp(res, i, &rl);
        }
        p->ctr[w * 4 + 0] = local.over_limit; p->ctr[w * 4 + 1] = local.hits;
        p->ctr[w * 4 + 2] = local.misses; p->ctr[w * 4 + 3] = local.unexpired_evictions;
    }
    mt_barrier(p);
}
static void* mt_main(void* arg) {
    struct mt_pool* p = ((struct mt_arg*)arg)->p; const int t = ((struct mt_arg*)arg)->t;
    uint32_t seen = 0;
    for (;;) {
        mt_wait_change(p, &p->gen, seen);                            /* a batch is posted, or the pool is being destroyed */
        if (__atomic_load_n(&p->stop, __ATOMIC_ACQUIRE)) return NULL;
        seen = __atomic_load_n(&p->gen, __ATOMIC_ACQUIRE);
        mt_run_batch(p, t);
        __atomic_add_fetch(&p->done, 1, __ATOMIC_SEQ_CST);
        mt_fwake(&p->done);
    }
}
static void mt_destroy(struct mt_pool* p) {
    if (!p) return;
    __atomic_store_n(&p->stop, 1, __ATOMIC_RELEASE);
    __atomic_add_fetch(&p->gen, 1, __ATOMIC_SEQ_CST);
    mt_fwake(&p->gen);
    for (int t = 1; t < p->T; t++) pthread_join(p->th[t], NULL);
    free(p->th); free(p->args); free(p->hashes); free(p->widx); free(p->order); free(p->hist); free(p->start); free(p->ctr);
    free(p);
}
static struct mt_pool* mt_create(oracle_t* o, int T) {
    struct mt_pool* p = (struct mt_pool*)calloc(1, sizeof(*p));
    p->o = o; p->T = T;
    p->th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    p->args = (struct mt_arg*)calloc((size_t)T, sizeof(*p->args));
    p->hist = (uint32_t*)calloc((size_t)T * (o->nworkers + 1), sizeof(uint32_t));
    p->start = (uint32_t*)calloc(o->nworkers + 1, sizeof(uint32_t));
    p->ctr = (uint64_t*)calloc((size_t)o->nworkers * 4, sizeof(uint64_t));
    cpu_set_t allowed; int ncpu = 0; static int cpus[CPU_SETSIZE];
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
    int usable = ncpu > 0 ? ncpu : 1;                               /* CPUs really available: a cgroup quota counts */
    {
        FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
        if (f) {
            char q[32] = {0}; unsigned long long period = 0;
            if (fscanf(f, "%31s %llu", q, &period) == 2 && period && strcmp(q, "max") != 0) {
                const unsigned long long quota = strtoull(q, NULL, 10);
                if (quota && (int)((quota + period - 1) / period) < usable) usable = (int)((quota + period - 1) / period);
            }
            fclose(f);
        }
    }
    p->spin_limit = T <= usable ? 4000u : 0u;
    for (int t = 1; t < T; t++) {                                   /* thread 0 is the caller */
        p->args[t].p = p; p->args[t].t = t;
        pthread_create(&p->th[t], NULL, mt_main, &p->args[t]);
        if (ncpu >= T) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[t % ncpu], &one); pthread_setaffinity_np(p->th[t], sizeof(one), &one); }
    }
    return p;
}
int oracle_eval_batch_mt(oracle_t* o, const guber_batch_t* b, guber_result_t* res, int threads) {
    const uint32_t W = o->nworkers, n = b->n;
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > W) threads = (int)W;                    /* a worker is single-threaded (workers.go:190-258) */
    if (o->mt && o->mt->T != threads) { mt_destroy(o->mt); o->mt = NULL; }
    if (!o->mt) o->mt = mt_create(o, threads);
    struct mt_pool* p = o->mt;
    if (n > p->cap) {
        free(p->hashes); free(p->widx); free(p->order);
        p->cap = n + n / 4 + 16;
        p->hashes = (uint64_t*)malloc(sizeof(uint64_t) * p->cap);
        p->widx = (uint32_t*)malloc(sizeof(uint32_t) * p->cap);
        p->order = (uint32_t*)malloc(sizeof(uint32_t) * p->cap);
    }
    p->b = b; p->res = res;
    memset(p->ctr, 0, sizeof(uint64_t) * W * 4);
    __atomic_store_n(&p->done, 0, __ATOMIC_RELAXED);
    __atomic_add_fetch(&p->gen, 1, __ATOMIC_SEQ_CST);
    mt_fwake(&p->gen);
    mt_run_batch(p, 0);
    for (uint32_t d; (d = __atomic_load_n(&p->done, __ATOMIC_ACQUIRE)) != (uint32_t)(p->T - 1);) mt_wait_change(p, &p->done, d);
    res->over_limit_count = res->cache_hits = res->cache_misses = res->unexpired_evictions = 0;
    for (uint32_t w = 0; w < W; w++) {
        res->over_limit_count += p->ctr[w * 4]; res->cache_hits += p->ctr[w * 4 + 1];
        res->cache_misses += p->ctr[w * 4 + 2]; res->unexpired_evictions += p->ctr[w * 4 + 3];
    }
    o->over_limit += res->over_limit_count; o->hits += res->cache_hits;
    o->misses += res->cache_misses; o->unexpired_evictions += res->unexpired_evictions;
    res->cache_size = oracle_size(o);
    return 0;
}

static void item_from_abi(const guber_item_t* in, citem_t* it) {
    memset(it, 0, sizeof(*it));
    it->algorithm = in->algorithm; it->expire_at = in->expire_at; it->invalid_at = in->invalid_at;
    if (in->algorithm == GUBER_ALGO_TOKEN_BUCKET) {
        it->vkind = VK_TOKEN; it->t_status = in->status; it->t_limit = in->limit; it->t_duration = in->duration;
        it->t_remaining = in->remaining; it->t_created_at = in->stamp;
    } else if (in->algorithm == GUBER_ALGO_LEAKY_BUCKET) {
        it->vkind = VK_LEAKY; it->l_limit = in->limit; it->l_duration = in->duration;
        it->l_remaining = in->remaining_f; it->l_updated_at = in->stamp; it->l_burst = in->burst;
    } else {
        it->vkind = VK_NIL; /* gubernator.go:435-455: no Value for an unknown algorithm */
    }
}
static void item_to_abi(const citem_t* it, guber_item_t* out) {
    memset(out, 0, sizeof(*out));
    out->key = (const uint8_t*)it->key; out->key_len = it->klen;
    out->expire_at = it->expire_at; out->invalid_at = it->invalid_at;
    if (it->vkind == VK_TOKEN) {
        out->algorithm = GUBER_ALGO_TOKEN_BUCKET; out->status = (uint8_t)it->t_status; out->limit = it->t_limit;
        out->duration = it->t_duration; out->remaining = it->t_remaining; out->stamp = it->t_created_at;
    } else if (it->vkind == VK_LEAKY) {
        out->algorithm = GUBER_ALGO_LEAKY_BUCKET; out->limit = it->l_limit; out->duration = it->l_duration;
        out->remaining_f = it->l_remaining; out->stamp = it->l_updated_at; out->burst = it->l_burst;
    } else {
        out->algorithm = (uint8_t)it->algorithm;
    }
}

/* workers.go:537-581 AddCacheItem -> handleAddCacheItem -> cache.Add */
int oracle_add_item(oracle_t* o, const guber_item_t* in, int64_t now_ms, int* existed) {
    citem_t it; item_from_abi(in, &it);
    uint64_t h = oracle_xxhash64(in->key, in->key_len, 0);
    int ex = lru_add(o, &o->workers[worker_index(o, h)], &it, (const char*)in->key, in->key_len, h, now_ms, NULL);
    if (existed) *existed = ex;
    return 0;
}
/* workers.go:583-626 GetCacheItem -> cache.GetItem */
int oracle_get_item(oracle_t* o, const uint8_t* key, uint32_t klen, int64_t now_ms, guber_item_t* out, int* found) {
    uint64_t h = oracle_xxhash64(key, klen, 0);
    citem_t* e = lru_get_item(o, &o->workers[worker_index(o, h)], (const char*)key, klen, h, now_ms);
    *found = e != NULL;
    if (e && out) item_to_abi(e, out);
    return 0;
}
int oracle_remove_item(oracle_t* o, const uint8_t* key, uint32_t klen) {
    uint64_t h = oracle_xxhash64(key, klen, 0);
    lru_remove(&o->workers[worker_index(o, h)], (const char*)key, klen, h);
    return 0;
}
/* lrucache.go:159-161 Size, summed over workers */
int64_t oracle_size(oracle_t* o) {
    int64_t s = 0;
    for (uint32_t i = 0; i < o->nworkers; i++) s += o->workers[i].len;
    return s;
}
/* lrucache.go:76-85 Each / workers.go:451-534 Store: visit every resident item (order unspecified) */
uint64_t oracle_each(oracle_t* o, guber_item_t* items, uint64_t cap) {
    uint64_t n = 0;
    for (uint32_t w = 0; w < o->nworkers; w++)
        for (citem_t* e = o->workers[w].head; e; e = e->next) {
            if (n < cap && items) item_to_abi(e, &items[n]);
            n++;
        }
    return n;
}
void oracle_counters(oracle_t* o, uint64_t out[4]) {
    out[0] = o->over_limit; out[1] = o->hits; out[2] = o->misses; out[3] = o->unexpired_evictions;
}

/* ------------------------------------------------------------------------------------------
 * Gregorian intervals (interval.go:84-148), UTC only (Go uses now.Location()).
 * ---------------------------------------------------------------------------------------- */
static int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}
static void civil_from_days(int64_t z, int64_t* y, unsigned* m, unsigned* d) {
    z += 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const unsigned doe = (unsigned)(z - era * 146097);
    const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t yy = (int64_t)yoe + era * 400;
    const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const unsigned mp = (5 * doy + 2) / 153;
    *d = doy - (153 * mp + 2) / 5 + 1;
    *m = mp < 10 ? mp + 3 : mp - 9;
    *y = yy + (*m <= 2);
}
static int64_t floordiv(int64_t a, int64_t b) { int64_t q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }
#define NS_PER_S 1000000000LL
#define NS_PER_DAY (86400LL * NS_PER_S)

/* interval.go:117-148 GregorianExpiration */
int oracle_gregorian_expiration(int64_t now_ns, int64_t d, int64_t* out) {
    int64_t day = floordiv(now_ns, NS_PER_DAY);
    int64_t y; unsigned m, dd;
    civil_from_days(day, &y, &m, &dd);
    switch (d) {
    case GUBER_GREGORIAN_MINUTES: {                                 /* :119-122 Truncate(Minute)+Minute-1ns */
        int64_t t = floordiv(now_ns, 60 * NS_PER_S) * 60 * NS_PER_S + 60 * NS_PER_S - 1;
        *out = floordiv(t, 1000000); return 0; }
    case GUBER_GREGORIAN_HOURS: {                                   /* :123-128 */
        int64_t t = floordiv(now_ns, 3600 * NS_PER_S) * 3600 * NS_PER_S + 3600 * NS_PER_S - 1;
        *out = floordiv(t, 1000000); return 0; }
    case GUBER_GREGORIAN_DAYS: {                                    /* :129-132 */
        int64_t t = day * NS_PER_DAY + NS_PER_DAY - 1;
        *out = floordiv(t, 1000000); return 0; }
    case GUBER_GREGORIAN_WEEKS: *out = 0; return -(int)GUBER_ITEM_E_GREGORIAN_WEEKS; /* :133-134 */
    case GUBER_GREGORIAN_MONTHS: {                                  /* :135-139 */
        int64_t ny = y; unsigned nm = m + 1; if (nm > 12) { nm = 1; ny++; }
        int64_t t = days_from_civil(ny, nm, 1) * NS_PER_DAY - 1;
        *out = floordiv(t, 1000000); return 0; }
    case GUBER_GREGORIAN_YEARS: {                                   /* :140-145 */
        int64_t t = days_from_civil(y + 1, 1, 1) * NS_PER_DAY - 1;
        *out = floordiv(t, 1000000); return 0; }
    }
    *out = 0; return -(int)GUBER_ITEM_E_GREGORIAN_INVALID;          /* :147 */
}
/* interval.go:84-110 GregorianDuration.  Months/years reproduce the reference's operator
 * precedence as written: end.UnixNano() - begin.UnixNano()/1000000 (:99, :105). */
int oracle_gregorian_duration(int64_t now_ns, int64_t d, int64_t* out) {
    int64_t day = floordiv(now_ns, NS_PER_DAY);
    int64_t y; unsigned m, dd;
    civil_from_days(day, &y, &m, &dd);
    switch (d) {
    case GUBER_GREGORIAN_MINUTES: *out = 60000; return 0;
    case GUBER_GREGORIAN_HOURS: *out = 3600000; return 0;
    case GUBER_GREGORIAN_DAYS: *out = 86400000; return 0;
    case GUBER_GREGORIAN_WEEKS: *out = 0; return -(int)GUBER_ITEM_E_GREGORIAN_WEEKS;
    case GUBER_GREGORIAN_MONTHS: {
        int64_t begin = days_from_civil(y, m, 1) * NS_PER_DAY;
        int64_t ny = y; unsigned nm = m + 1; if (nm > 12) { nm = 1; ny++; }
        int64_t end = days_from_civil(ny, nm, 1) * NS_PER_DAY - 1;
        *out = end - begin / 1000000; return 0; }
    case GUBER_GREGORIAN_YEARS: {
        int64_t begin = days_from_civil(y, 1, 1) * NS_PER_DAY;
        int64_t end = days_from_civil(y + 1, 1, 1) * NS_PER_DAY - 1;
        *out = end - begin / 1000000; return 0; }
    }
    *out = 0; return -(int)GUBER_ITEM_E_GREGORIAN_INVALID;
}

/* ------------------------------------------------------------------------------------------
 * Replicated consistent hash (replicated_hash.go:78-119)
 * ---------------------------------------------------------------------------------------- */
struct oracle_ring { uint64_t* hash; uint32_t* owner; uint32_t n; int kind; };
typedef struct { uint64_t h; uint32_t o; uint32_t seq; } ringpt_t;
static int ringpt_cmp(const void* a, const void* b) {
    const ringpt_t *x = (const ringpt_t*)a, *y = (const ringpt_t*)b;
    if (x->h != y->h) return x->h < y->h ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}
static uint64_t ring_hash(int kind, const uint8_t* p, size_t n) { return kind == 1 ? oracle_fnv1a_64(p, n) : oracle_fnv1_64(p, n); }

oracle_ring_t* oracle_ring_create(const char* const* peers, uint32_t n_peers, uint32_t replicas, int kind) {
    oracle_ring_t* r = (oracle_ring_t*)calloc(1, sizeof(*r));
    r->n = n_peers * replicas; r->kind = kind;
    ringpt_t* pts = (ringpt_t*)malloc(sizeof(ringpt_t) * (r->n ? r->n : 1));
    uint32_t k = 0;
    for (uint32_t p = 0; p < n_peers; p++) {
        uint8_t dig[16]; char hex[33];
        oracle_md5((const uint8_t*)peers[p], strlen(peers[p]), dig);   /* :81 fmt.Sprintf("%x", md5.Sum(addr)) */
        for (int i = 0; i < 16; i++) snprintf(hex + 2 * i, 3, "%02x", dig[i]);
        for (uint32_t i = 0; i < replicas; i++) {                        /* :82-88 strconv.Itoa(i) + key */
            char buf[64]; int len = snprintf(buf, sizeof buf, "%u%s", i, hex);
            pts[k].h = ring_hash(kind, (const uint8_t*)buf, (size_t)len); pts[k].o = p; pts[k].seq = k; k++;
        }
    }
    qsort(pts, r->n, sizeof(ringpt_t), ringpt_cmp);                      /* :90 sort by hash */
    r->hash = (uint64_t*)malloc(sizeof(uint64_t) * (r->n ? r->n : 1));
    r->owner = (uint32_t*)malloc(sizeof(uint32_t) * (r->n ? r->n : 1));
    for (uint32_t i = 0; i < r->n; i++) { r->hash[i] = pts[i].h; r->owner[i] = pts[i].o; }
    free(pts);
    return r;
}
void oracle_ring_destroy(oracle_ring_t* r) { if (r) { free(r->hash); free(r->owner); free(r); } }
/* replicated_hash.go:104-119 Get: first point with hash >= key hash, wrapping to 0 */
uint32_t oracle_ring_get(const oracle_ring_t* r, const uint8_t* key, uint32_t klen) {
    uint64_t h = ring_hash(r->kind, key, klen);
    uint32_t lo = 0, hi = r->n;
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (r->hash[mid] >= h) hi = mid; else lo = mid + 1; }
    if (lo == r->n) lo = 0;
    return r->owner[lo];
}
