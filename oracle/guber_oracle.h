/*
 * guber_oracle.h — CPU oracle for the gubernator hot path.  TEST INFRASTRUCTURE ONLY: see the
 * header comment of guber_oracle.c.  Shares the SoA batch / result / item struct layouts of
 * include/guber_gpu.h so the same buffers can be handed to the oracle and to the HIP engine.
 */
#ifndef GUBER_ORACLE_H
#define GUBER_ORACLE_H

#include "../include/guber_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle oracle_t;
typedef struct oracle_ring oracle_ring_t;

/* Config.Store (store.go:49-65) as C callbacks; `req` = index in the batch of the request being applied */
typedef struct oracle_store {
    int (*get)(void* user, uint32_t req, guber_item_t* out);                         /* 1 = found: *out filled (key ignored) */
    void (*on_change)(void* user, uint32_t req, const guber_item_t* item);
    void (*remove)(void* user, uint32_t req, const uint8_t* key, uint32_t key_len);
    void* user;
} oracle_store_t;

oracle_t* oracle_create(uint64_t cache_size, uint32_t workers);
void oracle_destroy(oracle_t* o);
int oracle_eval_batch(oracle_t* o, const guber_batch_t* b, guber_result_t* res);
int oracle_eval_batch_store(oracle_t* o, const guber_batch_t* b, guber_result_t* res, const oracle_store_t* st);
int oracle_eval_batch_mt(oracle_t* o, const guber_batch_t* b, guber_result_t* res, int threads);
int oracle_add_item(oracle_t* o, const guber_item_t* in, int64_t now_ms, int* existed);
int oracle_get_item(oracle_t* o, const uint8_t* key, uint32_t klen, int64_t now_ms, guber_item_t* out, int* found);
int oracle_remove_item(oracle_t* o, const uint8_t* key, uint32_t klen);
int64_t oracle_size(oracle_t* o);
uint64_t oracle_each(oracle_t* o, guber_item_t* items, uint64_t cap);
void oracle_counters(oracle_t* o, uint64_t out[4]);
uint32_t oracle_worker_index_for_hash63(uint32_t workers, uint64_t hash63);

uint64_t oracle_xxhash64(const uint8_t* p, size_t len, uint64_t seed);
uint64_t oracle_fnv1_64(const uint8_t* p, size_t len);
uint64_t oracle_fnv1a_64(const uint8_t* p, size_t len);
void oracle_md5(const uint8_t* msg, size_t len, uint8_t out[16]);

int oracle_gregorian_expiration(int64_t now_unix_nano, int64_t d, int64_t* expire_ms);
int oracle_gregorian_duration(int64_t now_unix_nano, int64_t d, int64_t* duration);

oracle_ring_t* oracle_ring_create(const char* const* peers, uint32_t n_peers, uint32_t replicas, int kind);
void oracle_ring_destroy(oracle_ring_t* r);
uint32_t oracle_ring_get(const oracle_ring_t* r, const uint8_t* key, uint32_t klen);

#ifdef __cplusplus
}
#endif
#endif
