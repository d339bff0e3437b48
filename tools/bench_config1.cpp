// BASELINE configs[0] through the C ABI without an interpreter in the way: one request per guber_eval_batch call
// (benchmark_test.go:63-84 shape: TOKEN_BUCKET, limit 10, duration 5 s, hits 1), 1000 keys cycled and a fresh key per call.
// build: make -C gubernator_amd/csrc bench_config1     run (GPU box): tools/bench_config1_c
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/guber_gpu.h"

int main() {
    guber_config_t cfg{};
    cfg.struct_size = sizeof cfg; cfg.cache_size = 1 << 20; cfg.max_batch = 1024;
    guber_engine_t* e = nullptr;
    if (guber_engine_create(&cfg, &e) != GUBER_OK) { fprintf(stderr, "create: %s\n", guber_last_error()); return 1; }
    const int64_t now = 1700000000000LL;
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> lat;
        for (int i = 0; i < 5200; ++i) {
            char key[32];
            const int len = mode == 0 ? snprintf(key, sizeof key, "bench_%04d", i % 1000) : snprintf(key, sizeof key, "fresh_%09d", i);
            uint8_t kb[48] = {0}; memcpy(kb, key, len);
            uint32_t off[2] = {0, (uint32_t)len};
            int64_t hits = 1, limit = 10, dur = 5000;
            uint8_t algo = 0; uint32_t beh = 0;
            guber_batch_t b{}; b.n = 1; b.key_bytes = kb; b.key_off = off; b.hits = &hits; b.limit = &limit; b.duration = &dur;
            b.algorithm = &algo; b.behavior = &beh; b.now_ms = now + i;
            uint8_t st, er; int64_t ol, orem, ors;
            guber_result_t r{}; r.status = &st; r.limit = &ol; r.remaining = &orem; r.reset_time = &ors; r.err = &er;
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = guber_eval_batch(e, &b, &r);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rc != GUBER_OK || er != 0) { fprintf(stderr, "eval rc %d err %d: %s\n", rc, er, guber_last_error()); return 1; }
            if (i >= 200) lat.push_back(us);
        }
        std::sort(lat.begin(), lat.end());
        double sum = 0; for (double v : lat) sum += v;
        printf("C caller, batch = 1, %-18s: p50 %6.1f us  p99 %6.1f us  min %6.1f us -> %8.0f decisions/s per caller thread\n",
               mode == 0 ? "1000 keys cycled" : "fresh key per op", lat[lat.size() / 2], lat[lat.size() * 99 / 100], lat[0], 1e6 / (sum / lat.size()));
    }
    guber_engine_destroy(e);
    return 0;
}
