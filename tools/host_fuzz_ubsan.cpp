#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
// host_fuzz_ubsan.cpp — the host-only C++ of the library (ring, Gregorian intervals, hashes, placement, error strings) under ASan + UBSan.
// g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -I include tools/host_fuzz_ubsan.cpp gubernator_amd/csrc/guber_host.cpp gubernator_amd/csrc/placement.cpp -o /tmp/host_fuzz && /tmp/host_fuzz
#include "../include/guber_gpu.h"
static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main() {
    // gregorian: every selector, extreme clocks
    int64_t out;
    const int64_t clocks[] = {0, 1, -1, 1573430400000000000ll, 4102444800000000000ll, INT64_MAX, INT64_MIN, 9000000000000000000ll};
    for (int64_t c : clocks) for (int64_t d = -3; d < 9; ++d) { guber_gregorian_expiration(c, d, &out); guber_gregorian_duration(c, d, &out); }
    for (int i = 0; i < 200000; ++i) { guber_gregorian_expiration((int64_t)rnd(), (int64_t)(rnd() % 8), &out); guber_gregorian_duration((int64_t)rnd(), (int64_t)(rnd() % 8), &out); }
    // hashes on odd lengths
    std::vector<uint8_t> buf(300);
    for (auto& b : buf) b = (uint8_t)rnd();
    uint64_t acc = 0;
    for (size_t n = 0; n <= 300; ++n) { acc ^= guber_xxhash64(buf.data(), n, n) ^ guber_fnv1_64(buf.data(), n) ^ guber_fnv1a_64(buf.data(), n); }
    // rings: 1..9 peers, both hashes, route ragged keys
    for (int kind = 0; kind < 2; ++kind) for (uint32_t peers = 1; peers < 10; ++peers) {
        std::vector<std::string> names; std::vector<const char*> ptr;
        for (uint32_t i = 0; i < peers; ++i) names.push_back("peer-" + std::to_string(i) + ".svc.local:81");
        for (auto& n : names) ptr.push_back(n.c_str());
        guber_ring_t* r = nullptr;
        if (guber_ring_create(ptr.data(), peers, 512, kind, &r)) return 1;
        std::vector<uint8_t> kb; std::vector<uint32_t> off{0};
        for (int i = 0; i < 1000; ++i) { const size_t L = rnd() % 40; for (size_t j = 0; j < L; ++j) kb.push_back((uint8_t)rnd()); off.push_back((uint32_t)kb.size()); }
        kb.resize(kb.size() + 8);
        std::vector<uint32_t> owner(1000);
        if (guber_ring_route(r, kb.data(), off.data(), 1000, owner.data())) return 1;
        for (uint32_t o : owner) if (o >= peers) return 1;
        guber_ring_destroy(r);
    }
    // placements: odd shard counts, extreme hashes, skewed observations, plan / commit / rebalance in any order
    for (uint32_t shards : {1u, 2u, 3u, 7u, 12u, 64u, 1000u, 4096u}) {
        guber_placement_t* p = nullptr;
        if (guber_placement_create(shards, (uint32_t)(rnd() % 3 == 0 ? 0 : 1 + rnd() % 70000), &p)) return 1;
        const uint64_t edge[] = {0, 1, ~0ull, 1ull << 63, (1ull << 63) - 1, (1ull << 63) + 1};
        for (uint64_t h : edge) if (guber_placement_shard(p, h) >= shards) return 1;
        for (int round = 0; round < 6; ++round) {
            const uint64_t hot = rnd();
            for (int i = 0; i < 20000; ++i) guber_placement_observe(p, (rnd() % 3 == 0) ? hot : rnd(), 1 + (uint32_t)(rnd() % 64));
            guber_placement_move_t mv[8]; uint32_t nm = 0;
            if (round % 3 == 0) (void)guber_placement_rebalance(p, 0.125, round == 0, mv, 8, &nm);
            else { (void)guber_placement_plan(p, round % 2 ? 0.0 : 0.5, mv, 8, &nm); if (round % 2) (void)guber_placement_commit(p); }
            for (int i = 0; i < 2000; ++i) if (guber_placement_shard(p, rnd()) >= shards) return 1;
            if (guber_placement_shard(p, hot) >= shards) return 1;
        }
        uint32_t a = 0, b = 0, c = 0;
        (void)guber_placement_info(p, &a, &b, &c);
        acc ^= a ^ b ^ c ^ guber_placement_version(p);
        guber_placement_destroy(p);
    }
    for (int c = -30; c < 5; ++c) (void)strlen(guber_strerror(c));
    for (int c = 0; c < 12; ++c) (void)strlen(guber_item_strerror((uint8_t)c));
    printf("host fuzz ok %llx\n", (unsigned long long)acc);
    return 0;
}
