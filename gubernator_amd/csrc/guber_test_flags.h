// guber_test_flags.h — bits of guber_config_t.flags that exist for the TEST SUITE (tests/: gubernator_amd.FLAG_TEST_*): each forces a code
// path that production traffic reaches only by size or by accident, so that the parity tests can drive it with small inputs.  Not part
// of the public header (include/guber_gpu.h reserves the bits); a binding never sets them.
#pragma once
#define GUBER_FLAG_TEST_WEAK_HASH 1u    /* keep 6 bits of the key hash so distinct keys collide and the exact-key verification / retry path is exercised */
#define GUBER_FLAG_TEST_FORCE_RADIX 2u  /* evaluate small batches with the large-batch (global radix sort) kernel sequence as well */
#define GUBER_FLAG_TEST_CAREFUL 4u      /* never claim speculatively (the retry-round code path) */
#define GUBER_FLAG_DIR_CLAIMS 16u       /* accepted and ignored (round-1 tuning knob: per-batch claims in the directory entries) */
#define GUBER_FLAG_TEST_NO_SMALL 32u    /* batches of <= 256 requests take the two-launch pipeline too (not the one-launch small path) */
#define GUBER_FLAG_TEST_FORCE_PART 64u  /* every batch (also <= 256 requests, also host-resident ones) takes the owner-partitioned three-launch pipeline */
