// guber_host.h — host-only pieces of the engine: the consistent-hash ring, calendar helpers, error
// strings.  Internal accessors used by guber_engine.hip live here next to the public C ABI.
#pragma once
#include <stdint.h>

#include "../../include/guber_gpu.h"

extern "C" int guber_ring_kind(const guber_ring_t* r);   // 0 fnv1, 1 fnv1a
extern "C" uint64_t guber_ring_id(const guber_ring_t* r);   // unique per guber_ring_create
namespace guber { struct TzTable; }
int guber_host_set_tz(const guber_tz_t* tz);                 // validates and stores the process's zone (host helpers); guber_set_timezone adds the devices
const guber::TzTable* guber_host_tz_table();
int guber_host_build_tz(const guber_tz_t* tz, guber::TzTable* out);   // validate + build, nothing published
void guber_host_publish_tz(const guber::TzTable& t);                   // the host helpers' copy (guber_set_timezone: after every device has it)

// The product library reads a handful of documented environment variables (INTEGRATION.md "Runtime knobs"); everything else a
// measurement ever switched — pipelines, owner counts, fusion, stage copies, pool policies — is read only by builds with -DGUBER_LAB
// (make -C gubernator_amd/csrc lab; the tests' CPU builds of the engine and the pool).  In the product the names are not even in the
// binary (tests/test_abi_cpu.py looks).
#ifdef GUBER_LAB
#include <cstdlib>
#define guber_lab_env(name) getenv(name)
#else
#define guber_lab_env(name) ((const char*)nullptr)
#endif
