// enginesim.cpp — TEST-ONLY: the engine itself — gubernator_amd/csrc/guber_engine.hip, HOST code and kernels, with the pool, the
// placement and the wire transcoder — compiled for the host against tests/hostsim/fakehip (the fiber emulation of the kernel language
// + a stand-in for the HIP runtime API in which every launch runs at once on the calling thread).  The library exports the C ABI of
// include/guber_gpu.h; tests/test_enginesim_cpu.py loads it THROUGH GUBER_HIP_LIB IN A PROCESS OF ITS OWN and drives the entry points the
// GPU suite drives, against the oracle, on a machine without a GPU: what it checks is the host logic of the engine (which launches,
// in which order, with which arguments: launch groups, the held-back k_eval3 of GUBER_FUSE_EP, preludes, eviction pre-passes) together
// with the kernels' logic.  It is not a CPU fallback: nothing in the product builds, links or loads it, the product library
// (gubernator_amd/libguber_hip.so) is built by hipcc for gfx950 only and fails loudly without a device.
#define FAKEHIP_RUNTIME
#include <hip/hip_runtime.h>
#include "fakehip/fiber_runtime.h"
#include "../../gubernator_amd/csrc/guber_engine.hip"
