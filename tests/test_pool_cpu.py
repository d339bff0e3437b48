"""The C++ pool's host logic (gubernator_amd/csrc/worker_pool.cpp: slot reservation by the callers, one dispatcher per device,
stage rotation, generations, placement passes with bucket migration, the C entry point, shutdown) on the CPU: tests/hostsim/pool_test.cpp links the pool against a test-only stub of the engine's C
ABI that answers with the oracle (tests/hostsim/engine_stub.cpp), and compares what callers get back with the oracle
evaluating every key's requests in the caller's order.  Run plain and under ThreadSanitizer / AddressSanitizer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["tests/hostsim/pool_test.cpp", "tests/hostsim/engine_stub.cpp", "gubernator_amd/csrc/worker_pool.cpp",
       "gubernator_amd/csrc/placement.cpp", "gubernator_amd/csrc/guber_host.cpp"]


def build(tag, flags):
    out = f"/tmp/guber_pool_test_{tag}"
    obj = f"/tmp/guber_pool_oracle_{tag}.o"
    subprocess.run(["gcc", "-O1", "-g", *flags, "-c", "oracle/guber_oracle.c", "-o", obj], cwd=ROOT, check=True)
    # -DGUBER_POOL_TEST_HOOKS: the pool tells pool_test.cpp where a caller is (a placement pass in the middle of a routing round: block 10)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-DGUBER_POOL_TEST_HOOKS", "-DGUBER_LAB", *flags, *SRC, obj, "-o", out, "-lpthread"], cwd=ROOT, check=True)
    return out


ENVS = {"eager": {},                                                          # the default policy: a batch goes as soon as the device has room
        "limit_or_wait": {"GUBER_POOL_EAGER": "0"},                            # the reference's peer-batcher policy alone
        "idle_flush": {"GUBER_POOL_EAGER": "0", "GUBER_POOL_IDLE_US": "20"},
        "few_active": {"GUBER_POOL_MAX_ACTIVE": "2", "GUBER_POOL_DEPTH": "1"},
        "direct": {"GUBER_POOL_DIRECT_CALLERS": "64"},                         # small RPCs evaluated by their callers whatever the load
        "per_shard_stages": {"GUBER_POOL_ROUTED": "0"},                        # every shard its own stages, the callers sort by shard (default: one front stage per device)
        "device_routes": {"GUBER_POOL_DEVROUTE": "1", "GUBER_STUB_ROUTE_LAT_US": "150"}}                         # one front stage, the DEVICE hashes / looks up / ranks (guber_stage_route; default: the callers do)


# Round 4 left two rare mismatches of the non-default arrangements under ThreadSanitizer's timing, retried here.  Both are root-caused and
# fixed (worker_pool.cpp): (1) device routing — a generation of <= 256 requests, routed on the host and submitted at
# once, overtook an earlier generation still waiting for the device to return its shares' sizes (and poll() took those in swap-remove
# order): the halves of an RPC that spanned the two were evaluated in the wrong order; now strictly first in, first out (the stub's
# GUBER_STUB_ROUTE_LAT_US keeps generations waiting long enough for the old code to fail two runs in three).  (2) stages per shard — a
# routing round interrupted by a placement pass put a moving key's earlier requests on the old shard's list and the later ones on the
# new shard's; everything was refused (stale version) and re-queued list by list, later requests first: block 10 of pool_test.cpp makes
# a pass happen inside every routing round and failed three runs in three without the fix.  No answer was ever applied twice — the
# totals were right, one RPC's answers permuted.  There is no retry any more: any mismatch fails.


@pytest.mark.parametrize("tag,flags,scale,env,repeats", [("plain", [], 1, "eager", 2), ("plain", [], 2, "limit_or_wait", 2), ("plain", [], 2, "idle_flush", 2),
                                                         ("plain", [], 2, "few_active", 2), ("plain", [], 2, "direct", 2), ("plain", [], 2, "per_shard_stages", 2), ("plain", [], 2, "device_routes", 2),
                                                         ("tsan", ["-fsanitize=thread"], 4, "direct", 1), ("tsan", ["-fsanitize=thread"], 4, "per_shard_stages", 1), ("tsan", ["-fsanitize=thread"], 3, "device_routes", 1),
                                                         ("tsan", ["-fsanitize=thread"], 3, "eager", 1), ("tsan", ["-fsanitize=thread"], 4, "idle_flush", 1),
                                                         ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], 3, "eager", 1)])
def test_pool_host_logic(tag, flags, scale, env, repeats):
    exe = build(tag, flags)
    for _ in range(repeats):                                         # (races show up in some runs only)
        p = subprocess.run([exe, str(scale)], capture_output=True, text=True, timeout=900, env=dict(os.environ, **ENVS[env]))
        tail = (p.stdout + p.stderr)[-3000:]
        assert p.returncode == 0 and "POOL TEST OK" in p.stdout, tail
        assert "ThreadSanitizer" not in p.stderr and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, tail


@pytest.mark.parametrize("callers,rpcs,items", [(12, 40, 60), (32, 40, 40), (48, 20, 200)])
def test_the_payload_stages_host_protocol_under_thread_sanitizer(callers, rpcs, items):
    """gubernator_amd/csrc/guber_wire_pool.h (guber_wire_pool_*: the callers' reservation word, the intake and front threads, the hand-over
    ring, the wake-ups, shutdown) compiled on its own against stand-ins for the device decoder and the front (the host transcoder and the
    oracle, completing after a few polls): tests/hostsim/wire_pool_tsan.cpp.  Three arrangements per run (four stages / two stages and one
    decode at a time / three RPCs per stage so that callers close full stages), per-key conservation over all answers, a pool destroyed
    without ever seeing a caller.  (The real decoder and front run under AddressSanitizer: tests/test_enginesim_cpu.py.)"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim"), "wire_pool_tsan"], check=True)
    for _ in range(2):
        p = subprocess.run(["/tmp/guber_wire_pool_tsan", str(callers), str(rpcs), str(items)], capture_output=True, text=True, timeout=900)
        tail = (p.stdout + p.stderr)[-3000:]
        assert p.returncode == 0 and "WIRE POOL TSAN OK" in p.stdout, tail
        assert "ThreadSanitizer" not in p.stderr, tail
