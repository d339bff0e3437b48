"""The C++ pool's host logic (gubernator_amd/csrc/worker_pool.cpp: slot reservation by the callers, stage rotation,
generations, shutdown) on the CPU: tests/hostsim/pool_test.cpp links the pool against a test-only stub of the engine's C
ABI that answers with the oracle (tests/hostsim/engine_stub.cpp), and compares what callers get back with the oracle
evaluating every key's requests in the caller's order.  Run plain and under ThreadSanitizer / AddressSanitizer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["tests/hostsim/pool_test.cpp", "tests/hostsim/engine_stub.cpp", "gubernator_amd/csrc/worker_pool.cpp",
       "gubernator_amd/csrc/guber_host.cpp"]


def build(tag, flags):
    out = f"/tmp/guber_pool_test_{tag}"
    obj = f"/tmp/guber_pool_oracle_{tag}.o"
    subprocess.run(["gcc", "-O1", "-g", *flags, "-c", "oracle/guber_oracle.c", "-o", obj], cwd=ROOT, check=True)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", *flags, *SRC, obj, "-o", out, "-lpthread"], cwd=ROOT, check=True)
    return out


@pytest.mark.parametrize("tag,flags,scale,idle_us", [("plain", [], 1, 0), ("plain", [], 2, 20), ("tsan", ["-fsanitize=thread"], 2, 0),
                                                     ("tsan", ["-fsanitize=thread"], 4, 20),
                                                     ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], 2, 0)])
def test_pool_host_logic(tag, flags, scale, idle_us):
    exe = build(tag, flags)
    env = dict(os.environ, GUBER_POOL_IDLE_US=str(idle_us))          # 20: the optional idle flush of the batcher
    p = subprocess.run([exe, str(scale)], capture_output=True, text=True, timeout=600, env=env)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0 and "POOL TEST OK" in p.stdout, tail
    assert "ThreadSanitizer" not in p.stderr and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, tail
