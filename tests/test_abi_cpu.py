"""CPU-side checks of the PRODUCT library (no GPU needed, no compute kernels launched): it loads,
exports every symbol include/guber_gpu.h declares, fails loudly without a device, and its host-only
functions (consistent-hash ring, calendar helpers, hashes, error strings) match the reference's
known answers."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import xxhash

import gubernator_amd as ga
import scenarios
import support


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(ga.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return ga.lib()


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(support.ROOT, "include", "guber_gpu.h")).read()
    declared = set(re.findall(r"\b(guber_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(ga.ABI_SYMBOLS), declared ^ set(ga.ABI_SYMBOLS)
    for name in ga.ABI_SYMBOLS:
        assert hasattr(L, name), name


def test_struct_layouts_match_header():
    assert C.sizeof(ga.GuberConfig) == 48
    assert C.sizeof(ga.GuberBatch) == 8 + 12 * 8 + 8
    assert C.sizeof(ga.GuberResult) == 5 * 8 + 5 * 8
    assert C.sizeof(ga.GuberItem) == 80
    assert C.sizeof(ga.GuberStats) == 120


def test_no_silent_cpu_fallback(L):
    """Without a GPU the product path must fail loudly, never evaluate on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ga.GuberError) as ei:
        ga.Engine(cache_size=1000)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_product_path_does_not_reference_oracle():
    for root, _, files in os.walk(os.path.join(support.ROOT, "gubernator_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.lower().replace("no cpu", ""), os.path.join(root, f)


def test_ring_distribution_kat(L):
    k = scenarios.load("kat_vectors.json")["ring_distribution"]
    keys = [f"192.168.{(i >> 8) & 255}.{i & 255}" for i in range(k["n_keys"])]
    for kind in ("fnv1", "fnv1a"):
        ring = ga.Ring(k["hosts"], k["replicas"], kind)
        owner = ring.route(keys)
        dist = {h: int((owner == i).sum()) for i, h in enumerate(k["hosts"])}
        assert dist == k[kind], (kind, dist)
        hh, _ = ring.points()
        assert len(hh) == 3 * 512 and (np.diff(hh.astype(np.float64)) >= 0).all()


def test_gregorian_kats(L):
    k = scenarios.load("kat_vectors.json")
    out = C.c_int64(0)
    for v in k["gregorian_expiration"]:
        assert L.guber_gregorian_expiration(v["now_ns"], v["d"], C.byref(out)) == 0
        assert out.value == v["expire"], v
    inv = k["gregorian_invalid"]
    assert L.guber_gregorian_expiration(inv["now_ns"], inv["d"], C.byref(out)) == -3 and out.value == 0
    assert L.guber_item_strerror(3).decode() == inv["error"]
    assert L.guber_gregorian_duration(inv["now_ns"], 3, C.byref(out)) == -2
    # product helpers agree with the oracle's on a sweep of instants and selectors
    ol = support.oracle_lib()
    o2 = C.c_int64(0)
    rng = np.random.default_rng(8)
    for _ in range(2000):
        ns = int(rng.integers(0, 4_000_000_000)) * 1_000_000_000 + int(rng.integers(0, 10 ** 9))
        d = int(rng.integers(0, 7))
        assert L.guber_gregorian_expiration(ns, d, C.byref(out)) == ol.oracle_gregorian_expiration(ns, d, C.byref(o2))
        assert out.value == o2.value
        assert L.guber_gregorian_duration(ns, d, C.byref(out)) == ol.oracle_gregorian_duration(ns, d, C.byref(o2))
        assert out.value == o2.value


def test_hashes(L):
    rng = np.random.default_rng(4)
    for n in list(range(0, 80)) + [200, 1000]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert L.guber_xxhash64(b, n, 0) == xxhash.xxh64(b, seed=0).intdigest()
    assert L.guber_fnv1_64(b"foobar", 6) == 0x340d8765a4dda9c2
    assert L.guber_fnv1a_64(b"foobar", 6) == 0x85944171f73967e8
    assert b"Invalid rate limit algorithm" in L.guber_item_strerror(1)
    assert L.guber_strerror(-2).startswith(b"no HIP device")


def test_argument_checks_need_no_device(L):
    """entry points reject malformed calls before touching HIP (so the checks hold on a machine without a GPU)"""
    assert L.guber_eval_batches_routed_dev(None, 0, None, None, None, 0, None) != 0
    assert L.guber_stage_submit(None) != 0 and L.guber_stage_wait(None) != 0
    st = C.c_void_p()
    assert L.guber_stage_create(None, 16, 0, C.byref(st)) != 0 and not st.value
    assert L.guber_eval_batches_dev(None, None, None, 0, None) != 0


def test_headers_are_plain_c99_and_the_go_call_sequence_links(L, tmp_path):
    """cgo compiles the binding's preamble as C.  tests/hostsim/abi_c99.c includes both public headers and makes the calls
    go/gpu_worker_pool.go makes, with the binding's casts: gcc -std=c99 -pedantic -Werror must take it, it must link against the
    product library, and without a GPU its pool creation must fail with GUBER_E_NO_DEVICE (exit 0) — never evaluate on the CPU."""
    import subprocess
    import torch
    exe = str(tmp_path / "abi_c99")
    libdir = os.path.join(support.ROOT, "gubernator_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(support.ROOT, "include"), "-o", exe,
                    os.path.join(support.ROOT, "tests", "hostsim", "abi_c99.c"), "-L", libdir, "-lguber_hip", f"-Wl,-rpath,{libdir}"], check=True)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device leg is for the CPU box (the --gpu leg runs in tests/test_gpu_host_layer.py)")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "no HIP device" in r.stdout or "no CPU fallback" in r.stdout, r.stdout


def test_the_product_library_reads_only_the_documented_environment_variables():
    """the experiments' knobs (pipelines, owner counts, fusion, pool policies ...) are read by -DGUBER_LAB builds only
    (gubernator_amd/csrc/guber_host.h guber_lab_env; `make -C gubernator_amd/csrc lab`, the tests' CPU builds).  In the product library
    the names are not even in the binary: every GUBER_* name it contains is one of the runtime knobs INTEGRATION.md documents."""
    import subprocess
    documented = {"GUBER_RCCL_LIB", "GUBER_POOL_MAX_ACTIVE", "GUBER_POOL_REBALANCE_MS", "GUBER_POOL_SPIN_US", "GUBER_POOL_DEPTH", "GUBER_POOL_DIRECT_MAX", "GUBER_POOL_EAGER"}
    out = subprocess.run(["strings", "-n", "8", ga.LIB_PATH if "enginesim" not in ga.LIB_PATH else os.path.join(support.ROOT, "gubernator_amd", "libguber_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    names = set(re.findall(r"^(GUBER_[A-Z0-9_]+)$", out, re.M))
    assert names <= documented, names - documented
    assert len(documented) <= 8
    text = open(os.path.join(support.ROOT, "INTEGRATION.md")).read()
    for n in documented:
        assert n in text, f"{n} is read by the product but INTEGRATION.md does not say what it does"
    hdr = open(os.path.join(support.ROOT, "include", "guber_gpu.h")).read()
    assert "FLAG_TEST" not in hdr                                     # (the test suite's flag bits live in gubernator_amd/csrc/guber_test_flags.h)
