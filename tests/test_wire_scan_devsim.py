"""The device wire decoder's chain walk — k_wire_scan and its table variant k_wire_scan_tab (guber_kernels_wire.h: GUBER_WIRE_TABLE) —
compiled for the host (tests/hostsim/wiresim.cpp: fakehip, one wave at a time) against the framing code they share with the host
transcoder and its AddressSanitizer fuzz (scan_toplevel over plain memory): generated and mutated payloads (bodies full of bytes that
look like tags, 1- / 2- / 3-byte and non-minimal lengths, unknown fields of every wire type, multi-byte and over-long tags,
truncations, flipped bytes) — item counts, verdicts and every record's offset and length agree.  The GPU twin is
tests/test_gpu_wire_dev.py (against the host transcoder)."""
import ctypes as C
import os
import subprocess

import pytest

from support import ROOT

HS = os.path.join(ROOT, "tests", "hostsim")


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", HS, "wiresim_lib"], check=True)
    L = C.CDLL(os.path.join(HS, "libwiresim.so"))
    L.ws_fuzz.restype = C.c_uint64
    L.ws_fuzz.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(C.c_ulonglong)]
    return L


@pytest.mark.parametrize("table", [0, 1])
@pytest.mark.parametrize("max_per_rpc", [0, 1000])
def test_chain_walk_agrees_with_the_shared_framing_code(lib, table, max_per_rpc):
    total = 0
    for seed in range(4):
        st = (C.c_ulonglong * 3)()
        bad = lib.ws_fuzz(300, 1000 + seed, table, max_per_rpc, st)
        assert bad == 0, (seed, bad)
        assert st[0] > 700 and st[2] > 25           # payloads, of which malformed / too large
        total += st[1]
    assert total > 300_000                           # records walked
