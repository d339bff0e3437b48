"""The device wire decoder's chain walks — the serial k_wire_scan, and the parallel k_wire_win_a / k_wire_win_b (a workgroup per 8 KB
window of a payload: pointer doubling; the first launch says where the chain enters every window, the second finds the records) with
the serial walk — and the numbering of the batch by its launch's last workgroup — behind them (guber_kernels_wire.h) —
compiled for the host (tests/hostsim/wiresim.cpp: fakehip) against the framing code they share with the host
transcoder and its AddressSanitizer fuzz (scan_toplevel over plain memory): generated and mutated payloads (bodies full of bytes that
look like tags, 1- / 2- / 3-byte and non-minimal lengths, unknown fields of every wire type, multi-byte and over-long tags,
truncations, flipped bytes; and payloads of nothing but plain records: thousands of them over several windows, records across and
longer than a window, empty records, ends on a window's edge) — item counts, verdicts and every record's offset and length agree, and
the parallel walk really finishes the plain payloads on its own.  The GPU twin is
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
    L.ws_fuzz.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(C.c_ulonglong)]      # (.., mode, max_per_rpc, stats[5])
    return L


@pytest.mark.parametrize("table", [0, 1], ids=["serial", "parallel_then_serial"])
@pytest.mark.parametrize("max_per_rpc", [0, 1000])
def test_chain_walk_agrees_with_the_shared_framing_code(lib, table, max_per_rpc):
    total = 0
    iters = 100 if table else 300                    # (the parallel walk is 1 024 fibers per payload on the CPU: fewer rounds of it)
    for seed in range(4):
        st = (C.c_ulonglong * 5)()
        bad = lib.ws_fuzz(iters, 1000 + seed, table, max_per_rpc, st)
        assert bad == 0, (seed, bad)
        assert st[0] > 2 * iters and st[2] > iters // 15          # payloads, of which malformed / too large
        assert not table or st[3] > iters // 3, st[3]             # payloads (of 1 KB and more) the parallel walk finished without the serial one
        assert not table or st[4] > iters // 12, st[4]            # ... of more than one window (the chain handed from window to window)
        total += st[1]
    assert total > 1000 * iters                                   # records walked
