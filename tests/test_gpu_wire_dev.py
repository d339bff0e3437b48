"""The protobuf wire format decoded ON THE DEVICE (include/guber_wire.h guber_wire_dev_*, kernels guber_kernels_wire.h): payloads
serialized by the protobuf runtime on the reference's schema -> k_wire_win_a / k_wire_win_b / k_wire_scan / k_wire_fill -> the batch the engine
evaluates.  Checked against what the requests say (as tests/test_wire_cpu.py checks the host transcoder), against the host
transcoder itself on fuzzed payloads (per RPC: the same verdict, the same items), and end to end against the oracle."""
import os
import time

import numpy as np
import pytest

import gubernator_amd as ga
import support
import wire_replay
from gubernator_amd import wire as gw
from test_wire_cpu import NOW, check_decoded, expected_key, rand_reqs

pytestmark = pytest.mark.gpu


def test_device_decode_matches_the_requests_and_the_host_transcoder():
    rng = np.random.default_rng(7)
    e = ga.Engine(cache_size=1 << 16, max_batch=16384, max_key_bytes=256)
    dec = gw.DevWireDecoder(e, max_items=16384, max_payload_bytes=4 << 20, max_rpcs=256)
    for rnd in range(6):
        rpcs = [rand_reqs(rng, int(rng.integers(1, 400))) for _ in range(int(rng.integers(1, 30)))]
        payloads = [wire_replay.pb_request(r, peer=bool(k & 1)) for k, r in enumerate(rpcs)]
        owner = rng.integers(0, 2, len(rpcs)).astype(np.uint8)
        status, first, count, n = dec.decode(payloads, NOW, is_owner=owner)
        assert (status == 0).all() and n == sum(len(r) for r in rpcs)
        cols = dec.columns()
        for k, reqs in enumerate(rpcs):
            assert count[k] == len(reqs)
            check_decoded(cols, int(first[k]), reqs, NOW, is_owner=int(owner[k]))
            for j, r in enumerate(reqs):
                want = 1 if not r["unique_key"] else (2 if not r["name"] else 0)          # gubernator.go:208-217
                assert cols["pre_err"][first[k] + j] == want
    dec.close(); e.close()


def _mutate(rng, p):
    p = bytearray(p)
    for _ in range(int(rng.integers(0, 4))):
        if not p:
            break
        i = int(rng.integers(0, len(p)))
        op = int(rng.integers(0, 5))
        if op == 0:
            p[i] ^= 1 << int(rng.integers(0, 8))
        elif op == 1:
            p.insert(i, int(rng.integers(0, 256)))
        elif op == 2:
            del p[i]
        elif op == 3:
            del p[i:]
        else:
            p[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9))).astype(np.uint8))
    return bytes(p)


def test_fuzzed_payloads_same_verdict_and_items_as_the_host_transcoder():
    """mutated payloads (bit flips, insertions, truncations — also right at the window boundaries of the scan): per RPC the device
    decoder rejects exactly what the host transcoder rejects, and accepts with the same items"""
    rng = np.random.default_rng(int(os.environ.get("GUBER_WIRE_FUZZ_SEED", "11")))      # (scripts/gpu_r05_p.sh: a soak over seeds)
    e = ga.Engine(cache_size=1 << 16, max_batch=32768, max_key_bytes=512)
    dec = gw.DevWireDecoder(e, max_items=32768, max_payload_bytes=8 << 20, max_rpcs=512)
    accepted = rejected = 0
    for rnd in range(8):
        base = [wire_replay.pb_request(rand_reqs(rng, int(rng.integers(1, 300)))) for _ in range(40)]
        payloads = [_mutate(rng, base[int(rng.integers(0, len(base)))]) for _ in range(200)]
        status, first, count, n = dec.decode(payloads, NOW, max_per_rpc=0)
        cols = dec.columns()
        for k, p in enumerate(payloads):
            wb = gw.WireBatch(max_items=4096, max_key_bytes=1 << 20)
            wb.reset(NOW)
            try:
                f0, c0 = wb.decode(p, max_per_rpc=0)
                host = wb.arrays()
                assert status[k] == 0, (rnd, k)
                assert count[k] == c0
                accepted += 1
                for j in range(c0):
                    i = int(first[k]) + j
                    assert cols["keys"][i] == host["keys"][j] or (len(host["keys"][j]) > 512 and cols["key_len"][i] == len(host["keys"][j]))
                    for name in ("hits", "limit", "duration", "burst", "created_at", "algorithm", "behavior"):
                        assert cols[name][i] == host[name][j], (name, rnd, k, j)
            except ga.GuberError as ex:
                assert ex.code == gw.E_WIRE_MALFORMED, ex
                assert status[k] == gw.E_WIRE_MALFORMED, (rnd, k, status[k])
                rejected += 1
            wb.close()
    assert accepted > 200 and rejected > 200, (accepted, rejected)
    dec.close(); e.close()


def _same_as_the_host_transcoder(dec, payloads, max_key):
    status, first, count, n = dec.decode(payloads, NOW, max_per_rpc=0)
    cols = dec.columns()
    for k, p in enumerate(payloads):
        wb = gw.WireBatch(max_items=4096, max_key_bytes=1 << 21)
        wb.reset(NOW)
        f0, c0 = wb.decode(p, max_per_rpc=0)
        host = wb.arrays()
        assert status[k] == 0 and count[k] == c0, (k, status[k], count[k], c0)
        for j in range(c0):
            i = int(first[k]) + j
            assert cols["keys"][i] == host["keys"][j] or (len(host["keys"][j]) > max_key and cols["key_len"][i] == len(host["keys"][j])), (k, j)
            for name in ("hits", "limit", "duration", "burst", "created_at", "algorithm", "behavior"):
                assert cols[name][i] == host[name][j], (name, k, j)
        wb.close()
    return int(n)


def test_payloads_of_many_windows_and_records_across_their_edges():
    """the parallel chain walk hands a payload's chain from 8 KB window to window (k_wire_win_a says where it enters each): payloads of
    up to sixteen windows whose records (1 .. 300 bytes) cross every edge at another place, a payload that ends exactly on an edge
    and one byte to either side of it, records of about a kilobyte (the longest a chain may enter a window with: beyond, the payload is
    the serial walk's), one record longer than a window — all of them the host transcoder's items, RPC by RPC"""
    rng = np.random.default_rng(int(os.environ.get("GUBER_WIRE_FUZZ_SEED", "23")))
    e = ga.Engine(cache_size=1 << 16, max_batch=32768, max_key_bytes=512)
    dec = gw.DevWireDecoder(e, max_items=32768, max_payload_bytes=8 << 20, max_rpcs=64)

    def req(i, klen):
        return dict(name="n", unique_key=("k%d_" % i) + "x" * klen, hits=int(rng.integers(0, 3)), limit=10, duration=60_000, algorithm=int(rng.integers(0, 2)),
                    behavior=0, burst=0, created_at=0)
    total = 0
    for rnd in range(3):
        payloads = []
        for _ in range(6):                                         # many windows, records of every length across the edges
            payloads.append(wire_replay.pb_request([req(i, int(rng.integers(0, 280))) for i in range(int(rng.integers(300, 1000)))]))
        for target in (8192, 8191, 8193, 16384, 24576 + 1):        # the payload's end on / beside a window's edge
            reqs, size = [], 0
            while True:
                r = req(len(reqs), int(rng.integers(0, 40)))
                sz = len(wire_replay.pb_request([r]))
                if size + sz > target - 60:
                    break
                reqs.append(r); size += sz
            pad = target - size                                     # the last record brings the payload to the byte
            for klen in range(0, 200):
                r = req(len(reqs), klen)
                if len(wire_replay.pb_request([r])) == pad:
                    reqs.append(r); size += pad
                    break
            payloads.append(wire_replay.pb_request(reqs))
            assert len(payloads[-1]) == size
        payloads.append(wire_replay.pb_request([req(i, int(rng.integers(900, 1015))) for i in range(60)]))     # about a kilobyte each
        payloads.append(wire_replay.pb_request([req(i, int(rng.integers(1015, 1300))) for i in range(40)]))    # more: somewhere the serial walk's
        payloads.append(wire_replay.pb_request([req(0, 5), req(1, 9000), req(2, 7)] + [req(3 + i, 20) for i in range(400)]))   # one record longer than a window
        total += _same_as_the_host_transcoder(dec, payloads, 512)
    assert total > 10_000
    dec.close(); e.close()


def test_device_decoded_batches_evaluate_like_the_oracle_and_report_throughput():
    rng = np.random.default_rng(3)
    e, o = ga.Engine(cache_size=1 << 18, max_batch=65536, max_key_bytes=64), support.Oracle(cache_size=1 << 20)
    dec = gw.DevWireDecoder(e, max_items=65536, max_payload_bytes=8 << 20, max_rpcs=1024)
    now = NOW
    for rnd in range(4):
        rpcs = [[dict(name="ns_%d" % rng.integers(0, 5), unique_key="acct:%d" % rng.integers(0, 3000), hits=1, limit=50, duration=60_000, algorithm=int(rng.integers(0, 2)),
                      behavior=0, burst=0, created_at=0) for _ in range(int(rng.integers(1, 1000)))] for _ in range(20)]
        payloads = [wire_replay.pb_request(r) for r in rpcs]
        status, first, count, n = dec.decode(payloads, now)
        got = dec.eval()
        flat = [r for reqs in rpcs for r in reqs]
        want = o.eval(support.HostBatch([expected_key(r) for r in flat], 1, 50, 60_000, now, algorithm=np.array([r["algorithm"] for r in flat], np.uint8)))
        support.assert_results_equal(got, want, f"round {rnd}")
        now += 700
    # throughput, decode only (host copy into the pinned buffer + 5 launches + the read-back of the per-RPC verdicts): a workgroup per
    # 8 KB window of a payload finds its part of the record chain by pointer doubling (k_wire_win_a, k_wire_win_b), the serial walk takes what they leave
    import ctypes
    import os
    on_cpu_engine = "enginesim" in os.environ.get("GUBER_HIP_LIB", "")     # (the kernel source on the CPU, tests/test_enginesim_cpu.py: the calls, not the rates)
    for per_rpc, nrpc in (((1000, 2), (100, 4), (10, 8)) if on_cpu_engine else ((1000, 64), (100, 640), (10, 4000))):
        reqs = [dict(name="bench", unique_key="k%08d" % i, hits=1, limit=100, duration=60_000, algorithm=0, behavior=0, burst=0, created_at=0) for i in range(per_rpc)]
        payloads = [wire_replay.pb_request(reqs)] * nrpc
        dec2 = gw.DevWireDecoder(e, max_items=65536, max_payload_bytes=8 << 20, max_rpcs=4096)
        dec2.decode(payloads, now)
        t0 = time.perf_counter()
        reps = 1 if on_cpu_engine else 20
        for _ in range(reps):
            _, _, _, n = dec2.decode(payloads, now)
        dt = (time.perf_counter() - t0) / reps
        print(f"device wire decode: {nrpc} payloads x {per_rpc} items = {n} items, {sum(map(len, payloads))} bytes in {dt * 1e6:.0f} us = {n / dt / 1e6:.0f} M items/s")
        # the same payloads already in the decoder's pinned buffer (a receive path that reads its sockets into guber_wire_dev_buffer):
        # no host copy — and the same batch, column by column
        ref = dec2.columns()
        buf = dec2.buffer()
        offs, pos = [], 0
        for p in payloads:
            pos = (pos + 15) & ~15
            buf[pos:pos + len(p)] = np.frombuffer(p, np.uint8)
            offs.append(pos); pos += len(p)
        lens = [len(p) for p in payloads]
        st2, first2, count2, n2 = dec2.decode_staged(offs, lens, now)
        assert n2 == n and (st2 == 0).all()
        got = dec2.columns()
        assert got["keys"] == ref["keys"] and all(np.array_equal(got[k], ref[k]) for k in ref if k != "keys")
        t0 = time.perf_counter()
        for _ in range(reps):
            dec2.decode_staged(offs, lens, now)
        dt = (time.perf_counter() - t0) / reps
        print(f"device wire decode, payloads in the pinned buffer: {nrpc} payloads x {per_rpc} items in {dt * 1e6:.0f} us = {n / dt / 1e6:.0f} M items/s")
        with pytest.raises(ga.GuberError):
            dec2.decode_staged([8], [16], now)                   # (an offset that is not 16-byte aligned)
        dec2.close()
    dec.close(); e.close(); o.close()


@pytest.mark.parametrize("fixed_width", [False, True], ids=["ragged_keys", "keys_of_one_width"])
def test_device_decoded_batches_through_a_front_over_four_tables(fixed_width):
    """guber_wire_dev_eval_front: serialized RPC payloads -> decoded on the device (items in the order of their RPCs: arrival order, keys as
    rows) -> routed to four tables by the front (XXH64 + the placement's rule) -> evaluated -> answered in the items' order: equal to ONE
    oracle fed the flat item list.  Keys of one width travel with their requests into the shares (packed), ragged ones stay in the decoder's
    rows; items the decoder pre-rejected (an empty unique_key: an empty key row) keep their place and are answered with the item error."""
    rng = np.random.default_rng(5)
    place = ga.Placement(4)
    e0 = ga.Engine(cache_size=1 << 16, max_batch=16384, max_key_bytes=64)
    engs = [e0] + [ga.Engine(cache_size=1 << 16, max_batch=16384, max_key_bytes=64, stream=e0.stream_handle()) for _ in range(3)]
    fr = ga.Front(engs, place, max_n=16384, depth=4)
    dec = gw.DevWireDecoder(e0, max_items=16384, max_payload_bytes=2 << 20, max_rpcs=256)
    o = support.Oracle(cache_size=1 << 20)
    now = NOW
    for rnd in range(4):
        def ukey():
            k = int(rng.integers(0, 2000))
            return ("acct:%06d" % k) if fixed_width else ("acct:%d" % k) + "x" * int(k % 5)
        rpcs = [[dict(name="ns_1" if fixed_width else "ns_%d" % rng.integers(0, 12), unique_key=ukey(), hits=1, limit=20, duration=60_000, algorithm=int(rng.integers(0, 2)),
                      behavior=0, burst=0, created_at=0) for _ in range(int(rng.integers(1, 600)))] for _ in range(12)]
        if not fixed_width:
            rpcs[3][0]["unique_key"] = ""                          # gubernator.go:208-211: answered in place with the reference's error
        payloads = [wire_replay.pb_request(r) for r in rpcs]
        status, first, count, n = dec.decode(payloads, now)
        assert (status == 0).all()
        got = dec.eval_front(fr)
        flat = [r for reqs in rpcs for r in reqs]
        ok = np.array([bool(r["unique_key"]) for r in flat])
        good = [r for r in flat if r["unique_key"]]
        want = o.eval(support.HostBatch([expected_key(r) for r in good], 1, 20, 60_000, now, algorithm=np.array([r["algorithm"] for r in good], np.uint8)))
        assert (got.err[:n][~ok] == 4).all()                       # GUBER_ITEM_E_EMPTY_KEY: never reached a bucket
        sel = ga.HostResult(len(good))
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(sel, name)[:] = getattr(got, name)[:n][ok]
        support.assert_results_equal(sel, want, f"round {rnd}")
        now += 700
    assert sum(e.size() for e in engs) == o.size() and min(e.size() for e in engs) > 0
    dec.close(); fr.close()
    for e in engs:
        e.close()
    place.close(); o.close()
