"""CPU tests of the wire front end (include/guber_wire.h, gubernator_amd/csrc/wire.cpp): the transcoder is
checked against the python protobuf runtime on the reference's message schema (tests/pb_schema.py), and the
golden functional scenarios are replayed through it with the oracle as evaluator."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import support
from pb_schema import PB
import wire_replay
import gubernator_amd as ga
from gubernator_amd import wire as gw

NOW = 1_700_000_000_000


def rand_reqs(rng, n, bad=True):
    out = []
    for i in range(n):
        kind = rng.integers(0, 12)
        name = "ns_%d" % rng.integers(0, 5) if not (bad and kind == 0) else ""
        uk = ("acct:%d" % rng.integers(0, 1000)) if not (bad and kind == 1) else ""
        if kind == 2:
            name, uk = "ünï_" + "x" * int(rng.integers(0, 70)), "ключ/🔑" + str(i)
        r = dict(name=name, unique_key=uk,
                 hits=int(rng.choice([0, 1, 1, 1, -1, 5, 2**62, -2**63, 2**63 - 1])),
                 limit=int(rng.choice([0, 1, 10, 100, -3, 2**40])),
                 duration=int(rng.choice([0, 1, 5, 1000, 60000, -1, 2**50])),
                 algorithm=int(rng.choice([0, 0, 1, 1, 7, -2])) if bad else int(rng.integers(0, 2)),
                 behavior=int(rng.choice([0, 0, 2, 8, 32, 34, 1 | 16])),
                 burst=int(rng.choice([0, 0, 20, -1])),
                 created_at=int(rng.choice([0, 0, NOW - 5, 12345])))
        out.append(r)
    return out


def expected_key(r):
    return (r["name"] + "_" + r["unique_key"]).encode() if r["name"] and r["unique_key"] else b""


def check_decoded(arr, base, reqs, now, is_owner=1):
    for j, r in enumerate(reqs):
        i = base + j
        assert arr["keys"][i] == expected_key(r), (i, r)
        assert arr["hits"][i] == r["hits"] and arr["limit"][i] == r["limit"] and arr["duration"][i] == r["duration"]
        assert arr["burst"][i] == r["burst"]
        assert arr["created_at"][i] == (r["created_at"] or now)             # gubernator.go:218-220
        assert arr["algorithm"][i] == (r["algorithm"] if r["algorithm"] in (0, 1) else 255)
        assert arr["behavior"][i] == r["behavior"]
        assert arr["is_owner"][i] == is_owner


def test_wire_symbols_exported_and_declared():
    hdr = open(os.path.join(support.ROOT, "include", "guber_wire.h")).read()
    declared = set(re.findall(r"\b(guber_wire_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(gw.WIRE_SYMBOLS), declared ^ set(gw.WIRE_SYMBOLS)
    L = ga.lib()
    for s in gw.WIRE_SYMBOLS:
        assert hasattr(L, s), s


def test_decode_matches_protobuf_runtime_and_aggregates_payloads():
    rng = np.random.default_rng(11)
    wb = gw.WireBatch(max_items=4096, max_key_bytes=1 << 18)
    wb.reset(NOW)
    all_reqs, slices = [], []
    for rpc in range(12):
        reqs = rand_reqs(rng, int(rng.integers(0, 200)))
        payload = wire_replay.pb_request(reqs, peer=bool(rpc & 1))
        # map entries (metadata, field 9) and unknown fields are skipped
        m = PB["GetRateLimitsReq"]()
        m.ParseFromString(payload)
        for q in m.requests[:3]:
            q.metadata["traceparent"] = "00-abc-def-01"
        payload = m.SerializeToString() + bytes([0x78, 0x05]) + bytes([0x82, 0x01, 0x02, 0x41, 0x42])   # fields 15 (varint), 16 (LEN)
        first, count = wb.decode(payload, is_owner=not (rpc & 1))
        assert (first, count) == (len(all_reqs), len(reqs))
        slices.append((first, reqs, 0 if rpc & 1 else 1))
        all_reqs += reqs
    assert len(wb) == len(all_reqs)
    arr = wb.arrays()
    for first, reqs, own in slices:
        check_decoded(arr, first, reqs, NOW, own)
    pre = wb.pre_errors()
    want = [1 if not r["unique_key"] else 2 if not r["name"] else 0 for r in all_reqs]
    assert pre.tolist() == want
    v = wb.view()
    assert v.n == len(all_reqs) and v.now_ms == NOW
    wb.close()


def test_last_value_wins_and_field_order_is_free():
    a = PB["RateLimitReq"](name="old", unique_key="k", hits=1, limit=2)
    b = PB["RateLimitReq"](name="new", hits=7, duration=9)
    body = a.SerializeToString() + b.SerializeToString()         # concatenation = merge: later singular fields win
    payload = bytes([0x0a, len(body)]) + body
    ref = PB["GetRateLimitsReq"]()
    ref.ParseFromString(payload)
    assert (ref.requests[0].name, ref.requests[0].hits, ref.requests[0].limit) == ("new", 7, 2)
    wb = gw.WireBatch(16, 1024)
    wb.reset(NOW)
    assert wb.decode(payload) == (0, 1)
    arr = wb.arrays()
    assert arr["keys"][0] == b"new_k" and arr["hits"][0] == 7 and arr["limit"][0] == 2 and arr["duration"][0] == 9
    wb.close()


def test_malformed_capacity_and_rpc_cap_append_nothing():
    rng = np.random.default_rng(3)
    reqs = rand_reqs(rng, 40, bad=False)
    payload = wire_replay.pb_request(reqs)
    wb = gw.WireBatch(max_items=64, max_key_bytes=4096)
    wb.reset(NOW)
    assert wb.decode(payload) == (0, 40)
    # every strict prefix is either a shorter valid message (cut at a record boundary) or malformed; a failed decode appends nothing
    ok = bad = 0
    for cut in range(len(payload)):
        wb.reset(NOW)
        wb.decode(payload[:0])
        try:
            first, count = wb.decode(payload[:cut])
            ref = PB["GetRateLimitsReq"]()
            ref.ParseFromString(payload[:cut])
            assert count == len(ref.requests) == len(wb)
            ok += 1
        except ga.GuberError as e:
            assert e.code == gw.E_WIRE_MALFORMED
            assert len(wb) == 0
            with pytest.raises(Exception):
                PB["GetRateLimitsReq"]().ParseFromString(payload[:cut])
            bad += 1
    assert ok >= 40 and bad > 100
    # invalid UTF-8 in a string field, over-long varint, zero field number
    for raw in (bytes([0x0a, 0x04, 0x0a, 0x02, 0xc3, 0x28]), bytes([0x0a, 0x0c, 0x18] + [0xff] * 10 + [0x01]), bytes([0x00, 0x00])):
        before = len(wb)
        with pytest.raises(ga.GuberError) as ei:
            wb.decode(raw)
        assert ei.value.code == gw.E_WIRE_MALFORMED and len(wb) == before
    # capacity: items, then key bytes
    wb.reset(NOW)
    wb.decode(payload)
    with pytest.raises(ga.GuberError) as ei:
        wb.decode(payload)
    assert ei.value.code == gw.E_WIRE_FULL and len(wb) == 40
    small = gw.WireBatch(max_items=64, max_key_bytes=100)
    small.reset(NOW)
    with pytest.raises(ga.GuberError) as ei:
        small.decode(payload)
    assert ei.value.code == gw.E_WIRE_FULL and len(small) == 0
    # the reference's per-RPC cap (gubernator.go:189-193)
    big = wire_replay.pb_request(rand_reqs(rng, 1001, bad=False))
    cap = gw.WireBatch(4096, 1 << 18)
    cap.reset(NOW)
    with pytest.raises(ga.GuberError) as ei:
        cap.decode(big, max_per_rpc=1000)
    assert ei.value.code == gw.E_WIRE_TOO_LARGE and len(cap) == 0
    assert "max size is '1000'" in str(ei.value)
    assert cap.decode(big, max_per_rpc=0) == (0, 1001)
    for w in (wb, small, cap):
        w.close()


def test_encode_is_byte_identical_to_protobuf_runtime():
    rng = np.random.default_rng(5)
    reqs = rand_reqs(rng, 300)
    wb = gw.WireBatch(1024, 1 << 16)
    wb.reset(NOW)
    wb.decode(wire_replay.pb_request(reqs))
    n = len(wb)
    res = wb.result()
    def arr(ptr, dt):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(n,))
    status, limit, remaining, reset, err = (arr(res.status, C.c_uint8), arr(res.limit, C.c_int64), arr(res.remaining, C.c_int64),
                                             arr(res.reset_time, C.c_int64), arr(res.err, C.c_uint8))
    status[:] = rng.integers(0, 2, n)
    limit[:] = rng.choice([0, 1, 100, -5, 2**62, -2**63], n)
    remaining[:] = rng.choice([0, 0, 7, -1, 2**63 - 1], n)
    reset[:] = rng.choice([0, NOW + 60000, -9], n)
    err[:] = 0
    pre = wb.pre_errors()
    for i, r in enumerate(reqs):
        if pre[i] == 0 and r["algorithm"] not in (0, 1):
            err[i] = 1                                            # GUBER_ITEM_E_INVALID_ALGORITHM, as the engine reports it
    for wrap in (1, 0):
        want = PB["GetRateLimitsResp"]()
        for i, r in enumerate(reqs):
            o = want.responses.add()
            if pre[i] == 1:
                o.error = "field 'unique_key' cannot be empty"
            elif pre[i] == 2:
                o.error = "field 'namespace' cannot be empty"
            elif err[i]:
                msg = "Invalid rate limit algorithm '%d'" % r["algorithm"]              # workers.go:318
                o.error = ("Error while apply rate limit for '%s_%s': %s" % (r["name"], r["unique_key"], msg)) if wrap else msg
            else:
                o.status, o.limit, o.remaining, o.reset_time = int(status[i]), int(limit[i]), int(remaining[i]), int(reset[i])
        got = wb.encode(0, n, wrap_errors=bool(wrap))
        assert got == want.SerializeToString(deterministic=True)
        # a slice in the middle (one RPC's share of an aggregated batch), parsed as the peer response message
        sl = wb.encode(17, 40, wrap_errors=bool(wrap))
        peer = PB["GetPeerRateLimitsResp"]()
        peer.ParseFromString(sl)
        assert [x.SerializeToString(deterministic=True) for x in peer.rate_limits] == \
               [x.SerializeToString(deterministic=True) for x in want.responses[17:57]]
    # too small an output buffer: GUBER_E_NOMEM and the size needed
    L = ga.lib()
    need = C.c_size_t()
    buf = (C.c_uint8 * 8)()
    assert L.guber_wire_encode_responses(wb.h, 0, n, 0, buf, 8, C.byref(need)) == -6 and need.value == len(got)
    wb.close()


def test_gregorian_items_get_host_precomputed_calendar_values():
    wb = gw.WireBatch(16, 1024)
    wb.reset(NOW)
    reqs = [dict(name="g", unique_key="k%d" % d, hits=1, limit=10, duration=d, algorithm=0, behavior=4, burst=0) for d in (0, 1, 2, 4, 5, 3, 99)]
    wb.decode(wire_replay.pb_request(reqs))
    arr = wb.arrays()
    for i, r in enumerate(reqs):
        e, d = support.gregorian(NOW, r["duration"])
        assert (arr["greg_expire"][i], arr["greg_duration"][i]) == (e, d), r
    wb.close()


def test_golden_functional_scenarios_through_the_wire_with_the_oracle():
    def make_eval():
        o = support.Oracle(cache_size=1 << 16)
        def evaluate(wb):
            o.lib.oracle_eval_batch(o.h, C.byref(wb.view()), C.byref(wb.result()))
        return evaluate, (lambda: None)
    n = wire_replay.run_functional_wire(lambda: gw.WireBatch(2048, 1 << 16), make_eval)
    assert n >= 80


def _decode_or_none(wb, payload):
    wb.reset(NOW)
    try:
        first, count = wb.decode(payload)
    except ga.GuberError as e:
        assert e.code == gw.E_WIRE_MALFORMED, e
        assert len(wb) == 0
        return None
    return wb.arrays()


def _reference_view(payload):
    m = PB["GetRateLimitsReq"]()
    try:
        m.ParseFromString(payload)
    except Exception:
        return None
    return m


REQ_NESTED = {0: (1,), 1: (9,)}                 # GetRateLimitsReq.requests, RateLimitReq.metadata entries
GLOBALS_NESTED = {0: (1,), 1: (2,), 2: (6,)}     # UpdatePeerGlobalsReq.globals, UpdatePeerGlobal.status, RateLimitResp.metadata entries


def _has_overflowing_varint(buf, depth=0, nested=REQ_NESTED):
    """True when some varint of the message occupies 10 bytes with a last byte >= 2 (more than 64 bits).  The
    reference's runtime (google.golang.org/protobuf v1.32.0, go.mod:32: protowire.ConsumeVarint) rejects such
    input as overflow — and so does the transcoder — while the python runtime truncates silently.  Only called on
    payloads the python runtime accepted, so the structure is known to be walkable."""
    pos = 0

    def varint():
        nonlocal pos
        n = 0
        while True:
            c = buf[pos]; pos += 1; n += 1
            if not c & 0x80:
                return n == 10 and c >= 2
    while pos < len(buf):
        start = pos
        if varint():
            return True
        tag = 0
        for k, c in enumerate(buf[start:pos]):
            tag |= (c & 0x7f) << (7 * k)
        wt, field = tag & 7, tag >> 3
        if wt == 0:
            if varint():
                return True
        elif wt == 1:
            pos += 8
        elif wt == 5:
            pos += 4
        elif wt == 2:
            s0 = pos
            if varint():
                return True
            ln = 0
            for k, c in enumerate(buf[s0:pos]):
                ln |= (c & 0x7f) << (7 * k)
            body = buf[pos:pos + ln]
            pos += ln
            # nested messages the schema knows (their content is parsed by the runtimes, unknown LEN fields are not)
            if field in nested.get(depth, ()):
                if _has_overflowing_varint(body, depth + 1, nested):
                    return True
        # groups: the fuzz corpus only has them at the top level with plain varint content; tags were checked above
    return False


def test_fuzzed_payloads_agree_with_the_protobuf_runtime():
    """Mutated payloads (bit flips, byte inserts / deletes, truncation, spliced garbage, unknown groups): the
    transcoder must accept exactly what the python protobuf runtime accepts, decode it to the same values, and
    never touch memory it should not (the batch is sized tightly)."""
    rng = np.random.default_rng(101)
    wb = gw.WireBatch(max_items=256, max_key_bytes=1 << 15)
    base = [wire_replay.pb_request(rand_reqs(rng, int(rng.integers(1, 12)))) for _ in range(40)]
    # unknown group field 20 wrapping a varint field and a nested group 21, then a normal record
    grp = bytes([0xa3, 0x01, 0x08, 0x05, 0xab, 0x01, 0xac, 0x01, 0xa4, 0x01])
    base.append(grp + base[0])
    base.append(base[1] + bytes([0xa3, 0x01, 0x08]))            # unterminated group
    base.append(bytes([0xa4, 0x01]) + base[2])                   # stray END_GROUP
    accepted = rejected = overflow = 0
    for it in range(6000):
        p = bytearray(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 4))):
            if not p:
                break
            op = rng.integers(0, 5)
            i = int(rng.integers(0, len(p)))
            if op == 0:
                p[i] ^= 1 << int(rng.integers(0, 8))
            elif op == 1:
                p.insert(i, int(rng.integers(0, 256)))
            elif op == 2:
                del p[i]
            elif op == 3:
                del p[i:]
            else:
                p[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        p = bytes(p)
        got, ref = _decode_or_none(wb, p), _reference_view(p)
        if got is None and ref is not None and _has_overflowing_varint(p):
            overflow += 1                                        # Go and the transcoder reject, python truncates
            continue
        assert (got is None) == (ref is None), (it, p.hex())
        if ref is None:
            rejected += 1
            continue
        accepted += 1
        assert got["n"] == len(ref.requests), (it, p.hex())
        for i, q in enumerate(ref.requests):
            want_key = (q.name + "_" + q.unique_key).encode() if q.name and q.unique_key else b""
            assert got["keys"][i] == want_key, (it, p.hex())
            assert (got["hits"][i], got["limit"][i], got["duration"][i], got["burst"][i]) == (q.hits, q.limit, q.duration, q.burst), (it, p.hex())
            assert got["created_at"][i] == (q.created_at or NOW)
            assert got["behavior"][i] == q.behavior & 0xffffffff
            assert got["algorithm"][i] == (q.algorithm if q.algorithm in (0, 1) else 255)
    assert accepted > 1500 and rejected > 1500 and overflow < 60, (accepted, rejected, overflow)
    wb.close()


def test_update_peer_globals_codec_against_the_protobuf_runtime():
    """UpdatePeerGlobalsReq (peers.proto:51-63): the sender's bytes equal the protobuf runtime's for the message
    broadcastPeers builds (global.go:234-262), and the receiver's decoded items are the CacheItems UpdatePeerGlobals
    constructs (gubernator.go:425-459)."""
    from gubernator_amd.abi import HostBatch, HostResult, item_dict
    rng = np.random.default_rng(8)
    n = 500
    keys = [("glob_%d" % i if i % 50 else "ключ_%d" % i).encode() for i in range(n)]
    algo = rng.integers(0, 2, n).astype(np.uint8)
    duration = rng.choice([0, 1, 60_000, -5, 2**50], n).astype(np.int64)
    created = rng.choice([0, NOW, NOW - 3, 12345], n).astype(np.int64)
    b = HostBatch(keys, 0, 10, duration, NOW, created_at=created, algorithm=algo)
    st = HostResult(n)
    st.status[:n] = rng.integers(0, 2, n)
    st.limit[:n] = rng.choice([0, 10, 100, -1], n)
    st.remaining[:n] = rng.choice([0, 5, 99, -7, 2**62], n)
    st.reset_time[:n] = rng.choice([0, NOW + 60_000, NOW - 1], n)
    st.err[:n] = (rng.random(n) < 0.05).astype(np.uint8)            # a few failed status reads: skipped by the sender
    got = gw.encode_globals(b, st)
    want = PB["UpdatePeerGlobalsReq"]()
    sent = []
    for i in range(n):
        if st.err[i]:
            continue
        g = want.globals.add(key=keys[i].decode(), algorithm=int(algo[i]), duration=int(duration[i]), created_at=int(created[i]))
        g.status.SetInParent()
        g.status.status, g.status.limit, g.status.remaining, g.status.reset_time = int(st.status[i]), int(st.limit[i]), int(st.remaining[i]), int(st.reset_time[i])
        sent.append(i)
    assert got == want.SerializeToString(deterministic=True)
    # receiver: decode what the runtime serialised (plus metadata / unknown fields, which are ignored)
    for g in want.globals[:5]:
        g.status.metadata["owner"] = "10.0.0.1:81"
    payload = want.SerializeToString() + bytes([0x78, 0x01])
    wi = gw.WireItems(1024, 1 << 16)
    now2 = NOW + 77
    items, cnt = wi.decode(payload, now2)
    assert cnt == len(sent)
    for j, i in enumerate(sent):
        d = item_dict(items[j])
        assert d["key"] == keys[i] and d["algorithm"] == algo[i] and d["expire_at"] == st.reset_time[i] and d["duration"] == duration[i]
        assert d["stamp"] == now2 and d["limit"] == st.limit[i] and d["invalid_at"] == 0
        if algo[i] == 1:   # gubernator.go:435-442
            assert d["remaining_f"] == float(st.remaining[i]) and d["burst"] == st.limit[i] and d["remaining"] == 0 and d["status"] == 0
        else:              # :443-451
            assert d["remaining"] == st.remaining[i] and d["status"] == st.status[i] and d["burst"] == 0 and d["remaining_f"] == 0.0
    # malformed payloads are rejected, capacity is reported
    with pytest.raises(ga.GuberError) as ei:
        wi.decode(payload[:-3] + b"\xff", now2)
    assert ei.value.code == gw.E_WIRE_MALFORMED
    small = gw.WireItems(10, 1 << 16)
    with pytest.raises(ga.GuberError) as ei:
        small.decode(payload, now2)
    assert ei.value.code == gw.E_WIRE_FULL
    # the oracle installs these items exactly as it installs hand-built ones: a replica answers from them
    o = support.Oracle(cache_size=1 << 12)
    for j in range(cnt):
        o.add_item(items[j], now2)
    k = sent[0]
    it = o.get_item(keys[k], now2)
    if st.reset_time[k] >= now2:
        assert it is not None and it["limit"] == st.limit[k]
    wi.close(); small.close()


def test_fuzzed_update_peer_globals_agree_with_the_protobuf_runtime():
    rng = np.random.default_rng(303)
    wi = gw.WireItems(256, 1 << 15)
    base = []
    for _ in range(30):
        m = PB["UpdatePeerGlobalsReq"]()
        for i in range(int(rng.integers(1, 8))):
            g = m.globals.add(key="k%d_ü" % rng.integers(0, 99), algorithm=int(rng.choice([0, 1, 1, 5])), duration=int(rng.choice([0, 60000, -1])),
                              created_at=int(rng.choice([0, NOW])))
            g.status.SetInParent()
            g.status.status, g.status.limit, g.status.remaining, g.status.reset_time = int(rng.integers(0, 2)), int(rng.choice([0, 10, -3])), \
                int(rng.choice([0, 7, 2**62])), int(rng.choice([0, NOW + 5]))
            if rng.random() < 0.3:
                g.status.error = "é"
                g.status.metadata["a"] = "b"
        base.append(m.SerializeToString())
    acc = rej = ovf = 0
    for it in range(4000):
        p = bytearray(base[int(rng.integers(0, len(base)))])
        for _ in range(int(rng.integers(0, 4))):
            if not p:
                break
            op, i = rng.integers(0, 5), int(rng.integers(0, len(p)))
            if op == 0:
                p[i] ^= 1 << int(rng.integers(0, 8))
            elif op == 1:
                p.insert(i, int(rng.integers(0, 256)))
            elif op == 2:
                del p[i]
            elif op == 3:
                del p[i:]
            else:
                p[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        p = bytes(p)
        ref = PB["UpdatePeerGlobalsReq"]()
        try:
            ref.ParseFromString(p)
        except Exception:
            ref = None
        try:
            items, cnt = wi.decode(p, NOW)
            got = [(bytes(C.string_at(items[j].key, items[j].key_len)), items[j].algorithm, items[j].duration, items[j].expire_at) for j in range(cnt)]
        except ga.GuberError as e:
            assert e.code == gw.E_WIRE_MALFORMED, e
            got = None
        if got is None and ref is not None and _has_overflowing_varint(p, nested=GLOBALS_NESTED):
            ovf += 1
            continue
        assert (got is None) == (ref is None), (it, p.hex())
        if ref is None:
            rej += 1
            continue
        acc += 1
        assert len(got) == len(ref.globals)
        for (k, a, d, ex), g in zip(got, ref.globals):
            assert k == g.key.encode() and d == (g.duration if g.algorithm in (0, 1) else 0) and ex == g.status.reset_time, (it, p.hex())
            assert a == (g.algorithm if 0 <= g.algorithm <= 254 else 255)
    assert acc > 800 and rej > 800 and ovf < 60, (acc, rej, ovf)
    wi.close()


def test_wire_parser_is_memory_safe_under_asan_fuzz(tmp_path):
    """tools/wire_fuzz_asan.cpp: 300 000 mutated payloads through decode / encode, compiled with -fsanitize=address,undefined
    against exact-size heap buffers — any out-of-bounds access or undefined behaviour aborts the run.  The same run drives the DEVICE
    decoder's framing logic (guber_kernels_wire.h scan_toplevel + the shared record parser) over a bounds-checked byte source and
    requires the host transcoder's verdict and item count on every payload."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "wire_fuzz")
    root = support.ROOT
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-Wno-attributes", "-Wno-unknown-pragmas", "-I", os.path.join(root, "include"),
                    "-I", os.path.join(root, "tests", "hostsim", "fakehip"), os.path.join(root, "tools", "wire_fuzz_asan.cpp"), os.path.join(root, "gubernator_amd", "csrc", "wire.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe, "300000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "no sanitizer report" in out.stdout
