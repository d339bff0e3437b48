"""The payload stage (include/guber_wire.h guber_wire_pool_*, gubernator_amd/csrc/guber_wire_pool.h): caller threads hand over the SERIALIZED
GetRateLimitsReq of their RPC and get the serialized GetRateLimitsResp back — V1Instance.GetRateLimits (gubernator.go:183-306) with the
unmarshalling, the validation, HashKey, the worker's choice, the evaluation and the answers' order on the device.  Checked byte for byte
against the host transcoder (csrc/wire.cpp, itself byte-identical to the protobuf runtime: tests/test_wire_cpu.py) around the ORACLE, on
the reference's golden functional scenarios, and — with concurrent callers, where no serial order exists — by per-key conservation.
Runs unchanged against the CPU build of the engine under AddressSanitizer (tests/test_enginesim_cpu.py)."""
import ctypes as C
import threading

import numpy as np
import pytest

import gubernator_amd as ga
import scenarios
import support
import wire_replay
from gubernator_amd import wire as gw
from test_wire_cpu import NOW, rand_reqs

pytestmark = pytest.mark.gpu

SMALL = dict(stages=3, max_items=8192, max_payload_bytes=1 << 20, max_rpcs=64)


def _engines(n, **kw):
    e0 = ga.Engine(**kw)
    return [e0] + [ga.Engine(stream=e0.stream_handle(), **kw) for _ in range(n - 1)]


def _expected(o, wb, payload, now, wrap=True):
    """the host transcoder around the oracle: the bytes the pool must return (or the code it must turn the message away with)"""
    wb.reset(now)
    first, count = wb.decode(payload, max_per_rpc=1000)
    o.lib.oracle_eval_batch(o.h, C.byref(wb.view()), C.byref(wb.result()))
    return wb.encode(first, count, wrap_errors=wrap)


@pytest.mark.parametrize("n_engines", [1, 4])
def test_one_caller_gets_the_host_transcoders_bytes_around_the_oracle(n_engines):
    """RPC after RPC from one thread (a serial order: the oracle can follow): requests with every field, empty names and keys
    (gubernator.go:208-217), algorithms that do not exist (workers.go:318), GLOBAL / RESET_REMAINING / DRAIN_OVER_LIMIT, keys of many
    widths.  Response bytes equal the host transcoder's around ONE oracle (the caches do not bind: W workers answer as one)."""
    rng = np.random.default_rng(31 + n_engines)
    engs = _engines(n_engines, cache_size=1 << 16, max_batch=8192, max_key_bytes=256)
    place = ga.Placement(n_engines) if n_engines > 1 else None
    pool = gw.WirePool(engs, place, **SMALL)
    o = support.Oracle(cache_size=1 << 20)
    wb = gw.WireBatch(4096, 1 << 20)
    now = NOW
    errors_seen = 0
    for k in range(60):
        reqs = rand_reqs(rng, int(rng.integers(1, 700)), bad=(k % 3 == 0))
        payload = wire_replay.pb_request(reqs, peer=bool(k & 1))
        pool.set_clock(now)
        got = pool.get_rate_limits(payload, wrap_errors=not (k & 1))
        want = _expected(o, wb, payload, now, wrap=not (k & 1))
        assert got == want, f"RPC {k}: {len(reqs)} requests"
        errors_seen += sum(1 for row in wire_replay.rows_of(got) if row[4])
        now += int(rng.integers(0, 900))
    assert errors_seen > 50                                      # the error path (texts built from the caller's own payload) was exercised
    assert pool.get_rate_limits(b"") == b""                      # no requests: an empty response
    st = pool.stats()
    assert st["rpcs"] == 60 and st["stages"] == 60               # one caller: every stage left because the decoder was idle, or BatchWait
    # (the oracle, handed the host transcoder's batch, gives the items that failed validation — an empty key — a bucket of their own; the
    #  engines never let them near one)
    assert sum(e.size() for e in engs) == o.size() - 1
    pool.close()
    for e in reversed(engs):
        e.close()
    if place:
        place.close()
    o.close(); wb.close()


@pytest.mark.parametrize("n_engines", [1, 4])
def test_a_small_rpc_in_an_idle_pool_is_evaluated_by_its_caller(n_engines):
    """guber_wire_pool.h wpl_direct: while (almost) nobody else is inside the pool an RPC of at most FOUR requests skips the stages — host
    transcoder, the placement's rule on the host (the table k_fr_count would pick: the buckets are shared with the stages), the engine's
    one-launch path (request by request when the RPC's requests live on several tables), host transcoder.  RPCs of 1 .. 4 requests (every
    kind of request, errors included) between RPCs of 5 .. 39 that go through the stages, on the same keys: every response equals the
    host transcoder's around ONE oracle fed the same sequence."""
    rng = np.random.default_rng(91 + n_engines)
    engs = _engines(n_engines, cache_size=1 << 16, max_batch=8192, max_key_bytes=256)
    place = ga.Placement(n_engines) if n_engines > 1 else None
    pool = gw.WirePool(engs, place, **SMALL)
    o = support.Oracle(cache_size=1 << 20)
    wb = gw.WireBatch(4096, 1 << 20)
    now = NOW
    singles = 0
    for k in range(240):
        n = (1 if k < 3 else int(rng.integers(1, 5))) if k % 4 != 3 else int(rng.integers(5, 40))   # (up to four requests: the caller's own)
        reqs = rand_reqs(rng, n, bad=(k % 3 == 0))
        for r in reqs:                                           # few keys: the two paths meet on the same buckets all the time
            if r["unique_key"].startswith("acct:"):
                r["unique_key"] = "acct:%d" % rng.integers(0, 12)
        payload = wire_replay.pb_request(reqs, peer=bool(k & 1))
        pool.set_clock(now)
        got = pool.get_rate_limits(payload, wrap_errors=not (k & 2))
        want = _expected(o, wb, payload, now, wrap=not (k & 2))
        assert got == want, f"RPC {k}: {n} request(s) {reqs[:1]}"
        singles += n <= 4
        now += int(rng.integers(0, 400))
        if k == 2:                                               # three one-request RPCs so far: no stage has been through the decoder
            st = pool.stats()
            assert st["rpcs"] == 3 and st["host_decode_ns"] == 0 and st["decode_us_sum"] == 0
    st = pool.stats()
    assert st["host_decode_ns"] > 0                              # (the RPCs of several requests went through the stages)
    assert st["rpcs"] == 240 and st["stages"] == 240 and singles == 180   # (an RPC evaluated by its caller counts as a batch of its own)
    with pytest.raises(ga.GuberError) as ei:                     # a truncated one-request message: the runtimes' verdict, from this path too
        pool.get_rate_limits(wire_replay.pb_request(rand_reqs(rng, 1, bad=False))[:-2])
    assert ei.value.code == -20
    pool.close()
    for e in reversed(engs):
        e.close()
    if place:
        place.close()
    o.close(); wb.close()


def test_long_rpcs_are_encoded_on_the_device_in_pieces():
    """k_wire_enc (csrc/guber_kernels_wire.h) writes an RPC's GetRateLimitsResp in pieces of 1 024 items, the bytes that do not fill a
    16-byte word carried into the next piece: RPCs of 1 .. 4 096 items (the 1000-item cap lifted), among them exactly 1 024, 1 025, 2 048
    and 4 096, without item errors (the device's encoding) and with (the host transcoder's, from the raw answers) — the host transcoder's
    bytes around the oracle, byte for byte; buckets run empty on the way (status and remaining change the items' widths)."""
    rng = np.random.default_rng(77)
    engs = _engines(2, cache_size=1 << 16, max_batch=8192, max_key_bytes=256)
    place = ga.Placement(2)
    pool = gw.WirePool(engs, place, stages=3, max_items=8192, max_payload_bytes=2 << 20, max_rpcs=64, max_per_rpc=0xffffffff)
    o = support.Oracle(cache_size=1 << 20)
    wb = gw.WireBatch(4096, 2 << 20)
    now = NOW
    sizes = [1, 2, 15, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 4095, 4096] + [int(x) for x in rng.integers(1025, 4097, 6)]
    for k, n in enumerate(sizes):
        with_errors = k % 5 == 4
        reqs = rand_reqs(rng, n, bad=with_errors)
        if not with_errors:                                      # (rand_reqs' DURATION_IS_GREGORIAN requests carry durations that are no interval: item errors)
            for r in reqs:
                r["behavior"] = int(rng.choice([0, 0, 2, 8, 32, 34, 1]))
        payload = wire_replay.pb_request(reqs)
        pool.set_clock(now)
        wb.reset(now)
        first, count = wb.decode(payload, max_per_rpc=0)
        assert count == n
        o.lib.oracle_eval_batch(o.h, C.byref(wb.view()), C.byref(wb.result()))
        want = wb.encode(first, count, wrap_errors=True)
        got = pool.get_rate_limits(payload)
        assert got == want, f"RPC {k}: {n} requests"
        assert with_errors == any(row[4] for row in wire_replay.rows_of(got))   # (which of the two encoders this RPC went through)
        now += int(rng.integers(0, 2000))
    # a message that holds no RateLimitReq at all (an unknown varint field): alone in its stage nothing is evaluated and no kernel runs — an
    # empty response, not what an earlier RPC left at that place of the stage
    for _ in range(4):
        assert pool.get_rate_limits(b"\x78\x01") == b""
    pool.close()
    for e in reversed(engs):
        e.close()
    place.close()
    o.close(); wb.close()


def test_messages_that_are_turned_away_whole():
    """a truncated message (the protobuf runtime's error), more than 1000 requests (gubernator.go:189-193): the call says so, nothing of the
    message is evaluated, and the RPCs that shared its stage are answered as if it had not been there"""
    rng = np.random.default_rng(5)
    engs = _engines(2, cache_size=1 << 16, max_batch=8192, max_key_bytes=128)
    place = ga.Placement(2)
    pool = gw.WirePool(engs, place, **SMALL)
    pool.set_clock(NOW)
    good = wire_replay.pb_request(rand_reqs(rng, 50, bad=False))
    with pytest.raises(ga.GuberError) as ei:
        pool.get_rate_limits(good[:-3])
    assert ei.value.code == -20                                  # GUBER_E_WIRE_MALFORMED
    with pytest.raises(ga.GuberError) as ei:
        pool.get_rate_limits(wire_replay.pb_request(rand_reqs(rng, 1001, bad=False)))
    assert ei.value.code == -21                                  # GUBER_E_WIRE_TOO_LARGE
    assert sum(e.size() for e in engs) == 0                      # nothing reached a bucket
    with pytest.raises(ga.GuberError) as ei:
        pool.get_rate_limits(bytes(2 << 20))                     # larger than a stage
    assert ei.value.code == -22                                  # GUBER_E_WIRE_FULL
    assert len(wire_replay.rows_of(pool.get_rate_limits(good))) == 50
    pool.close()
    for e in reversed(engs):
        e.close()
    place.close()


def test_golden_functional_scenarios_through_the_payload_stage():
    """tests/golden/functional_vectors.json (the reference's functional_test.go tables): request -> protobuf runtime -> payload stage ->
    protobuf runtime -> the reference's expectations"""
    n_checked = 0
    for sc in scenarios.load("functional_vectors.json")["scenarios"]:
        engs = _engines(2, cache_size=1 << 12, max_batch=2048, max_key_bytes=128)
        place = ga.Placement(2)
        pool = gw.WirePool(engs, place, stages=2, max_items=2048, max_payload_bytes=1 << 18, max_rpcs=8)
        now = sc["start_ms"]
        steps = [([s["req"]], [s["expect"]], s["advance_ms"]) for s in sc.get("steps", [])]
        steps += [(s["reqs"], s["expect"], s["advance_ms"]) for s in sc.get("batch_steps", [])]
        for si, (reqs, expects, adv) in enumerate(steps):
            pool.set_clock(now)
            rows = wire_replay.rows_of(pool.get_rate_limits(wire_replay.pb_request(reqs)))
            assert len(rows) == len(reqs)
            for j, (exp, row) in enumerate(zip(expects, rows)):
                where = f"{sc['name']} step {si}[{j}] ({sc['source']}) via the payload stage"
                status, limit, remaining, reset_time, error = row
                if exp.get("error"):
                    assert error == exp["error"], f"{where}: {error!r}"
                    assert (status, limit, remaining, reset_time) == (0, 0, 0, 0), where
                else:
                    assert error == "", f"{where}: unexpected error {error!r}"
                    scenarios.check_expect(exp, (status, limit, remaining, reset_time, 0), now, where)
                n_checked += 1
            now += adv
        pool.close()
        for e in reversed(engs):
            e.close()
        place.close()
    assert n_checked >= 80


@pytest.mark.parametrize("callers,stages", [(12, 3), (24, 2)])
def test_concurrent_callers_share_stages_and_every_hit_is_applied_exactly_once(callers, stages):
    """caller threads at once (ctypes releases the GIL inside the call): RPCs share stages, stages follow each other through the GPU.  No
    serial order exists to replay, so what every serialisation of the reference implies is checked per key over ALL answers (token bucket,
    hits 1, limit 40, one window: algorithms.go:162-198): admitted <= limit, the admitted hits' `remaining` are exactly
    {limit-1 .. limit-admitted}, a refusal says remaining 0 and comes only once the window's tokens are gone; and every RPC gets exactly its
    own answers, in its requests' order (each caller's RPCs carry a key of the caller's own at a position of its own)."""
    engs = _engines(4, cache_size=1 << 16, max_batch=8192, max_key_bytes=64)
    place = ga.Placement(4)
    pool = gw.WirePool(engs, place, stages=stages, max_items=8192, max_payload_bytes=1 << 20, max_rpcs=64, batch_wait_us=300, decodes_queued=1)
    pool.set_clock(NOW)
    LIMIT, RPCS, ITEMS, KEYS = 40, 25, 300, callers * 200      # (37.5 hits a key on average: some keys run dry, some do not)
    # (the payloads are drawn and serialized before the threads start and the answers are parsed after they have ended: inside the threads
    #  there is the call and nothing else, so the calls really overlap — the interpreter's lock is released inside them)
    plans = []
    for t in range(callers):
        rng = np.random.default_rng(100 + t)
        mine = []
        for q in range(RPCS):
            ks = rng.integers(0, KEYS, ITEMS)
            reqs = [dict(name="conc", unique_key="k%04d" % k, hits=1, limit=LIMIT, duration=3_600_000, algorithm=0, behavior=0) for k in ks]
            pos = int(rng.integers(0, ITEMS))
            reqs[pos] = dict(name="own", unique_key="caller%02d" % t, hits=1, limit=1_000_000, duration=3_600_000, algorithm=0, behavior=0)
            mine.append((ks, pos, wire_replay.pb_request(reqs)))
        plans.append(mine)
    raw = [[None] * RPCS for _ in range(callers)]
    failures = []
    start = threading.Barrier(callers)

    def caller(t):
        try:
            start.wait()
            for q, (_, _, payload) in enumerate(plans[t]):
                raw[t][q] = pool.get_rate_limits(payload)
        except Exception as e:  # noqa: BLE001
            failures.append((t, repr(e)))

    # two more callers whose messages are turned away whole, into the same stages: a truncated message and one of 1001 requests in turn (its keys are
    # the others' keys with a limit of 1: had any of it been evaluated, conservation below would not hold)
    rng_bad = np.random.default_rng(7)
    too_many = wire_replay.pb_request([dict(name="conc", unique_key="k%04d" % k, hits=1, limit=1, duration=3_600_000, algorithm=0, behavior=0)
                                       for k in rng_bad.integers(0, KEYS, 1001)])
    truncated = plans[0][0][2][:-5]
    turned_away = []

    def bad_caller(t):
        try:
            start.wait()
            for q in range(RPCS):
                try:
                    pool.get_rate_limits(too_many if (q + t) & 1 else truncated)
                    failures.append((t, "a message that must be turned away was answered"))
                except ga.GuberError as e:
                    turned_away.append(e.code)
        except Exception as e:  # noqa: BLE001
            failures.append((t, repr(e)))

    start = threading.Barrier(callers + 2)
    th = [threading.Thread(target=caller, args=(t,)) for t in range(callers)] + [threading.Thread(target=bad_caller, args=(t,)) for t in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert sorted(set(turned_away)) == [-21, -20] and len(turned_away) == 2 * RPCS, turned_away[:8]
    assert not failures, failures[:3]
    admitted = np.zeros(KEYS, np.int64); refused = np.zeros(KEYS, np.int64); sum_rem = np.zeros(KEYS, np.int64)
    for t in range(callers):
        for q, (ks, pos, _) in enumerate(plans[t]):
            rows = wire_replay.rows_of(raw[t][q])
            assert len(rows) == ITEMS
            # the caller's own key: its RPCs are the only ones that touch it, one after the other -> remaining counts down by one per RPC
            assert rows[pos][:3] == (0, 1_000_000, 1_000_000 - 1 - q), (t, q, rows[pos])
            for j, (k, row) in enumerate(zip(ks, rows)):
                if j == pos:
                    continue
                status, limit, remaining, reset_time, error = row
                assert error == "" and limit == LIMIT and status in (0, 1)
                if status == 0:
                    admitted[k] += 1; sum_rem[k] += remaining
                    assert 0 <= remaining < LIMIT
                else:
                    refused[k] += 1
                    assert remaining == 0
    assert (admitted <= LIMIT).all()
    assert (sum_rem == admitted * LIMIT - admitted * (admitted + 1) // 2).all()      # the i-th admitted hit left LIMIT - i
    assert (admitted[refused > 0] == LIMIT).all()
    assert refused.sum() > 0 and (admitted < LIMIT).any()
    st = pool.stats()
    assert st["rpcs"] == (callers + 2) * RPCS and st["stages"] < st["rpcs"]          # RPCs shared stages (the messages turned away count as RPCs, not as items)
    assert st["items"] == callers * RPCS * ITEMS
    pool.close()
    for e in reversed(engs):
        e.close()
    place.close()


def test_fuzzed_payloads_get_the_host_transcoders_verdict_and_bytes():
    """mutated payloads (bit flips, insertions, deletions, truncations, inserted garbage: tests/test_gpu_wire_dev.py's mutator) one after the other through
    the stage: what the host transcoder turns away (csrc/wire.cpp, itself pinned on the protobuf runtime's verdicts by tests/test_wire_cpu.py) the stage
    turns away with the same code and evaluates nothing of; what it accepts the stage answers with the same BYTES as the host transcoder around the
    oracle — whatever the mutation made of the requests (huge hits, negative limits, unknown algorithms and behaviours, empty names)"""
    from test_gpu_wire_dev import _mutate
    rng = np.random.default_rng(23)
    engs = _engines(3, cache_size=1 << 16, max_batch=8192, max_key_bytes=512)
    place = ga.Placement(3)
    pool = gw.WirePool(engs, place, stages=3, max_items=8192, max_payload_bytes=1 << 20, max_rpcs=64)
    o = support.Oracle(cache_size=1 << 20)
    wb = gw.WireBatch(8192, 4 << 20)
    base = [wire_replay.pb_request(rand_reqs(rng, int(rng.integers(1, 300)))) for _ in range(30)]
    now = NOW
    accepted = rejected = 0
    for k in range(400):
        payload = _mutate(rng, base[int(rng.integers(0, len(base)))])
        pool.set_clock(now)
        wb.reset(now)
        try:
            first, count = wb.decode(payload, max_per_rpc=1000)
        except ga.GuberError as e:
            with pytest.raises(ga.GuberError) as ei:
                pool.get_rate_limits(payload)
            assert ei.value.code == e.code, (k, e.code, ei.value.code)
            rejected += 1
            continue
        o.lib.oracle_eval_batch(o.h, C.byref(wb.view()), C.byref(wb.result()))
        assert pool.get_rate_limits(payload) == wb.encode(first, count), f"payload {k}"
        accepted += 1
        now += int(rng.integers(0, 50))
    assert accepted > 100 and rejected > 50, (accepted, rejected)
    pool.close()
    for e in reversed(engs):
        e.close()
    place.close()
    o.close(); wb.close()
