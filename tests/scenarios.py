"""Replays the reference's golden vectors (tests/golden/*.json) against any backend exposing
eval(HostBatch) -> HostResult, add_item, get_item, each, size.  Used for the CPU oracle (CPU suite)
and for the HIP engine through the C ABI (-m gpu suite)."""
import json
import os

import support
from support import HostBatch, GREGORIAN

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def validate(req):
    """gubernator.go:208-217: per-item validation the host layer performs before the hot path."""
    if req["unique_key"] == "":
        return "field 'unique_key' cannot be empty"
    if req["name"] == "":
        return "field 'namespace' cannot be empty"
    return ""


def batch_of(reqs, now_ms):
    keys = [r["name"] + "_" + r["unique_key"] for r in reqs]  # client.go:39-41 HashKey
    ge, gd = [], []
    for r in reqs:
        if r["behavior"] & GREGORIAN:
            e, d = support.gregorian(now_ms, r["duration"])
            ge.append(e)
            gd.append(d)
        else:
            ge.append(0)
            gd.append(0)
    return HostBatch(keys, [r["hits"] for r in reqs], [r["limit"] for r in reqs], [r["duration"] for r in reqs],
                     now_ms, burst=[r["burst"] for r in reqs], created_at=[now_ms] * len(reqs),
                     algorithm=[r["algorithm"] for r in reqs], behavior=[r["behavior"] for r in reqs],
                     greg_expire=ge, greg_duration=gd)


def check_expect(exp, row, now_ms, where):
    status, limit, remaining, reset_time, err = row
    if "status" in exp:
        assert status == exp["status"], f"{where}: status {status} != {exp['status']}"
    if "remaining" in exp:
        assert remaining == exp["remaining"], f"{where}: remaining {remaining} != {exp['remaining']}"
    if "limit" in exp:
        assert limit == exp["limit"], f"{where}: limit {limit} != {exp['limit']}"
    if exp.get("reset_nonzero"):
        assert reset_time != 0, f"{where}: reset_time is 0"
    if "reset_time" in exp:
        assert reset_time == exp["reset_time"], where
    if "reset_s_offset" in exp:
        assert reset_time // 1000 - now_ms // 1000 == exp["reset_s_offset"], \
            f"{where}: reset_time {reset_time} now {now_ms} want offset {exp['reset_s_offset']}"
    if "error" in exp and exp["error"] == "":
        assert err == 0, f"{where}: unexpected item error {err}"


def run_functional(make_backend):
    """One fresh backend per scenario (the reference uses distinct keys per test; a fresh table is
    equivalent and keeps scenarios independent)."""
    n_checked = 0
    for sc in load("functional_vectors.json")["scenarios"]:
        be = make_backend()
        now = sc["start_ms"]
        for si, step in enumerate(sc.get("steps", [])):
            where = f"{sc['name']} step {si} ({sc['source']})"
            verr = validate(step["req"])
            if verr:
                assert verr == step["expect"]["error"], where
                n_checked += 1
                continue
            res = be.eval(batch_of([step["req"]], now))
            check_expect(step["expect"], res.rows()[0], now, where)
            n_checked += 1
            now += step["advance_ms"]
        for si, step in enumerate(sc.get("batch_steps", [])):
            res = be.eval(batch_of(step["reqs"], now))
            for j, exp in enumerate(step["expect"]):
                check_expect(exp, res.rows()[j], now, f"{sc['name']} batch step {si}[{j}]")
                n_checked += 1
            now += step["advance_ms"]
        if hasattr(be, "close"):
            be.close()
    return n_checked


def run_store(make_backend):
    n = 0
    for case in load("store_vectors.json")["cases"]:
        be = make_backend()
        now = case["now_ms"]
        for p in case["preload"]:
            it = support.make_item(p["key"], p["algorithm"], limit=p.get("limit", 0), duration=p.get("duration", 0),
                                   remaining=p.get("remaining", 0), remaining_f=p.get("remaining_f", 0.0),
                                   stamp=p.get("stamp", 0), burst=p.get("burst", 0), expire_at=p.get("expire_at", 0))
            be.add_item(it, now)
        res = be.eval(batch_of([case["req"]], now))
        check_expect(case["expect_resp"], res.rows()[0], now, case["name"])
        key = case["req"]["name"] + "_" + case["req"]["unique_key"]
        item = be.get_item(key, now)
        assert item is not None, case["name"]
        for k, v in case["expect_item"].items():
            if k == "expire_at_minus_stamp":
                assert item["expire_at"] - item["stamp"] == v, f"{case['name']}: {item}"
            else:
                assert item[k] == v, f"{case['name']}: item.{k} = {item[k]} want {v}"
        if "expect_size" in case:
            assert be.size() == case["expect_size"]
            allitems = be.each()
            assert len(allitems) == case["expect_size"]
            assert allitems[0]["key"] == key.encode()
        n += 1
        if hasattr(be, "close"):
            be.close()
    return n


def run_store_events(make_backend):
    """tests/golden/store_events_vectors.json (store_test.go TestStore): the backend's
    eval_store(batch, store) must issue exactly the Store calls the reference's mock expects, in order, and the
    OnChange item must satisfy the test's matchItem."""
    n = 0
    for case in load("store_events_vectors.json")["cases"]:
        be = make_backend()
        now = case["now_ms"]
        for si, step in enumerate(case["steps"]):
            where = f"{case['name']} step {si} ({case['source']})"
            key = step["req"]["name"] + "_" + step["req"]["unique_key"]
            store = support.MockStore({key: step["get"]} if step["get"] else {})
            res = be.eval_store(batch_of([step["req"]], now), store)
            check_expect(step["expect_resp"], res.rows()[0], now, where)
            assert store.kinds() == step["calls"], f"{where}: {store.kinds()}"
            for c in store.calls:
                assert c[1] == 0 and c[2] == key, where
            item = [c for c in store.calls if c[0] == "on_change"][-1][3]
            exp = step["expect_item"]
            for f in ("algorithm", "limit", "duration", "stamp"):
                if f in exp:
                    assert item[f] == exp[f], f"{where}: item.{f} = {item[f]}"
            assert item["key"] == exp["key"], where
            if "expire_at_minus_stamp" in exp:
                assert item["expire_at"] - item["stamp"] == exp["expire_at_minus_stamp"], where
            n += 1
        if hasattr(be, "close"):
            be.close()
    return n


def run_cache_vectors(make_backend, evicting=True):
    """tests/golden/cache_vectors.json (lrucache_test.go TestLRUCache) on a backend with add_item / get_item / remove_item /
    size (+ counters() for the eviction cases).  evicting=False skips the cases that need the bounded LRU (the HBM table
    compacts instead of evicting)."""
    n = 0
    for case in load("cache_vectors.json")["cases"]:
        if case["evicting"] and not evicting:
            continue
        be = make_backend(case["cache_size"])
        now = 1_700_000_000_000
        for op in case["ops"]:
            where = f"{case['name']} {op} ({case['source']})"
            if op[0] == "add":
                remaining = op[4] if len(op) > 4 else 7
                existed = be.add_item(support.make_item(op[1], support.LEAKY if case["evicting"] else support.TOKEN, limit=10, duration=op[2],
                                                        remaining=remaining, remaining_f=float(remaining), stamp=now, burst=10,
                                                        expire_at=now + op[2]), now)
                assert bool(existed) == op[3], where
            elif op[0] == "get":
                it = be.get_item(op[1], now)
                assert (it is not None) == op[2], where
                if len(op) > 3:
                    assert it["remaining"] == op[3], where
            elif op[0] == "remove":
                be.remove_item(op[1])
            elif op[0] == "size":
                assert be.size() == op[1], where
            elif op[0] == "advance":
                now += op[1]
            elif op[0] == "unexpired_evictions":
                assert be.counters()[3] == op[1], where
            n += 1
        if hasattr(be, "close"):
            be.close()
    return n
