"""Golden functional scenarios replayed THROUGH THE WIRE FORMAT: request -> protobuf bytes (python runtime) ->
guber_wire_decode_requests -> evaluator (oracle on CPU, HIP engine on GPU) -> guber_wire_encode_responses ->
protobuf bytes parsed by the python runtime -> expectations of tests/golden/functional_vectors.json."""
import scenarios
from pb_schema import PB


def pb_request(reqs, peer=False):
    m = PB["GetPeerRateLimitsReq" if peer else "GetRateLimitsReq"]()
    for r in reqs:
        q = m.requests.add(name=r["name"], unique_key=r["unique_key"], hits=r["hits"], limit=r["limit"], duration=r["duration"],
                           algorithm=r["algorithm"], behavior=r["behavior"], burst=r.get("burst", 0))
        if r.get("created_at"):
            q.created_at = r["created_at"]
    return m.SerializeToString()


def rows_of(payload):
    m = PB["GetRateLimitsResp"]()
    m.ParseFromString(payload)
    return [(x.status, x.limit, x.remaining, x.reset_time, x.error) for x in m.responses]


def run_functional_wire(make_wire, make_evaluator):
    """make_evaluator() -> (evaluate(wire_batch), close)."""
    n_checked = 0
    for sc in scenarios.load("functional_vectors.json")["scenarios"]:
        wb = make_wire()
        evaluate, close = make_evaluator()
        now = sc["start_ms"]
        steps = [([s["req"]], [s["expect"]], s["advance_ms"]) for s in sc.get("steps", [])]
        steps += [(s["reqs"], s["expect"], s["advance_ms"]) for s in sc.get("batch_steps", [])]
        for si, (reqs, expects, adv) in enumerate(steps):
            wb.reset(now)
            first, count = wb.decode(pb_request(reqs), max_per_rpc=1000)
            assert (first, count) == (0, len(reqs))
            evaluate(wb)
            rows = rows_of(wb.encode(first, count))
            assert len(rows) == len(reqs)
            for j, (exp, row) in enumerate(zip(expects, rows)):
                where = f"{sc['name']} step {si}[{j}] ({sc['source']}) via wire"
                status, limit, remaining, reset_time, error = row
                if exp.get("error"):
                    assert error == exp["error"], f"{where}: {error!r}"
                    assert (status, limit, remaining, reset_time) == (0, 0, 0, 0), where
                else:
                    assert error == "", f"{where}: unexpected error {error!r}"
                    scenarios.check_expect(exp, (status, limit, remaining, reset_time, 0), now, where)
                n_checked += 1
            now += adv
        close()
        wb.close()
    return n_checked
