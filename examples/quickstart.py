#!/usr/bin/env python3
"""Smallest end-to-end use of the engine through its Python binding (needs an MI355X; the binding is ctypes over the C ABI
of include/guber_gpu.h, so the same calls map one-to-one to cgo / JNI / N-API).

  1. an engine = one HBM table on one GPU
  2. a batch of RateLimitReq-shaped rows (structure of arrays), evaluated in request order like gubernator's
     V1.GetRateLimits -> WorkerPool.GetRateLimit
  3. the same through the protobuf wire format (what a daemon would hand over from the socket)
  4. the payload stage: what a gRPC handler thread calls with the raw bytes of its RPC — decode, routing to the GPU's tables, evaluation and the
     answers' order all on the device (guber_wire_pool_*)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gubernator_amd as ga                                    # noqa: E402
from gubernator_amd import wire                                 # noqa: E402
from gubernator_amd.abi import HostBatch                        # noqa: E402

now_ms = 1_700_000_000_000
engine = ga.Engine(cache_size=100_000, max_batch=4096)

# three hits on one token bucket (limit 2 per 9 s) and one leaky-bucket request, in one batch
batch = HostBatch(keys=[b"requests_per_sec_account:1234"] * 3 + [b"mails_per_hour_account:1234"],
                  hits=1, limit=[2, 2, 2, 100], duration=[9_000, 9_000, 9_000, 3_600_000], now_ms=now_ms,
                  algorithm=[0, 0, 0, 1])
res = engine.eval(batch)
for key, (status, limit, remaining, reset_time, err) in zip(batch_keys := ["req/s"] * 3 + ["mail/h"], res.rows()):
    print(f"{key:7s} status={'OVER_LIMIT' if status else 'UNDER_LIMIT'} remaining={remaining} reset_in={reset_time - now_ms} ms")
print("resident buckets:", engine.size(), "| counters (over, hits, misses, evictions, size):", res.counters())

# the same decisions from serialized GetRateLimitsReq bytes (two fields per item are enough: see gubernator.proto:137-182)
def req(name, unique_key, hits, limit, duration):
    def varint(v):
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7f) | 0x80); v >>= 7
        out.append(v)
        return bytes(out)
    body = b"\x0a" + varint(len(name)) + name + b"\x12" + varint(len(unique_key)) + unique_key
    body += b"\x18" + varint(hits) + b"\x20" + varint(limit) + b"\x28" + varint(duration)
    return b"\x0a" + varint(len(body)) + body

payload = req(b"requests_per_sec", b"account:9", 1, 2, 9_000) * 3
wb = wire.WireBatch(max_items=4096, max_key_bytes=1 << 16, pinned=True)
wb.reset(now_ms)
first, count = wb.decode(payload, max_per_rpc=1000)
wb.eval(engine)
print("GetRateLimitsResp bytes:", wb.encode(first, count).hex())
engine.close()

# the payload stage over four tables of one GPU: serialized GetRateLimitsReq in, serialized GetRateLimitsResp out; thread-safe, one call per RPC
place = ga.Placement(4)                                        # key -> table: the reference's worker rule (workers.go:180-184), hot keys isolated online
first_table = ga.Engine(cache_size=100_000, max_batch=8192)
tables = [first_table] + [ga.Engine(cache_size=100_000, max_batch=8192, stream=first_table.stream_handle()) for _ in range(3)]
pool = wire.WirePool(tables, place)                            # defaults: twelve stages of 49 152 items, BatchWait 500 us, 1000 requests per RPC
payload = req(b"requests_per_sec", b"account:7", 1, 2, 9_000) * 3 + req(b"mails_per_hour", b"account:7", 1, 100, 3_600_000)
print("payload stage:", pool.get_rate_limits(payload).hex())
print("stages so far:", pool.stats()["stages"])
pool.close()
for t in reversed(tables):
    t.close()
place.close()
