"""Randomised soak of the batch pipelines' kernel source on the CPU (tests/hostsim/libdevsim*.so: make -C tests/hostsim devsim_lib)
against the oracle: per seed a table size, a key population, a batch size, an owner mode (following / 128 / 256), a record / message
form and k_eval3 whole or split, adversarial streams through the owner-partitioned pipeline, every answer and the counters compared.
With `ep` as a fourth argument: GUBER_FUSE_EP streams instead (ds_eval_stream_ep: 1 .. 4 tables, k_part_multi / k_own_multi /
k_evalpart_multi / k_eval3_multi, the fused launch's workgroups ascending, descending or shuffled, owner mode following or pinned).
Test infrastructure, not part of the suite:   python tools/soak_devsim.py <first seed> <last seed> <seconds> [ep]"""
import sys, os, time, ctypes as C, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import streams
from support import GuberBatch, GuberResult, HostBatch, HostResult, Oracle, assert_results_equal, gregorian

HS = os.path.join(ROOT, "tests", "hostsim")
def load(name):
    L = C.CDLL(os.path.join(HS, name))
    L.ds_create_bounded.restype = C.c_void_p; L.ds_create_bounded.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint64]
    L.ds_destroy.argtypes = [C.c_void_p]
    L.ds_eval.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_int, C.c_int]
    L.ds_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.ds_pin_owner_bits.argtypes = [C.c_void_p, C.c_uint32]
    L.ds_fuse_ep.argtypes = [C.c_void_p, C.c_int]
    L.ds_block_order.argtypes = [C.c_uint32]
    L.ds_chaos.argtypes = [C.c_uint32]
    L.ds_eval_stream_ep.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_uint32]
    return L

def one_ep(seed):
    """a GUBER_FUSE_EP stream: nh tables, nr rounds, in pieces of 1 .. 4 rounds per ds_eval_stream_ep call (a call's last k_eval3 goes on
    its own, the next call starts with a k_part of its own: what guber_eval_batches_routed_dev does at its ends)"""
    rng = np.random.default_rng(seed)
    L = load("libdevsim.so")
    nh = int(rng.integers(1, 5)); nr = int(rng.choice([3, 6, 10]))
    slots = int(rng.choice([4096, 16384, 1 << 17, 1 << 20]))
    n_keys = min(int(rng.choice([50, 97, 400, 1000, 3000, 12000])), slots // 8)
    bs = int(rng.choice([300, 1500, 5000, 12000]))
    obits = int(rng.choice([0, 0, 7, 8])); order = int(rng.integers(0, 3)); chaos = int(rng.random() < 0.3)
    hs = [L.ds_create_bounded(slots, 16384, 0, 0) for _ in range(nh)]
    for h in hs:
        if obits: L.ds_pin_owner_bits(h, obits)
        L.ds_fuse_ep(h, 1)
    orcs = [Oracle(cache_size=1 << 20) for _ in range(nh)]
    cfg = f"seed {seed} EP tables {nh} rounds {nr} slots {slots} keys {n_keys} batch {bs} owners {obits} order {order} chaos {chaos}"
    try:
        gens = [streams.adversarial_batches(seed * 7 + j, nr, bs, n_keys=n_keys, greg_fn=gregorian) for j in range(nh)]
        rounds = [[next(g) for g in gens] for _ in range(nr)]
        res = [[HostResult(b.n) for b in rnd] for rnd in rounds]
        L.ds_block_order(order); L.ds_chaos(chaos)
        r0 = 0
        while r0 < nr:
            k = min(nr - r0, int(rng.integers(1, 5)))
            bsa, rsa = (GuberBatch * (nh * k))(), (GuberResult * (nh * k))()
            for r in range(k):
                for j in range(nh):
                    bsa[r * nh + j], rsa[r * nh + j] = rounds[r0 + r][j].c, res[r0 + r][j].c
            rc = L.ds_eval_stream_ep((C.c_void_p * nh)(*hs), nh, bsa, rsa, k)
            assert rc == 0, f"rc {rc}"
            r0 += k
        for j in range(nh):
            for r in range(nr):
                assert_results_equal(res[r][j], orcs[j].eval(rounds[r][j]), f"{cfg} table {j} round {r}")
            out = (C.c_longlong * 6)(); L.ds_counters(hs[j], out)
            co = orcs[j].counters()
            assert (out[0], out[1], out[2]) == (co[0], co[1], co[2]) and out[3] == orcs[j].size() and out[4] == 0, f"{cfg}: counters {tuple(out)} vs {co}"
        return "ok " + cfg
    except Exception as e:   # noqa
        return "FAIL " + cfg + " :: " + str(e)[:400]
    finally:
        L.ds_block_order(0); L.ds_chaos(0)
        for h in hs: L.ds_destroy(h)

def one(seed):
    rng = np.random.default_rng(seed)
    libname = "libdevsim.so"
    L = load(libname)
    slots = int(rng.choice([4096, 16384, 1 << 17, 1 << 20]))
    n_keys = int(rng.choice([50, 97, 400, 1000, 3000, 12000]))
    n_keys = min(n_keys, slots // 8)
    bs = int(rng.choice([300, 1500, 5000, 12000]))
    obits = int(rng.choice([0, 0, 7, 8])); split = 0
    nb = int(rng.choice([4, 8, 14]))
    h = L.ds_create_bounded(slots, 16384, 0, 0)
    if obits: L.ds_pin_owner_bits(h, obits)
    orc = Oracle(cache_size=1 << 20)
    cfg = f"seed {seed} lib {libname} slots {slots} keys {n_keys} batch {bs} x{nb} owners {obits} split {split}"
    try:
        for k, b in enumerate(streams.adversarial_batches(seed, nb, bs, n_keys=n_keys, greg_fn=gregorian)):
            res = HostResult(b.n)
            rc = L.ds_eval(h, C.byref(b.c), C.byref(res.c), 1, 0)
            assert rc == 0, f"rc {rc}"
            assert_results_equal(res, orc.eval(b), f"{cfg} batch {k}")
        out = (C.c_longlong * 6)(); L.ds_counters(h, out)
        co = orc.counters()
        assert (out[0], out[1], out[2]) == (co[0], co[1], co[2]) and out[3] == orc.size(), f"{cfg}: counters {tuple(out)} vs {co} size {orc.size()}"
        assert out[4] == 0, f"{cfg}: retries {out[4]}"
        return "ok " + cfg
    except Exception as e:   # noqa
        return "FAIL " + cfg + " :: " + str(e)[:400]
    finally:
        L.ds_destroy(h)

if __name__ == "__main__":
    lo, hi, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    t0 = time.time()
    for seed in range(lo, hi):
        if time.time() - t0 > budget: break
        print((one_ep if len(sys.argv) > 4 and sys.argv[4] == "ep" else one)(seed), flush=True)
