"""bench.py's host-side arithmetic (no GPU): what `roofline.traffic` is made of."""
import importlib.util
import json
import os

from support import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("guber_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_roofline_traffic_is_the_mix_of_pipelines_the_profile_segment_ran():
    """a batch goes through ONE pipeline: the step's PMC traffic is every kernel's bytes per batch weighted by the batches it carried,
    over the batches that entered a pipeline — not the sum over every kernel that appeared (which counted a batch twice)"""
    b = _bench()
    tj = {"token": {"k_front": 20, "k_eval2": 10, "k_part": 8, "k_own": 9, "k_eval3": 7}, "token_fused": {"k_part": 6, "k_own": 7, "k_eval3": 5}}
    B = 65536
    # one table, claims only
    assert b.pipeline_traffic(tj, "token", ["k_front", "k_eval2"], {"k_front": 4, "k_eval2": 4}, {}, B) == 30
    # fused owner-partitioned launches of 4 batches each + a few single claims launches: (2 x 30 + 12 x 18) / 14
    launches = {"k_front": 2, "k_eval2": 2, "k_part_multi": 3, "k_own_multi": 3, "k_eval3_multi": 3}
    per_launch = {k: 4 * B for k in ("k_part_multi", "k_own_multi", "k_eval3_multi")}
    assert b.pipeline_traffic(tj, "token", list(launches), launches, per_launch, B) == int((2 * 30 + 12 * 18) / 14)
    # GUBER_FUSE_EP: k_part_multi once, then k_evalpart_multi (one group's k_eval3 + the next group's k_part) per round, k_eval3_multi last
    launches = {"k_part_multi": 1, "k_own_multi": 3, "k_evalpart_multi": 2, "k_eval3_multi": 1}
    assert b.pipeline_traffic(tj, "token", list(launches), launches, {k: 4 * B for k in launches}, B) == 18
    # a kernel the file does not know: no figure rather than a wrong one
    assert b.pipeline_traffic(tj, "token", ["k_front", "k_eval2", "k_other"], {"k_front": 1, "k_eval2": 1, "k_other": 1}, {}, B) is None
    # the committed file carries every kernel of both pipelines
    real = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
    assert set(real["token"]) >= {"k_front", "k_eval2", "k_part", "k_own", "k_eval3"} and set(real["token_fused"]) >= {"k_part", "k_own", "k_eval3"}
    got = b.pipeline_traffic(real, "token", list(launches), launches, per_launch, B)
    assert 149 * B < got < 4 * 149 * B
