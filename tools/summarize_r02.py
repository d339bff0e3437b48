#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_profile_r02.sh (gpurun_out/<tag>/) into profiles/<tag>_rocprof_summary.{md,json}
and profiles/roofline_traffic.json.  Per kernel the LAST `N` dispatches are used (the timed region and the latency leg of
bench.py; the earlier ones are the residency pass, which inserts every key).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section): raw = FETCH + WRITE, corrected =
2 x FETCH + WRITE (an upper bound for this mix of 16..64-byte random reads).
usage: summarize_r02.py <tag> [N]"""
import collections, csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
base = os.path.join(ROOT, "gpurun_out", tag)
out = {"tag": tag, "last_n_dispatches": N, "kernels": {}, "counters": {}}
lines = [f"# rocprofv3 summary {tag} (bench.py, 10M keys, one routed Zipf-1.1 stream, batch 65536, 1xMI355X; last {N} dispatches per kernel)", ""]


def find(d, pat):
    hits = glob.glob(os.path.join(base, d, "**", pat), recursive=True)
    return hits[0] if hits else None


for s in (1, 4, "fused"):
    path = find(f"trace_s{s}" if s != "fused" else "trace_fused", "*kernel_trace.csv")
    if not path:
        continue
    per = collections.defaultdict(list)
    grid = collections.defaultdict(list)
    cols = {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            per[name.replace("guber::", "")].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            grid[name.replace("guber::", "")].append(int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0))
            cols[name.replace("guber::", "")] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    title = {1: "1 logical shard(s): kernel durations (one batch in flight)", 4: "4 logical shard(s), one thread + stream each: kernel durations (shards overlapping)",
             "fused": "12 logical shards, one dispatcher, 3 streams (the bench default): up to four tables per launch (k_front_multi / k_eval2_multi)"}[s]
    lines += [f"## {title}", "",
              "| kernel | launches | avg us | min us | p50 us | max us | VGPR | SGPR | LDS B | scratch B | avg requests per launch | algorithmic GB/s |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    out["kernels"][f"shards_{s}"] = {}
    alg_b = {"k_front": 76, "k_eval2": 73, "k_front_multi": 76, "k_eval2_multi": 73}
    for name, d in per.items():
        d = d[-N:]
        g = grid[name][-N:]
        st = dict(launches=len(d), avg_us=sum(d) / len(d) / 1e3, min_us=min(d) / 1e3, p50_us=statistics.median(d) / 1e3, max_us=max(d) / 1e3,
                  avg_requests_per_launch=sum(g) / len(g), **cols[name])
        if name in alg_b:
            st["algorithmic_GBps"] = alg_b[name] * st["avg_requests_per_launch"] / (st["avg_us"] * 1e3)
        out["kernels"][f"shards_{s}"][name] = st
        c = cols[name]
        lines.append(f"| {name} | {st['launches']} | {st['avg_us']:.2f} | {st['min_us']:.2f} | {st['p50_us']:.2f} | {st['max_us']:.2f} | {c['VGPR_Count']} | {c['SGPR_Count']} | {c['LDS_Block_Size']} | {c['Scratch_Size']} | "
                     f"{st['avg_requests_per_launch']:.0f} | " + (f"{st['algorithmic_GBps']:.0f}" if "algorithmic_GBps" in st else "") + " |")
    lines.append("")
pm = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(base, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    path = find(os.path.basename(d), "*counter_collection.csv")
    if not path:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            per[name.replace("guber::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            v = v[-N:]
            pm[k][c] = sum(v) / len(v)
if pm:
    ctrs = sorted({c for k in pm for c in pm[k]})
    lines += ["## hardware counters per launch (separate PMC passes, one table, one batch in flight)", "",
              "| kernel | " + " | ".join(ctrs) + " |", "|---|" + "---|" * len(ctrs)]
    for k in sorted(pm):
        if k in ("k_front", "k_eval2"):
            lines.append(f"| {k} | " + " | ".join(f"{pm[k].get(c, float('nan')):.1f}" for c in ctrs) + " |")
    lines.append("")
    out["counters"] = {k: dict(v) for k, v in pm.items()}
    alg = {"k_front": 76 * 65536, "k_eval2": 73 * 65536}
    tr, tr_raw = {}, {}
    lines += ["## HBM-side traffic per launch", "", "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | raw bytes | corrected bytes (2xFETCH+WRITE) | algorithmic bytes | raw / algorithmic |", "|---|---|---|---|---|---|---|"]
    for k in ("k_front", "k_eval2"):
        if k in pm and "FETCH_SIZE" in pm[k] and "WRITE_SIZE" in pm[k]:
            f, w = pm[k]["FETCH_SIZE"], pm[k]["WRITE_SIZE"]
            raw, cor = (f + w) * 1024, (2 * f + w) * 1024
            tr[k], tr_raw[k] = int(cor), int(raw)
            lines.append(f"| {k} | {f:.1f} | {w:.1f} | {raw:.0f} | {cor:.0f} | {alg[k]} | {raw / alg[k]:.2f} |")
    if tr:
        json.dump({"token": tr, "token_raw": tr_raw, "source": f"profiles/{tag}_rocprof_summary.md",
                   "note": "bytes per launch: token = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, an upper bound for 64-byte requests), "
                           "token_raw = (FETCH_SIZE + WRITE_SIZE)*1024"},
                  open(os.path.join(ROOT, "gpurun_out", tag, "roofline_traffic.json"), "w"), indent=1)
open(os.path.join(base, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(base, f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
