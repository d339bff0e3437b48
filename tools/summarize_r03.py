#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_profile_r03.sh (gpurun_out/<tag>/) into <tag>_rocprof_summary.{md,json}
and roofline_traffic.json (copied to profiles/ by hand).  Kernel durations: End - Start timestamps of the kernel trace.
`dominant_kernel` = the kernel with the largest summed duration among the batch kernels; its roofline line uses the formula
of bench.py: bytes per request x average requests per launch / average duration / 8000 GB/s.  Only the dispatches of the
non-replayed stream count (for one table: the last N launches of k_front / k_eval2 — the earlier ones are the residency pass).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM
section): raw = FETCH + WRITE, corrected = 2 x FETCH + WRITE (an upper bound for this mix of 16..64-byte random reads).
usage: summarize_r03.py <tag> [pmc_batches]"""
import collections, csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 512
base = os.path.join(ROOT, "gpurun_out", tag)
ALG = {"k_front": 76, "k_eval2": 73, "k_front_multi": 76, "k_eval2_multi": 73}
out = {"tag": tag, "kernels": {}, "counters": {}}
lines = [f"# rocprofv3 summary {tag} (bench.py, 10M keys, one NON-REPLAYED Zipf-1.1 stream, batch 65536, 1xMI355X)", ""]


def find(d, pat):
    hits = glob.glob(os.path.join(base, d, "**", pat), recursive=True)
    return hits[0] if hits else None


def bench_line(log):
    try:
        for ln in open(os.path.join(base, log)):
            if ln.startswith("{") and '"metric"' in ln:
                return json.loads(ln)
    except Exception:
        pass
    return None


for key, d, lastn, title in (("fused", "trace_fused", None, "the default bench command (12 logical shards, one dispatcher, 3 streams): up to four tables per launch"),
                             ("shards_1", "trace_s1", 1024 + 16 + 128 + 128, "one table (--shards 1, 1024 timed batches): one batch in flight")):
    path = find(d, "*kernel_trace.csv")
    if not path:
        continue
    per, grid, cols = collections.defaultdict(list), collections.defaultdict(list), {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            k = name.replace("guber::", "")
            per[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            grid[k].append(int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0))
            cols[k] = {c: row.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size")}
    bl = bench_line(d + ".log")
    lines += [f"## {title}", ""]
    if bl:
        lines += [f"bench line of this traced run: value {bl['value'] / 1e9:.3f} G decisions/s, ms_per_step {bl['ms_per_step']}, "
                  f"live roofline (HIP events): kernel {bl['roofline']['kernel']} avg {bl['roofline']['kernel_avg_us'].get(bl['roofline']['kernel'])} us, frac {bl['roofline']['frac']}", ""]
    lines += ["| kernel | launches | avg us | min us | p50 us | max us | VGPR | SGPR | LDS B | scratch B | avg requests per launch | algorithmic GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    ks = {}
    for name, dd in per.items():
        g = grid[name]
        if lastn and name in ("k_front", "k_eval2"):
            dd, g = dd[-lastn:], g[-lastn:]
        st = dict(launches=len(dd), total_us=sum(dd) / 1e3, avg_us=sum(dd) / len(dd) / 1e3, min_us=min(dd) / 1e3, p50_us=statistics.median(dd) / 1e3, max_us=max(dd) / 1e3,
                  avg_requests_per_launch=sum(g) / len(g), **cols[name])
        if name in ALG:
            st["algorithmic_GBps"] = ALG[name] * st["avg_requests_per_launch"] / (st["avg_us"] * 1e3)
            st["frac"] = st["algorithmic_GBps"] / 8000.0
        ks[name] = st
        c = cols[name]
        lines.append(f"| {name} | {st['launches']} | {st['avg_us']:.2f} | {st['min_us']:.2f} | {st['p50_us']:.2f} | {st['max_us']:.2f} | {c['VGPR_Count']} | {c['SGPR_Count']} | {c['LDS_Block_Size']} | {c['Scratch_Size']} | "
                     f"{st['avg_requests_per_launch']:.0f} | " + (f"{st['algorithmic_GBps']:.0f} | {st['frac']:.4f}" if "frac" in st else " | ") + " |")
    lines.append("")
    out["kernels"][key] = ks
    cand = {k: v for k, v in ks.items() if k in ALG and (key != "fused" or k.endswith("_multi"))}
    if cand and key == "fused":
        dom = max(cand, key=lambda k: cand[k]["total_us"])
        st = cand[dom]
        out["dominant_kernel"] = {"name": dom, "avg_us": round(st["avg_us"], 3), "requests_per_launch": round(st["avg_requests_per_launch"], 1), "bytes_per_request": ALG[dom],
                                  "achieved_GBps": round(st["algorithmic_GBps"], 2), "frac": round(st["frac"], 6), "launches": st["launches"]}
        try:
            out["command"] = "rocprofv3 --kernel-trace --stats -- " + open(os.path.join(base, "trace_fused.cmd")).read().strip()
        except Exception:
            pass
        if bl:
            out["bench_line_of_traced_run"] = {"value": bl["value"], "ms_per_step": bl["ms_per_step"], "roofline": bl["roofline"]}
        lines += [f"**dominant kernel: {dom}: {ALG[dom]} B x {st['avg_requests_per_launch']:.0f} requests / {st['avg_us']:.2f} us = {st['algorithmic_GBps']:.0f} GB/s = {st['frac']:.4f} of 8 TB/s**", ""]

pm = {}
for d in sorted(glob.glob(os.path.join(base, "pmc_s*"))):
    if not os.path.isdir(d):
        continue
    s = os.path.basename(d).split("_")[1]
    path = find(os.path.basename(d), "*counter_collection.csv")
    if not path:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    grid = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            per[name.replace("guber::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            if k in ("k_front", "k_eval2") and s == "s1":
                v = v[-NB:]
            pm.setdefault(s, {}).setdefault(k, {})[c] = (sum(v) / len(v), len(v))
for s in sorted(pm):
    ctrs = sorted({c for k in pm[s] for c in pm[s][k]})
    lines += [f"## hardware counters per launch, {'one table' if s == 's1' else '12 shards fused'} (separate PMC passes, {NB} distinct batches, non-replayed)", "",
              "| kernel | launches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
    for k in sorted(pm[s]):
        if k in ALG:
            n = max(v[1] for v in pm[s][k].values())
            lines.append(f"| {k} | {n} | " + " | ".join(f"{pm[s][k][c][0]:.1f}" if c in pm[s][k] else "" for c in ctrs) + " |")
    lines.append("")
    out["counters"][s] = {k: {c: v[0] for c, v in cs.items()} for k, cs in pm[s].items()}
if "s1" in pm:
    tr, tr_raw = {}, {}
    lines += ["## HBM-side traffic per 65536-request launch (one table)", "", "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | raw bytes | corrected bytes (2xFETCH+WRITE) | algorithmic bytes | raw / algorithmic | corrected / algorithmic |", "|---|---|---|---|---|---|---|---|"]
    for k in ("k_front", "k_eval2"):
        c = pm["s1"].get(k, {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = c["FETCH_SIZE"][0], c["WRITE_SIZE"][0]
            raw, cor = (f + w) * 1024, (2 * f + w) * 1024
            tr[k], tr_raw[k] = int(cor), int(raw)
            lines.append(f"| {k} | {f:.1f} | {w:.1f} | {raw:.0f} | {cor:.0f} | {ALG[k] * 65536} | {raw / (ALG[k] * 65536):.2f} | {cor / (ALG[k] * 65536):.2f} |")
    lines.append("")
    if tr:
        json.dump({"token": tr, "token_raw": tr_raw, "source": f"profiles/{tag}_rocprof_summary.md",
                   "note": "PMC bytes per 65536-request launch on the non-replayed stream, one table: token = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, an "
                           "upper bound for 64-byte requests), token_raw = (FETCH_SIZE + WRITE_SIZE)*1024; a fused launch is scaled by its requests"},
                  open(os.path.join(base, "roofline_traffic.json"), "w"), indent=1)
open(os.path.join(base, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(base, f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
