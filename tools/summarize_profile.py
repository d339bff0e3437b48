#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/prof_<algo>/, gpurun_out/pmc_<CTR>/) into the committed
profiles/: per-kernel duration statistics (a) over the last N_ISO dispatches = bench.py's profiling + latency
legs, one batch in flight, the condition roofline.kernel_avg_us is measured under, and (b) over the N_TIMED
dispatches before them = the timed region, where the logical shards' kernels overlap on the GPU; and HBM
traffic per launch from the PMC passes.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes
(MI355X_MICROARCH.md, HBM section), so the corrected read volume is 2 x FETCH_SIZE (upper bound for our
mix of 16-byte and 128-byte random reads; the raw figure is kept next to it).
usage: summarize_profile.py <tag> <n_iso> <n_pmc> [n_timed]"""
import csv, json, os, sys, collections, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n_stats, n_pmc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n_timed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
out = {"tag": tag, "kernels": {}, "traffic": {}}
lines = [f"# rocprofv3 summary {tag} (bench.py, 10M keys, Zipf-1.1, batch 65536, 1xMI355X)", ""]
for algo in ("token", "leaky"):
    path = os.path.join(ROOT, "gpurun_out", f"prof_{algo}", f"{algo}_kernel_trace.csv")
    if not os.path.exists(path):
        continue
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            per[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    lines += [f"## {algo}: kernel durations over the last {n_stats} dispatches (one batch in flight)", "",
              "| kernel | launches | avg us | min us | p50 us | max us |", "|---|---|---|---|---|---|"]
    out["kernels"][algo] = {}
    for name, d in per.items():
        d = d[-n_stats:]
        st = dict(launches=len(d), avg_us=sum(d) / len(d) / 1e3, min_us=min(d) / 1e3, p50_us=statistics.median(d) / 1e3, max_us=max(d) / 1e3)
        out["kernels"][algo][name.replace("guber::", "")] = st
        lines.append(f"| {name} | {st['launches']} | {st['avg_us']:.2f} | {st['min_us']:.2f} | {st['p50_us']:.2f} | {st['max_us']:.2f} |")
    lines.append("")
    if n_timed:
        lines += [f"## {algo}: the {n_timed} dispatches before those (timed region, logical shards overlapping)", "",
                  "| kernel | launches | avg us | min us | p50 us | max us |", "|---|---|---|---|---|---|"]
        out["kernels"][algo + "_timed_region"] = {}
        for name, d in per.items():
            d = d[-(n_stats + n_timed):-n_stats]
            if not d:
                continue
            st = dict(launches=len(d), avg_us=sum(d) / len(d) / 1e3, min_us=min(d) / 1e3, p50_us=statistics.median(d) / 1e3, max_us=max(d) / 1e3)
            out["kernels"][algo + "_timed_region"][name.replace("guber::", "")] = st
            lines.append(f"| {name} | {st['launches']} | {st['avg_us']:.2f} | {st['min_us']:.2f} | {st['p50_us']:.2f} | {st['max_us']:.2f} |")
        lines.append("")
pm = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(ROOT, "gpurun_out", f"pmc_{ctr}", "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name and row["Counter_Name"] == ctr:
            per[name.replace("guber::", "")].append(float(row["Counter_Value"]))
    pm[ctr] = {k: sum(v[-n_pmc:]) / len(v[-n_pmc:]) for k, v in per.items()}
if pm:
    lines += [f"## token: HBM traffic per launch (PMC, separate passes, last {n_pmc} dispatches)", "",
              "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | raw bytes | corrected bytes (2xFETCH+WRITE) | algorithmic bytes |", "|---|---|---|---|---|---|"]
    alg = {"k_front": 76 * 65536, "k_eval2": 73 * 65536}
    tr, tr_raw = {}, {}
    for k in sorted(set(pm.get("FETCH_SIZE", {})) | set(pm.get("WRITE_SIZE", {}))):
        f, w = pm.get("FETCH_SIZE", {}).get(k, 0.0), pm.get("WRITE_SIZE", {}).get(k, 0.0)
        raw, cor = (f + w) * 1024, (2 * f + w) * 1024
        tr[k] = int(cor)
        tr_raw[k] = int(raw)
        lines.append(f"| {k} | {f:.1f} | {w:.1f} | {raw:.0f} | {cor:.0f} | {alg.get(k, '')} |")
        out["traffic"][k] = dict(fetch_kib=f, write_kib=w, raw_bytes=raw, corrected_bytes=cor)
    json.dump({"token": tr, "token_raw": tr_raw,
               "note": "bytes per launch: token = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, an upper bound for 64-byte requests), "
                       "token_raw = (FETCH_SIZE + WRITE_SIZE)*1024"},
              open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
