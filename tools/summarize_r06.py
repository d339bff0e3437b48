#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_profile_r06.sh (gpurun_out/<tag>/) into <tag>_rocprof_summary.{md,json}, roofline_traffic.json and
roofline_issue.json (copied to profiles/ by hand).
Kernel durations: End - Start timestamps of the kernel traces; a kernel's algorithmic GB/s = bytes per request x average requests
per launch / average duration; the front's kernels (k_fr_*) move coordination bytes only: no algorithmic figure, their time is in the step.
PMC passes (routed arrangement, GUBER_FUSE_EP=0, <pmc_batches> + 16 warm-up batches of 65 536, all of them counted): per kernel the launches and the
counter's average per launch; per 65 536-request batch = sum over the launches / batches.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section): raw = FETCH + WRITE, corrected = 2 x FETCH + WRITE (an upper bound for this mix
of 16..64-byte random reads).  Issue time per batch = SQ_ACTIVE_INST_ANY (quad-cycles a SIMD spent issuing, summed over the chip) x 4 / (1024 SIMDs x 2.4 GHz).
Exits 1 when the traced run's line and 149 B x 65536 / ms_per_step / 8 TB/s disagree by more than 5 %.
usage: summarize_r06.py <tag> [pmc_batches]"""
import collections, csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
WARM = 16
base = os.path.join(ROOT, "gpurun_out", tag)
ALG = {"k_front": 76, "k_eval2": 73, "k_front_multi": 76, "k_eval2_multi": 73,
       "k_part": 20, "k_own": 56, "k_eval3": 73, "k_part_multi": 20, "k_own_multi": 56, "k_eval3_multi": 73, "k_evalpart_multi": 93}
PIPE = ("k_fr_count", "k_fr_scan", "k_fr_scatter", "k_part_multi", "k_own_multi", "k_eval3_multi", "k_evalpart_multi", "k_fr_out")
violations, notes = [], []
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


TITLES = {"routed": "the default bench command — the ROUTED headline: one raw stream, generations of 16 x 65 536 requests through guber_front_eval_dev (k_fr_count / k_fr_scan / k_fr_scatter on the "
                    "routing stream, the shares of 12 tables through k_evalpart_multi + k_own_multi on 3 streams, k_fr_out on those streams in turn)",
          "presplit": "--headline presplit: per-shard batches split outside the clock, answers in the shards' order (rounds 2-5's headline)",
          "shards_1": "--headline presplit --shards 1: one table, the two-launch pipeline with claims, one batch in flight"}
for key in ("routed", "presplit", "shards_1"):
    path = find("trace_" + key, "*kernel_trace.csv")
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
    bl = bench_line("trace_" + key + ".log")
    lines += [f"## {TITLES[key]}", ""]
    if bl:
        lines += [f"bench line of this traced run: value {bl['value'] / 1e9:.3f} G decisions/s, ms_per_step {bl['ms_per_step']}, roofline.frac (pipeline bytes / step) {bl['roofline']['frac']}, "
                  f"HIP events of the line: {bl['roofline']['kernel_avg_us']}", ""]
        pipe = 149 * 65536 / (bl["ms_per_step"] * 1e-3) / 1e9 / 8000
        if abs(pipe - bl["roofline"]["frac"]) / pipe > 0.05:
            violations.append(f"{key}: roofline.frac {bl['roofline']['frac']} is not 149 B x 65536 / ms_per_step / 8 TB/s = {pipe:.5f}")
        out.setdefault("bench_lines", {})[key] = {"value": bl["value"], "ms_per_step": bl["ms_per_step"], "roofline_frac": bl["roofline"]["frac"]}
    lines += ["| kernel | launches | avg us | min us | p50 us | max us | VGPR | SGPR | LDS B | scratch B | avg workgroups x threads per launch | algorithmic GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    ks = {}
    lastn = (1024 + 16 + 4096 + 1024) if key == "shards_1" else None
    for name, dd in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        g = grid[name]
        if lastn and name in ("k_front", "k_eval2"):
            dd, g = dd[-lastn:], g[-lastn:]
        st = dict(launches=len(dd), total_us=sum(dd) / 1e3, avg_us=sum(dd) / len(dd) / 1e3, min_us=min(dd) / 1e3, p50_us=statistics.median(dd) / 1e3, max_us=max(dd) / 1e3,
                  avg_grid_threads=sum(g) / len(g), **cols[name])
        if name in ALG and not name.startswith("k_own"):          # (k_own's grid is 256 owners x 256 threads whatever the batch: its requests come from its neighbours' grids)
            st["algorithmic_GBps"] = ALG[name] * st["avg_grid_threads"] / (st["avg_us"] * 1e3)
            st["frac"] = st["algorithmic_GBps"] / 8000.0
        ks[name] = st
        c = cols[name]
        lines.append(f"| {name} | {st['launches']} | {st['avg_us']:.2f} | {st['min_us']:.2f} | {st['p50_us']:.2f} | {st['max_us']:.2f} | {c['VGPR_Count']} | {c['SGPR_Count']} | {c['LDS_Block_Size']} | {c['Scratch_Size']} | "
                     f"{st['avg_grid_threads']:.0f} | " + (f"{st['algorithmic_GBps']:.0f} | {st['frac']:.4f}" if "frac" in st else " | ") + " |")
    lines.append("")
    out["kernels"][key] = ks
    if key == "routed":
        cand = {k: v for k, v in ks.items() if k in PIPE}
        if cand:
            dom = max(cand, key=lambda k: cand[k]["total_us"])
            st = cand[dom]
            out["dominant_kernel"] = {"name": dom, "avg_us": round(st["avg_us"], 3), "launches": st["launches"], "max_us": round(st["max_us"], 3),
                                      "requests_per_launch": round(st["avg_grid_threads"], 1) if not dom.startswith("k_own") else None,
                                      "achieved_GBps": round(st.get("algorithmic_GBps", 0.0), 2), "frac": round(st.get("frac", 0.0), 6)}
            out["command"] = "rocprofv3 --kernel-trace --stats -- " + open(os.path.join(base, "trace_routed.cmd")).read().strip()
            lines += [f"**the kernel the GPU spends most time in: {dom}: {st['launches']} launches, avg {st['avg_us']:.2f} us, max {st['max_us']:.2f} us**", ""]

# ---- PMC passes over the routed arrangement ----
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(base, "pmc_[0-9]*"))):
    if not os.path.isdir(d):
        continue
    path = find(os.path.basename(d), "*counter_collection.csv")
    if not path:
        continue
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            pm[name.replace("guber::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
batches = NB + WARM                                             # every batch of the counted runs went through the routed pipeline (profile / latency segments off)
if pm:
    ctrs = sorted({c for k in pm for c in pm[k]})
    lines += [f"## hardware counters, routed arrangement (separate PMC passes, {batches} batches of 65 536 = {batches // 16} generations each, GUBER_FUSE_EP=0 on the laboratory build: per launch averages)", "",
              "| kernel | launches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
    for k in PIPE:
        if k in pm:
            n = max(len(v) for v in pm[k].values())
            lines.append(f"| {k} | {n} | " + " | ".join(f"{sum(pm[k][c]) / len(pm[k][c]):.1f}" if c in pm[k] else "" for c in ctrs) + " |")
    lines.append("")
    out["counters"]["routed"] = {k: {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in pm.items() if k in PIPE}
    # HBM-side traffic per batch
    lines += ["## HBM-side traffic per 65 536-request batch, routed arrangement", "",
              "| kernel | FETCH_SIZE KiB per batch | WRITE_SIZE KiB per batch | raw bytes | corrected bytes (2xFETCH+WRITE) | algorithmic bytes |", "|---|---|---|---|---|---|"]
    tr, tot_raw, tot_cor = {}, 0.0, 0.0
    for k in PIPE:
        c = pm.get(k, {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = sum(c["FETCH_SIZE"]) / batches, sum(c["WRITE_SIZE"]) / batches
            raw, cor = (f + w) * 1024, (2 * f + w) * 1024
            tr[k] = {"raw": int(raw), "corrected": int(cor)}
            tot_raw += raw; tot_cor += cor
            lines.append(f"| {k} | {f:.1f} | {w:.1f} | {raw:.0f} | {cor:.0f} | {ALG.get(k, 0) * 65536 or 'coordination only'} |")
    alg = 149 * 65536
    if tr:
        lines += [f"| **pipeline** | | | **{tot_raw:.0f} = {tot_raw / alg:.2f} x algorithmic** | **{tot_cor:.0f} = {tot_cor / alg:.2f} x** | {alg} |", ""]
        front_raw = sum(v["raw"] for k, v in tr.items() if k.startswith("k_fr_"))
        lines += [f"of which the front's copies (k_fr_*): {front_raw:.0f} raw bytes per batch = {front_raw / 65536:.0f} B per request", ""]
        prev = {}
        try:                                                        # (the pre-split arrangement's figures of round 5 stay in the file: bench.py --headline presplit reads them)
            prev = {k: v for k, v in json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json"))).items() if k.startswith("token")}
        except Exception:
            pass
        json.dump({**prev, "routed": {"per_kernel": tr, "raw_per_batch": int(tot_raw), "corrected_per_batch": int(tot_cor)}, "source": f"profiles/{tag}_rocprof_summary.md",
                   "note": ("PMC bytes per 65536-request batch of the ROUTED arrangement, every kernel of the pipeline incl. the front's copies (k_fr_*): corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                            "(gfx950 correction, an upper bound for 64-byte requests), raw = (FETCH_SIZE + WRITE_SIZE)*1024; separate --pmc passes of the bench command with GUBER_FUSE_EP=0, committed")},
                  open(os.path.join(base, "roofline_traffic.json"), "w"), indent=1)
    # issue
    if any("SQ_ACTIVE_INST_ANY" in pm[k] for k in pm):
        issue = {"arrangement": "routed", "clock_GHz": 2.4, "simds": 1024, "kernels": {}, "source": f"profiles/{tag}_rocprof_summary.md",
                 "note": ("SQ counter passes (rocprofv3 --pmc, one group per pass) over the ROUTED arrangement with GUBER_FUSE_EP=0: per 65536-request batch the instructions issued and SQ_ACTIVE_INST_ANY "
                          "(quad-cycles a SIMD spent issuing, summed over the chip); issue_us_per_batch = SQ_ACTIVE_INST_ANY x 4 cycles / (1024 SIMDs x 2.4 GHz)")}
        lines += ["## instruction issue per 65 536-request batch, routed arrangement (SQ counters)", "",
                  "| kernel | waves per batch | VALU / wave | SALU / wave | LDS / wave | VMEM / wave | SMEM / wave | branch / wave | all / wave | issuing quad-cycles per batch | issue us per batch (1024 SIMDs, 2.4 GHz) |", "|---|---|---|---|---|---|---|---|---|---|---|"]
        tot_us = 0.0
        for k in PIPE:
            c = pm.get(k, {})
            if "SQ_WAVES" not in c:
                continue
            waves = sum(c["SQ_WAVES"])
            per = {n[9:].lower(): sum(c[n]) / waves for n in c if n.startswith("SQ_INSTS_")}
            allw = sum(per.values())
            anyq = sum(c.get("SQ_ACTIVE_INST_ANY", [0.0])) / batches
            us = anyq * 4 / (1024 * 2.4e3)
            tot_us += us
            issue["kernels"][k.replace("_multi", "")] = {"insts_per_wave": {a: round(b, 1) for a, b in per.items()}, "insts_per_wave_all": round(allw, 1), "waves_per_batch": round(waves / batches, 1),
                                                         "active_inst_any_quad_cycles_per_batch": round(anyq, 1), "issue_us_per_batch": round(us, 4)}
            lines.append(f"| {k} | {waves / batches:.0f} | {per.get('valu', 0):.0f} | {per.get('salu', 0):.0f} | {per.get('lds', 0):.0f} | {per.get('vmem_rd', 0) + per.get('vmem_wr', 0):.0f} | {per.get('smem', 0):.0f} | "
                         f"{per.get('branch', 0):.0f} | {allw:.0f} | {anyq:.0f} | {us:.3f} |")
        issue["issue_us_per_batch"] = round(tot_us, 4)
        lines += ["", f"**issue time of one 65 536-request batch, every kernel of the routed pipeline: {tot_us:.3f} us** (what the chip's 1024 SIMDs would need if every one of them issued all the time)", ""]
        json.dump(issue, open(os.path.join(base, "roofline_issue.json"), "w"), indent=1)
        out["issue"] = issue
if violations:
    lines += ["## CONSISTENCY VIOLATIONS (> 5 %)", ""] + [f"* {v}" for v in violations] + [""]
out["notes"] = notes
out["consistency"] = violations or "bench lines and 149 B x 65536 / ms_per_step agree within 5 %"
open(os.path.join(base, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(base, f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
sys.exit(1 if violations else 0)
