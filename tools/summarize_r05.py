#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_profile_r05.sh (gpurun_out/<tag>/) into <tag>_rocprof_summary.{md,json},
roofline_traffic.json and roofline_issue.json (copied to profiles/ by hand).  Exits 1 when the bench line of the traced run and the trace disagree
by more than 5 % on the dominant kernel's average duration (the line's HIP events against rocprofv3's timestamps).  Kernel durations: End - Start timestamps of the kernel trace.
`dominant_kernel` = the kernel with the largest summed duration among the batch kernels; its roofline line uses the formula
of bench.py: bytes per request x average requests per launch / average duration / 8000 GB/s.  Only the dispatches of the
non-replayed stream count (for one table: the last N launches of k_front / k_eval2 — the earlier ones are the residency pass).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM
section): raw = FETCH + WRITE, corrected = 2 x FETCH + WRITE (an upper bound for this mix of 16..64-byte random reads).
roofline_issue.json (round 5): from the SQ counter passes (sq_counters.json, GUBER_FUSE_EP=0 so that the three kernels are counted one
by one): per kernel and 65536-request batch the instructions issued by kind and the quad-cycles in which a SIMD was issuing
(SQ_ACTIVE_INST_ANY); bench.py turns their sum into roofline.issue = issue time / step time over 1024 SIMDs at 2.4 GHz.
usage: summarize_r05.py <tag> [pmc_batches]"""
import collections, csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 512
base = os.path.join(ROOT, "gpurun_out", tag)
ALG = {"k_front": 76, "k_eval2": 73, "k_front_multi": 76, "k_eval2_multi": 73,
       "k_part": 20, "k_own": 56, "k_eval3": 73, "k_part_multi": 20, "k_own_multi": 56, "k_eval3_multi": 73, "k_evalpart_multi": 93}
SINGLE = ("k_front", "k_eval2", "k_part", "k_own", "k_eval3")
violations = []
notes = []
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


for key, d, lastn, title in (("fused", "trace_fused", None, "the default bench command (12 logical shards, one dispatcher, 3 streams): up to four tables per launch, owner-partitioned pipeline"),
                             ("shards_1", "trace_s1", 1024 + 16 + 4096 + 1024, "one table (--shards 1, 1024 timed batches), default policy: the two-launch pipeline with claims, one batch in flight"),
                             ("shards_1_part", "trace_s1_part", 1024 + 16 + 4096 + 1024, "one table, GUBER_PIPELINE=part: the owner-partitioned pipeline, one batch in flight")):
    path = find(d, "*kernel_trace.csv")
    if not path:
        continue
    per, grid, cols = collections.defaultdict(list), collections.defaultdict(list), {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            k = name.replace("guber::", "")
            per[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            gx = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
            if k.startswith("k_own"):                                  # 256 owners x 256 threads per batch, whatever the batch size: count batches as full ones
                gx = gx
            grid[k].append(gx)
            cols[k] = {c: row.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size")}
    bl = bench_line(d + ".log")
    lines += [f"## {title}", ""]
    if bl:
        lines += [f"bench line of this traced run: value {bl['value'] / 1e9:.3f} G decisions/s, ms_per_step {bl['ms_per_step']}, roofline.frac (pipeline bytes / step) {bl['roofline']['frac']}, "
                  f"HIP events of the line: {bl['roofline']['kernel_avg_us']}", ""]
    lines += ["| kernel | launches | avg us | min us | p50 us | max us | VGPR | SGPR | LDS B | scratch B | avg requests per launch | algorithmic GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    ks = {}
    for name, dd in per.items():
        g = grid[name]
        if lastn and name in SINGLE:
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
            # the line's HIP events cover the PROFILE segment: the last `launches_profiled` launches of the kernel in the trace
            ev = bl["roofline"]["kernel_avg_us"].get(dom)
            nprof = int((bl["roofline"].get("launches_profiled") or {}).get(dom, 0))
            if ev and nprof and len(per[dom]) >= nprof:
                seg = per[dom][-nprof:]
                seg_avg = sum(seg) / len(seg) / 1e3
                out["dominant_kernel"]["profile_segment"] = {"launches": nprof, "trace_avg_us": round(seg_avg, 3), "bench_line_hip_events_avg_us": ev}
                lines += [f"profile segment (the last {nprof} launches of {dom}, what the line's HIP events bracket): trace {seg_avg:.2f} us, bench line {ev} us", ""]
                if abs(ev - seg_avg) / seg_avg > 0.05:
                    # not a violation of the line's roofline (that is the pipeline figure, checked below): the line's per-kernel figure is
                    # labelled a diagnostic — a HIP event pair on a stream that shares the GPU with two other streams brackets the kernel
                    # AND the time it waited for a free slot, rocprofv3's timestamps bracket the kernel alone
                    notes.append(f"{dom}: the line's HIP events give {ev} us per launch, the trace {seg_avg:.2f} us over the same {nprof} launches "
                                 "(events include the wait behind the other streams' kernels; the line labels this figure a diagnostic)")
            pipe = 149 * 65536 / (bl["ms_per_step"] * 1e-3) / 1e9 / 8000
            if abs(pipe - bl["roofline"]["frac"]) / pipe > 0.05:
                violations.append(f"roofline.frac {bl['roofline']['frac']} is not 149 B x 65536 / ms_per_step / 8 TB/s = {pipe:.5f}")
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
            if k in SINGLE and s.startswith("s1"):
                v = v[-NB:]
            pm.setdefault(s, {}).setdefault(k, {})[c] = (sum(v) / len(v), len(v))
for s in sorted(pm):
    ctrs = sorted({c for k in pm[s] for c in pm[s][k]})
    lines += [f"## hardware counters per launch, {'one table' if s == 's1' else 'one table, GUBER_PIPELINE=part' if s == 's1p' else '12 shards fused'} (separate PMC passes, {NB} distinct batches, non-replayed)", "",
              "| kernel | launches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
    for k in sorted(pm[s]):
        if k in ALG:
            n = max(v[1] for v in pm[s][k].values())
            lines.append(f"| {k} | {n} | " + " | ".join(f"{pm[s][k][c][0]:.1f}" if c in pm[s][k] else "" for c in ctrs) + " |")
    lines.append("")
    out["counters"][s] = {k: {c: v[0] for c, v in cs.items()} for k, cs in pm[s].items()}
tr, tr_raw, trf, trf_raw = {}, {}, {}, {}
lines += ["## HBM-side traffic per 65536-request batch", "", "| configuration | kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | raw bytes | corrected bytes (2xFETCH+WRITE) | algorithmic bytes | raw / algorithmic | corrected / algorithmic |", "|---|---|---|---|---|---|---|---|---|"]
per_launch_batches = {}
try:
    for k, st in out["kernels"].get("fused", {}).items():
        per_launch_batches[k] = st["avg_requests_per_launch"] / 65536.0
except Exception:
    pass
for s_, label, ks_ in (("s1", "one table, claims", ("k_front", "k_eval2")), ("s1p", "one table, owner-partitioned", ("k_part", "k_own", "k_eval3")),
                       ("s12", "12 shards fused, per batch of the launch", ("k_part_multi", "k_own_multi", "k_eval3_multi"))):
    tot_raw = tot_cor = tot_alg = 0
    for k in ks_:
        c = pm.get(s_, {}).get(k, {})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            nb = per_launch_batches.get(k, 1.0) if s_ == "s12" else 1.0
            f, w = c["FETCH_SIZE"][0] / nb, c["WRITE_SIZE"][0] / nb
            raw, cor = (f + w) * 1024, (2 * f + w) * 1024
            base_k = k.replace("_multi", "")
            if s_ != "s12":
                tr[base_k], tr_raw[base_k] = int(cor), int(raw)
            else:
                trf[base_k], trf_raw[base_k] = int(cor), int(raw)
            tot_raw += raw; tot_cor += cor; tot_alg += ALG[k] * 65536
            lines.append(f"| {label} | {k} | {f:.1f} | {w:.1f} | {raw:.0f} | {cor:.0f} | {ALG[k] * 65536} | {raw / (ALG[k] * 65536):.2f} | {cor / (ALG[k] * 65536):.2f} |")
    if tot_alg:
        lines.append(f"| {label} | **pipeline** | | | {tot_raw:.0f} | {tot_cor:.0f} | {tot_alg} | **{tot_raw / tot_alg:.2f}** | **{tot_cor / tot_alg:.2f}** |")
lines.append("")
if tr:
    json.dump({"token": tr, "token_raw": tr_raw, "token_fused": trf, "token_fused_raw": trf_raw, "source": f"profiles/{tag}_rocprof_summary.md",
               "note": "PMC bytes per 65536-request batch and kernel on the non-replayed stream, one table: token = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, an "
                       "upper bound for 64-byte requests), token_raw = (FETCH_SIZE + WRITE_SIZE)*1024; token_fused[_raw] = the same per batch of a fused launch (12 shards); the bench line weights every kernel by the batches it carried (bench.py pipeline_traffic)"},
              open(os.path.join(base, "roofline_traffic.json"), "w"), indent=1)
# ---- the issue side (SQ counters): instructions per wave and SIMD-issue time per batch ----
try:
    sq = json.load(open(os.path.join(base, "sq_counters.json")))
    issue = {"clock_GHz": 2.4, "simds": 1024, "kernels": {}, "source": f"profiles/{tag}_sq_counters.txt",
             "note": ("SQ counter passes (rocprofv3 --pmc, one group per pass) over the headline configuration with GUBER_FUSE_EP=0: per launch of a fused kernel "
                      "(~3.9 batches) SQ_WAVES, SQ_INSTS_* and SQ_ACTIVE_INST_ANY (quad-cycles a SIMD spent issuing, summed over the chip); per batch = per launch x 1024 waves / SQ_WAVES. "
                      "issue_us_per_batch = SQ_ACTIVE_INST_ANY x 4 cycles / (1024 SIMDs x 2.4 GHz)")}
    lines += ["## instruction issue (SQ counters, GUBER_FUSE_EP=0: the three kernels counted one by one)", "",
              "| kernel | waves per launch | VALU / wave | SALU / wave | LDS / wave | VMEM / wave | SMEM / wave | branch / wave | all / wave | issuing quad-cycles per batch | issue us per batch (1024 SIMDs, 2.4 GHz) |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    tot_us = 0.0
    for k in ("k_part_multi", "k_own_multi", "k_eval3_multi"):
        c = {n: v["avg_per_launch"] for n, v in sq.get(k, {}).items()}
        w = c.get("SQ_WAVES")
        if not w:
            continue
        per = {n[9:].lower(): c[n] / w for n in c if n.startswith("SQ_INSTS_")}
        allw = sum(per.values())
        batches = w / 1024.0
        anyq = c.get("SQ_ACTIVE_INST_ANY", 0.0) / batches
        us = anyq * 4 / (1024 * 2.4e3)
        tot_us += us
        issue["kernels"][k.replace("_multi", "")] = {"insts_per_wave": {a: round(b, 1) for a, b in per.items()}, "insts_per_wave_all": round(allw, 1), "waves_per_batch": 1024,
                                                     "active_inst_any_quad_cycles_per_batch": round(anyq, 1), "issue_us_per_batch": round(us, 4)}
        lines.append(f"| {k} | {w:.0f} | {per.get('valu', 0):.0f} | {per.get('salu', 0):.0f} | {per.get('lds', 0):.0f} | {per.get('vmem_rd', 0) + per.get('vmem_wr', 0):.0f} | {per.get('smem', 0):.0f} | "
                     f"{per.get('branch', 0):.0f} | {allw:.0f} | {anyq:.0f} | {us:.3f} |")
    issue["issue_us_per_batch"] = round(tot_us, 4)
    lines += ["", f"**issue time of one 65536-request batch: {tot_us:.3f} us** (the three kernels; what the chip's 1024 SIMDs would need if every one of them issued all the time)", ""]
    json.dump(issue, open(os.path.join(base, "roofline_issue.json"), "w"), indent=1)
    out["issue"] = issue
except Exception as ex:   # noqa: BLE001
    notes.append(f"no SQ counters for the issue roofline: {ex!r}")
if violations:
    lines += ["## CONSISTENCY VIOLATIONS (> 5 %)", ""] + [f"* {v}" for v in violations] + [""]
if notes:
    lines += ["## notes", ""] + [f"* {v}" for v in notes] + [""]
out["notes"] = notes
out["consistency"] = violations or "bench line and trace agree within 5 %"
open(os.path.join(base, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(base, f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
sys.exit(1 if violations else 0)
