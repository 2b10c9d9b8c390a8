#!/usr/bin/env python3
"""Recompute the figures DESIGN.md section 4 / 6 / 7 quote from the committed profiles of a round (no GPU needed):

    python tools/round_numbers.py [r05]

Reads profiles/<tag>_bench_n1.json (the default bench line), <tag>_bench_c{2,4,5}.json, <tag>_bench_kernel_stats.csv (rocprofv3
--kernel-trace --stats of the same command), <tag>_pmc_traffic_summary.json (separate --pmc passes) and
<tag>_b1_vi12x1_serving.json, and prints the fractions both ways: from the bench line's per-launch timestamps and from the
rocprof averages x launches per step."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")


def line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


j = line(P("bench_n1.json"))
r, d, f, box = j["roofline"], j["depthwise"], j["fused"], j.get("box", {})
print(f"headline  {j['ms_per_step']:.3f} ms/step = {j['value']:.0f}x real time, {j['utts_per_sec']:.0f} utt/s   [{j['config']['workload'][:60]}...]")
print(f"box       bare MFMA stream {box.get('measured_mfma_tflops')} TF, streaming pass {box.get('measured_copy_gbs')} GB/s "
      f"(cache-resident {box.get('measured_copy_gbs_cache_resident')})")
print(f"GEMM      {r['ms_per_step']:.3f} ms in {r['launches_per_step']} launches: {r['achieved']:.0f} TF executed = {r['frac']:.3f} of {r['peak']:.0f} "
      f"= {r.get('frac_of_measured')} of measured;  family incl. fused launches {r['gemm_family']['ms_per_step']:.3f} ms: "
      f"{r['gemm_family']['frac']:.3f} / {r['gemm_family'].get('frac_of_measured')}")
print(f"depthwise {d['ms_per_step']:.3f} ms in {d['launches_per_step']} launches: {d['achieved']:.0f} GB/s = {d['frac']:.3f} of 8 TB/s "
      f"= {d.get('frac_of_measured_copy')} of the box's HBM copy rate, {d.get('frac_of_measured_copy_cache_resident')} of the cache-resident one")
print(f"fused     {f['ms_per_step']:.3f} ms in {f['launches_per_step']} launches: {f['achieved_TFLOPs_executed']} TF ({f['mfma_frac']}), "
      f"{f['achieved_GBps']} GB/s ({f['hbm_frac']})")
print(f"other     {j['other_ms_per_step']}   classes sum {r['gemm_family']['ms_per_step'] + d['ms_per_step'] + sum(j['other_ms_per_step'].values()):.3f} ms <= step")
sol = j.get("sol", {})
print(f"sol       MFMA floor at the measured rate {sol.get('mfma_floor_ms_at_measured_sustained')} ms (step = {sol.get('step_over_mfma_floor_sustained')} x), "
      f"fully fused HBM floor {sol.get('hbm_floor_ms_fully_fused_graph_at_measured_copy')} ms, batch-1 launches {sol.get('b1_launches')}")
og = j.get("other_gemm_arithmetic")
if og:
    print(f"fp32 mode {og['ms_per_step']} ms = {og['value']:.0f}x, {og['pointwise_fp32_equivalent_tflops']} TF = {og['pointwise_fp32_equivalent_tflops'] / 157.3:.3f} of 157.3")
cb = j.get("cpu_baseline", {})
print(f"cpu       {cb.get('value')}x on {cb.get('cores')} cores ({cb.get('kind')}), full batch {cb.get('full_batch', {}).get('value')}x")
for k, v in j.get("configs", {}).items():
    print(f"configs[{k}] in the default line: {v.get('ms_per_step')} ms = {v.get('value')}x")
lat = j.get("latency", {})
if lat:
    s = lat.get("vi12x1_b1", {}).get("vi12x1_b1_6.6s", {})
    print(f"latency   15x5 b1 10 s {lat.get('b1_10s_ms')} ms, b8 {lat.get('b8_10s_ms')} ms; serving shape: greedy {s.get('greedy_ms')}, search beam 50 "
          f"{s.get('beam50_search_ms')} / {s.get('beam50_search_ctc_like_ms')} (ctc-like), beam 100 {s.get('beam100_search_ms')} / {s.get('beam100_search_ctc_like_ms')}")
for c in (2, 4, 5):
    if os.path.exists(P(f"bench_c{c}.json")):
        k = line(P(f"bench_c{c}.json"))
        extra = ""
        if "beam" in k:
            extra = f"  search alone {k['beam']['ms_per_batch_alone']} ms, last-batch tail {k['beam']['last_batch_tail_ms']} ms"
        if "resample" in k:
            extra = f"  resample {k['resample']['ms_per_batch']} ms"
        print(f"--config {c}: {k['ms_per_step']} ms = {k['value']:.0f}x; GEMM {k['roofline']['frac']} / {k['roofline'].get('frac_of_measured')} of measured; "
              f"depthwise {k['depthwise']['frac']} / {k['depthwise'].get('frac_of_measured_copy')} of measured copy{extra}")

# ---- the same class times from the rocprof side: average duration x launches per step
steps = None
rows = list(csv.DictReader(open(P("bench_kernel_stats.csv"))))
short = lambda n: n.replace("void vasr::(anonymous namespace)::", "").replace("vasr::(anonymous namespace)::", "").split("(")[0]
cls = {"pw_gemm": 0.0, "dw_": 0.0, "dwpw_fused": 0.0}
for row in rows:
    n = short(row["Name"])
    if n.startswith("stft_logmel"):
        steps = int(row["Calls"])
for row in rows:
    n = short(row["Name"])
    if steps and n.startswith(("pw_gemm", "dw_", "dwpw_fused", "stft", "normalize", "logsoftmax", "ctc_collapse")):
        per_step = int(row["Calls"]) / steps
        us = float(row["AverageNs"]) / 1e3
        print(f"  rocprof {n[:58]:58s} {per_step:5.1f} / step x {us:7.1f} us")
        for key in cls:
            if n.startswith(key) and not (key == "pw_gemm" and "<4, 1, 2" in n):
                cls[key] += per_step * us / 1e3
if steps:
    print(f"rocprof class sums: plain GEMM {cls['pw_gemm']:.3f} ms, depthwise {cls['dw_']:.3f} ms, fused {cls['dwpw_fused']:.3f} ms "
          f"(bench line: {r['ms_per_step']:.3f} / {d['ms_per_step']:.3f} / {f['ms_per_step']:.3f})")
    tf = 3.0 * r["flops_per_step"] / (cls["pw_gemm"] * 1e-3) / 1e12
    print(f"  -> GEMM from the rocprof side: {tf:.0f} TF executed = {tf / 2500:.3f} of 2.5 PF")
if os.path.exists(P("pmc_traffic_summary.json")):
    t = json.load(open(P("pmc_traffic_summary.json")))
    for k, v in t.items():
        if isinstance(v, dict) and "fetch_bytes_corrected_per_launch" in v and k.startswith(("pw_gemm", "dw_toeplitz", "dwpw", "stft")):
            print(f"  traffic {k[:56]:56s} fetch {v['fetch_bytes_corrected_per_launch'] / 1e6:7.1f} MB + write {v.get('write_bytes_per_launch', 0) / 1e6:7.1f} MB per launch")
if os.path.exists(P("b1_vi12x1_serving.json")):
    print("serving  ", open(P("b1_vi12x1_serving.json")).read().strip())
