#!/usr/bin/env python3
"""Summarise a gpurun_out/<tag> directory written by tools/gpu.sh quick (dev tool)."""
import csv, glob, json, os, sys
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    try:
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        j = json.loads(lines[-1]); r = j["roofline"]; dw = j["depthwise"]
        tag = os.path.basename(f)[6:-5]
        v = f.replace(".json", ".txt")
        if os.path.exists(v): tag += " [" + open(v).read().strip() + "]"
        print(f"{tag:40s} {j['value']:>9.0f}x  {j['ms_per_step']:.3f} ms | pw {r['ms_per_step']:.3f} ms frac {r['frac']:.3f} | dw {dw['ms_per_step']:.3f} ms frac {dw['frac']:.3f} | {j['other_ms_per_step']}")
        for k in ("beam", "resample"):
            if k in j: print("     ", k, {a: b for a, b in j[k].items() if a not in ("kernel", "bound", "lm")})
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json", ".err")).read()[-500:])
for f in glob.glob(os.path.join(d, "stats/*/*kernel_stats.csv")):
    for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
        n = r["Name"].replace("void vasr::(anonymous namespace)::", "").replace("vasr::(anonymous namespace)::", "")[:64]
        print(f"  {n:66s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={float(r['Percentage']):5.2f}")
p = os.path.join(d, "pytest.log")
if os.path.exists(p): print(open(p).read().strip().splitlines()[-2:])
