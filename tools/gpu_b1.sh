#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-b1}; mkdir -p $O; cd $R
python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-other-gemm --no-side-configs > $O/b1.json 2>$O/b1.err
python - <<PY
import json
j=json.loads([l for l in open("$O/b1.json").read().splitlines() if l.startswith("{")][-1])
print("B=1: %.3f ms/step gemm %.3f (%d launches) dw %.3f (%d) fused %.3f other %s" % (j["ms_per_step"], j["roofline"]["ms_per_step"], j["roofline"]["launches_per_step"], j["depthwise"]["ms_per_step"], j["depthwise"]["launches_per_step"], j["fused"]["ms_per_step"], j["other_ms_per_step"]))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-other-gemm --no-side-configs > /dev/null 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:25]:
    print("%-90s calls %6s avg %8.1f ns  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), 100*float(r["TotalDurationNs"])/tot))
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
