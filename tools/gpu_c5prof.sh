#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-c5prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config ${2:-5} --steps 3 --warmup 1 --no-cpu-baseline --no-other-gemm > $O/bench.json 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-84s calls %5s avg %10.1f us  %5.1f%%" % (r["Name"].replace("vasr::(anonymous namespace)::","").replace("void ","")[:84], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
