#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03g}; mkdir -p $O; cd $R
for cfg in "--config 5 --steps 4 --warmup 1" "--steps 20 --warmup 5" "--batch 128 --steps 10 --warmup 3" "--config 2 --batch 256 --steps 10 --warmup 3"; do
for v in "" "VASR_NT_MB=0" "VASR_NT_MB=60"; do
  n=$(echo "${v:-default}_$cfg" | tr ' =-' '___'); env $v python bench.py $cfg --no-cpu-baseline --no-other-gemm --no-side-configs > $O/b_$n.json 2> $O/b_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/b_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("%-58s %.3f ms/step  gemm %.3f (%.3f)  dw %.3f (frac %.3f)  fused %.3f" % ("$n", j["ms_per_step"], j["roofline"]["ms_per_step"], j["roofline"]["frac"], j["depthwise"]["ms_per_step"], j["depthwise"]["frac"], j["fused"]["ms_per_step"]))
except Exception as e: print("$n bench ERR", e)
PY
done; done
