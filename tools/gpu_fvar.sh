#!/bin/bash
# fused-kernel variants: default library and every viet-asr_amd/lib/var_*.so through bench.py; prints the fused / gemm / dw class times
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-fvar}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-gemm"
for f in default $R/viet-asr_amd/lib/var_*.so; do
  n=$(basename $f .so); [ $f = default ] && unset VASR_LIB_PATH || export VASR_LIB_PATH=$f
  $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("%-12s %.3f ms/step  gemm-family %.3f  dw %.3f  fused %.3f ms = %.1f us/launch" % ("$n", j["ms_per_step"], j["roofline"]["ms_per_step"], j["depthwise"]["ms_per_step"], j["fused"]["ms_per_step"], 1e3*j["fused"]["ms_per_step"]/max(1,j["fused"]["launches_per_step"])))
except Exception as e: print("$n bench ERR", e)
PY
done
