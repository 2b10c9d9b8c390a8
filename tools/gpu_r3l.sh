#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03l}; mkdir -p $O; cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
for v in "" "VASR_PW3_TILE=5" "VASR_PW3_TILE=2"; do
  n=${v:-default}; env $v python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-other-gemm > $O/c5_$n.json 2> $O/c5_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/c5_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("%-18s %.3f ms/step  gemm %.3f (frac %.3f)  dw %.3f  fused %.3f" % ("$n", j["ms_per_step"], j["roofline"]["ms_per_step"], j["roofline"]["frac"], j["depthwise"]["ms_per_step"], j["fused"]["ms_per_step"]))
except Exception as e: print("$n bench ERR", e)
PY
done
