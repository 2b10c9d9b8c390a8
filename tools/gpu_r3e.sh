#!/bin/bash
# fused on/off at the other workloads
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03e}; mkdir -p $O
cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so   # the build that reads the VASR_* switches
for cfg in "--config 5 --steps 3 --warmup 1" "--config 2 --steps 20 --warmup 5" "--seconds 10.3 --steps 20 --warmup 5" "--batch 16 --steps 20 --warmup 5" "--ragged --steps 20 --warmup 5"; do
for v in "" "VASR_FUSED=0" "VASR_FUSED_MIN_TILES=1"; do
  n=$(echo "${v:-default}_$cfg" | tr ' =-' '___'); env $v python bench.py $cfg --no-cpu-baseline --no-other-gemm > $O/b_$n.json 2> $O/b_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/b_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("%-60s %.3f ms/step  gemm-family %.3f  dw %.3f (frac %.3f)  fused %.3f ms / %d" % ("$n", j["ms_per_step"], j["roofline"]["ms_per_step"], j["depthwise"]["ms_per_step"], j["depthwise"]["frac"], j["fused"]["ms_per_step"], j["fused"]["launches_per_step"]))
except Exception as e: print("$n bench ERR", e)
PY
done; done
