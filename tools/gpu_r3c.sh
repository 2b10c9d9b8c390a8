#!/bin/bash
# bench A/B lines (fused on/off) + golden parity, quick
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03c}; mkdir -p $O
cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so   # the build that reads the VASR_* switches
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm"
for v in "" "VASR_FUSED=0"; do
  n=${v:-default}; env $v $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("%-14s %.3f ms/step  gemm-family %.3f (frac %.3f)  dw %.3f (frac %.3f)  fused %.3f ms / %d" % ("$n", j["ms_per_step"], j["roofline"]["ms_per_step"], j["roofline"]["frac"], j["depthwise"]["ms_per_step"], j["depthwise"]["frac"], j["fused"]["ms_per_step"], j["fused"]["launches_per_step"]))
except Exception as e: print("$n bench ERR", e)
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "${KEXPR:-goldens}" 2>&1 | tail -3
