#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03f}; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
(time python bench.py) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_default.json").read().splitlines() if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["depthwise"]["frac"], j["depthwise"]["frac_of_achievable"])
print(json.dumps(j.get("configs"), indent=1)[:3000]); print(j.get("latency")); print(j.get("cpu_baseline")); print(j.get("other_gemm_arithmetic"))
PY
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "bench" 2>&1 | tail -5
