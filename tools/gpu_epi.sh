#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/bench_pw.py 512 512 256 256 2>/dev/null | grep -v amdgpu
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm --no-side-configs > gpurun_out/epi_bench.json 2> gpurun_out/epi_bench.err
python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/epi_bench.json").read().splitlines() if l.startswith("{")][-1])
print("bench: %.0fx %.3f ms pw %.3f (frac %.3f) dw %.3f fused %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["ms_per_step"], j["roofline"]["frac"], j["depthwise"]["ms_per_step"], j["fused"]["ms_per_step"]))
PY
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "golden or fused or alternate" 2>&1 | tail -2
