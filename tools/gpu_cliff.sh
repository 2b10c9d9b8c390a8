#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02j; mkdir -p $O; cd $R
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm"
for s in 10 10.3 12 15 20; do
  for t in 0 5 2; do
    VASR_PW3_TILE=$t $B --seconds $s > $O/b_${s}_t$t.json 2> $O/b_${s}_t$t.err
    python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/b_${s}_t$t.json").read().splitlines() if l.startswith("{")][-1])
    print("seconds=$s tile=$t: %.0fx %.3f ms pw %.3f dw %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["ms_per_step"], j["depthwise"]["ms_per_step"]))
except Exception as e: print("seconds=$s tile=$t ERR", e)
PY
  done
done
