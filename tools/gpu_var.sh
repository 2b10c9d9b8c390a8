#!/bin/bash
# time the default library and every viet-asr_amd/lib/var_*.so: isolated GEMM layers + the bench line + (optional) goldens
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-var}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export VASR_BENCH_KEEP_AMAX=1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm"
for f in default $R/viet-asr_amd/lib/var_*.so; do
  n=$(basename $f .so); [ $f = default ] && unset VASR_LIB_PATH || export VASR_LIB_PATH=$f
  echo "== $n"; python tools/bench_pw.py 512 512 256 256 512 1024 2>/dev/null | grep -v amdgpu
  $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("   bench: %.0fx %.3f ms pw %.3f dw %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["ms_per_step"], j["depthwise"]["ms_per_step"]))
except Exception as e: print("   bench ERR", e)
PY
  if [ -n "${KEXPR:-}" ]; then timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$KEXPR" 2>&1 | tail -1; fi
done
