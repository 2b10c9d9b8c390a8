#!/bin/bash
# round 3, call B: the fused dw -> pw kernel: parity (forced on the goldens), config-3 test, bench A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03b; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so   # the build that reads the VASR_* switches
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -p no:cacheprovider -x \
  -k "FUSED or config3 or config2" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm"
timeout 300 $B > $O/bench_fused.json 2> $O/bench_fused.err
VASR_FUSED=0 timeout 300 $B > $O/bench_unfused.json 2> $O/bench_unfused.err
tail -15 $O/pytest.log
python - <<'PY'
import json,os
for n in ("fused","unfused"):
    try:
        d=json.loads(open(f"{os.environ.get('GRAFT_REPO_ROOT','/root/repo')}/gpurun_out/r03b/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["roofline"]["ms_per_step"], d["depthwise"]["ms_per_step"], d.get("fused"))
    except Exception as e: print(n, "ERR", e)
PY
