#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1
for f in $R/viet-asr_amd/lib/libvasr_hip_dev.so $R/viet-asr_amd/lib/var_*.so; do
  export VASR_LIB_PATH=$f; echo "== $(basename $f)"; timeout 300 python tools/probes/p4_gemm_probe.py 2>&1 | grep -v amdgpu | tail -2
done
