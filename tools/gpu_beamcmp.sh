#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for f in $R/viet-asr_amd/lib/libvasr_hip_dev.so $R/viet-asr_amd/lib/var_t512s1024.so; do
  export VASR_LIB_PATH=$f; echo "== $(basename $f)"; python tests/devtools/bench_beam.py 2>&1 | grep -v amdgpu
done
