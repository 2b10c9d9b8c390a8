#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for f in dev $R/viet-asr_amd/lib/var_*.so; do
  [ $f = dev ] && export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so || export VASR_LIB_PATH=$f
  echo "== $(basename $f)"; python tests/devtools/bench_beam.py 2>&1 | grep -v amdgpu | grep "128"
  timeout 300 python tests/devtools/fuzz_beam.py 150 0 2>&1 | tail -1
done
