#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for f in dev $R/viet-asr_amd/lib/var_*.so; do
  [ $f = dev ] && export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so || export VASR_LIB_PATH=$f
  echo "== $(basename $f)"; python tools/probes/beam_pack.py 2>&1 | grep -v amdgpu | tail -5
done
