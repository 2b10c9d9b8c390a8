#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1 B=512 T=1501
for f in dev $R/viet-asr_amd/lib/var_*.so; do
  [ $f = dev ] && export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so || export VASR_LIB_PATH=$f
  for upw in 0 1 2 4; do
    [ $upw = 0 ] && unset VASR_DW_UPW || export VASR_DW_UPW=$upw
    echo "== $(basename $f) upw=$upw"; python tools/bench_dw.py 51 75 2>&1 | grep -v amdgpu
    [ $f != dev ] && break
  done
done
