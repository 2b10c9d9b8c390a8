#!/bin/bash
# phase-shift experiment: 256 x 128 tiles on four wavefronts (two workgroups per CU), second arrival delayed
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
echo "== default tile"; python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu
for d in ${DELAYS:-0 200 400 600 800}; do
  echo "== tile 6, delay $d x 10 ns"; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu
done
