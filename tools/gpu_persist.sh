#!/bin/bash
# multi-round GEMM launches: one tile per workgroup against a persistent walk (VASR_PW_PERSIST workgroups)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so VASR_BENCH_KEEP_AMAX=1
for B in 64 128 256 512; do for p in 0 256; do
  [ $p = 0 ] && unset VASR_PW_PERSIST || export VASR_PW_PERSIST=$p
  echo -n "B=$B persist=$p: "; B=$B python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-70
done; done
