#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1
for f in $R/viet-asr_amd/lib/var_abl4w4b2nt.so $R/viet-asr_amd/lib/var_nt.so; do
  export VASR_LIB_PATH=$f; echo "== $(basename $f)"
  for nt in 0 1 2; do for d in 0 400 700; do
    echo -n "tile 6 nt $nt delay $d: "; VASR_PW_NT_MODE=$nt VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c26-60
  done; done
  for nt in 0 1; do echo -n "tile 1 nt $nt: "; VASR_PW_NT_MODE=$nt VASR_PW3_TILE=1 python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c26-60; done
done
