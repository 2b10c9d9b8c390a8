#!/bin/bash
# phase shift again, with the activation staging compiled out (what an LDS-DMA of pre-split activations would leave of it)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for f in $R/viet-asr_amd/lib/libvasr_hip_dev.so $R/viet-asr_amd/lib/var_*.so; do
  export VASR_LIB_PATH=$f; echo "== $(basename $f)"
  echo -n "tile 1: "; VASR_PW3_TILE=1 python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60
  for d in 0 300 500 700; do
    echo -n "tile 6 delay $d: "; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60
  done
done
