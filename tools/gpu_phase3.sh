#!/bin/bash
# is the epilogue of the two-workgroups-per-CU tile bandwidth-bound (half the workgroups store in half the time) or not?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1
for f in $R/viet-asr_amd/lib/libvasr_hip_dev.so $R/viet-asr_amd/lib/var_abl4w4b2.so; do
  export VASR_LIB_PATH=$f; echo "== $(basename $f)"
  for t in 1 6; do for e in 0 2 1; do
    [ $e = 0 ] && unset VASR_DEBUG_NO_EPILOGUE || export VASR_DEBUG_NO_EPILOGUE=$e
    echo -n "tile $t epilogue-skip $e: "; VASR_PW3_TILE=$t python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60
  done; done
  unset VASR_DEBUG_NO_EPILOGUE
  for d in 300 600; do echo -n "tile 6 delay $d: "; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60; done
  export VASR_DEBUG_NO_EPILOGUE=1
  for d in 300 600; do echo -n "tile 6 delay $d no epilogue: "; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60; done
  unset VASR_DEBUG_NO_EPILOGUE
done
