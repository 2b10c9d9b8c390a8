#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02r; mkdir -p $O; cd $R
export VASR_BENCH_KEEP_AMAX=1
{ echo "== default"; python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu; VASR_DEBUG_NO_EPILOGUE=1 python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | sed 's/^/   no epilogue: /'
  for f in $R/viet-asr_amd/lib/var_fill*.so; do echo "== $(basename $f .so)"; VASR_LIB_PATH=$f python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu; VASR_LIB_PATH=$f VASR_DEBUG_NO_EPILOGUE=1 python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | sed 's/^/   no epilogue: /'; done; } | tee $O/fill.txt
