#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
export VASR_BENCH_KEEP_AMAX=1
T="python tools/bench_pw.py 512 512 256 256"
{ echo "== default"; $T; echo "== default, no epilogue"; VASR_DEBUG_NO_EPILOGUE=1 $T
  for f in $R/viet-asr_amd/lib/abl_VASR_ABLATE_*.so; do echo "== $(basename $f)"; VASR_LIB_PATH=$f $T; echo "== $(basename $f), no epilogue"; VASR_LIB_PATH=$f VASR_DEBUG_NO_EPILOGUE=1 $T; done
  echo "== tile 6"; VASR_PW3_TILE=6 $T; echo "== tile 2"; VASR_PW3_TILE=2 $T; } > $O/abl.txt 2>&1
cat $O/abl.txt
