#!/bin/bash
# Toeplitz depthwise occupancy variants: isolated layers at 64 x 10 s and 512 x 30 s, then the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1
for f in dev $R/viet-asr_amd/lib/var_*.so; do
  [ $f = dev ] && export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so || export VASR_LIB_PATH=$f
  echo "== $(basename $f)"
  B=64 T=501 python tools/bench_dw.py 33 51 63 75 2>&1 | grep -v amdgpu
  B=512 T=1501 python tools/bench_dw.py 51 75 2>&1 | grep -v amdgpu
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm --no-side-configs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('   bench: %.0fx %.3f ms pw %.3f dw %.3f (frac %.3f) fused %.3f' % (j['value'], j['ms_per_step'], j['roofline']['ms_per_step'], j['depthwise']['ms_per_step'], j['depthwise']['frac'], j['fused']['ms_per_step']))"
done
