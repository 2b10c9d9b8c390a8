#!/bin/bash
# A/B of the beam kernels on ONE box: viet-asr_amd/lib/var_beam_old.so (a library built with the previous beam_group.hip / beam_wave.hip,
# not tracked) against the current library -- serving-shape latencies (four-wavefront kernel) and configs[3] (one-wavefront kernel)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/${1:-r5ab}; mkdir -p $O
timeout 300 python -m pytest tests/test_beam.py -x -q -m gpu -p no:cacheprovider > $O/pytest_beam.log 2>&1; tail -1 $O/pytest_beam.log
timeout 200 python tools/soak_beam.py 2000 500000 4 ${2:-100} 2> /dev/null | tee $O/soak.json
OLD=$PWD/viet-asr_amd/lib/var_beam_old.so
for i in 1 2; do
  echo "old $(VASR_LIB_PATH=$OLD python tools/b1_serving.py --calls 90 2>/dev/null)"
  echo "new $(python tools/b1_serving.py --calls 90 2>/dev/null)"
done | tee $O/ab.txt
c4() { timeout 200 python bench.py --config 4 --steps 20 --warmup 3 --no-other-gemm --no-cpu-baseline 2> /dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('config 4: %.3f ms/step, search alone %s ms, acoustic alone %s ms' % (j['ms_per_step'], j['beam']['ms_per_batch_alone'], j['beam']['acoustic_ms_per_batch_alone']))"; }
{ echo "old $(VASR_LIB_PATH=$OLD c4)"; echo "new $(c4)"; } | tee $O/ab_c4.txt
