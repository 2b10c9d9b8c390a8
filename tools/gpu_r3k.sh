#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03k}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "goldens or stage_entry or edge_cases or real_recordings or dag" 2>&1 | tail -3
timeout 300 python tests/devtools/fuzz_frontend.py 150 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm --no-side-configs > $O/def.json 2>$O/def.err
python -c "
import json; j=json.loads([l for l in open('$O/def.json').read().splitlines() if l.startswith('{')][-1]); print('default', j['ms_per_step'], j['other_ms_per_step'])"; done
