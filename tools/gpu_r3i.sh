#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03i}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_audio_data.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "resampl or config5 or randomised" 2>&1 | tail -4
timeout 300 python tests/devtools/fuzz_audio.py 200 2>&1 | tail -2
python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline --no-other-gemm > $O/c5.json 2> $O/c5.err
python -c "
import json; j=json.loads([l for l in open('$O/c5.json').read().splitlines() if l.startswith('{')][-1]); print(j['ms_per_step'], j['value'], j['resample'])"
