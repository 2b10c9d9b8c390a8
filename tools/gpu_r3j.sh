#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03j}; mkdir -p $O; cd $R
for i in 1 2; do
python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-other-gemm > $O/c4_$i.json 2> $O/c4.err
python -c "
import json; j=json.loads([l for l in open('$O/c4_$i.json').read().splitlines() if l.startswith('{')][-1]); print('c4', j['ms_per_step'], j['value'], j['beam']['ms_per_batch_alone'], j['beam']['last_batch_tail_ms'])"
done
python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-other-gemm --no-side-configs > $O/b1.json 2>$O/b1.err
python -c "
import json; j=json.loads([l for l in open('$O/b1.json').read().splitlines() if l.startswith('{')][-1]); print('b1', j['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm > $O/def.json 2>$O/def.err
python -c "
import json; j=json.loads([l for l in open('$O/def.json').read().splitlines() if l.startswith('{')][-1]); print('default', j['ms_per_step'], j['latency'], j['roofline']['frac'], j['roofline']['gemm_family']['frac'], j['fused']['mfma_frac'], j['fused']['hbm_frac'])"
