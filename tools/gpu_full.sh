#!/bin/bash
# Full GPU suite + round profile (kernel stats, PMC traffic passes, plain bench line with cpu_baseline, the other configs).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02}
O=$R/gpurun_out/$TAG; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=15 > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
for c in 2 4 5; do timeout 300 python bench.py --config $c --steps $([ $c = 5 ] && echo 5 || echo 20) --warmup 3 --no-other-gemm > $O/bench_c$c.json 2> $O/bench_c$c.err; done
python bench.py --seconds 10.3 --no-cpu-baseline --no-other-gemm > $O/bench_10p3s.json 2> /dev/null
tail -5 $O/pytest_full.log; tail -2 $O/smoke.log
