#!/bin/bash
# the whole GPU suite + smoke + default bench, as the driver runs them at round end
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-full}; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
(time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cp $R/gpurun_out/parity_errors.jsonl $O/ 2>/dev/null
tail -8 $O/pytest.log; tail -2 $O/smoke.log; head -c 600 $O/bench_default.json
