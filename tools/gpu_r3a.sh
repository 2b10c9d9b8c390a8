#!/bin/bash
# round 3, call A: beam search after the cap / cache rework, the new parity cases, row-independent mode; fuzz; beam timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03a; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests/test_beam.py tests/test_gpu_round3.py tests/test_serving.py tests/test_gpu_configs.py -m gpu -q --timeout 900 -p no:cacheprovider \
  -k "not bench" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python tests/devtools/fuzz_beam.py 500 0 > $O/fuzz_beam.log 2>&1
timeout 300 python tests/devtools/bench_beam.py > $O/bench_beam.log 2>&1
timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
cp $R/gpurun_out/parity_errors.jsonl $O/ 2>/dev/null
tail -5 $O/pytest.log; tail -3 $O/fuzz_beam.log; cat $O/bench_beam.log
