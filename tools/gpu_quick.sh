#!/bin/bash
# quick GPU check: selected tests + default bench (+ variants given as "ENV=.. ENV=.." strings in $VARIANTS, ';' separated) + kernel stats
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-q}; KEXPR=${2:-"matrix_pipe or goldens"}
O=$R/gpurun_out/$TAG; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "$KEXPR" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm"
$B > $O/bench_default.json 2> $O/bench_default.err
i=0
IFS=';' read -ra VS <<< "${VARIANTS:-}"
for v in "${VS[@]}"; do i=$((i+1)); env $v $B > $O/bench_v$i.json 2> $O/bench_v$i.err; echo "$v" > $O/bench_v$i.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-gemm > $O/bench_under_rocprof.json 2> $O/stats.err
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
cd $R; tail -4 $O/pytest.log
