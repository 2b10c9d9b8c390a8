#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of bench.py.
# Writes small summaries under gpurun_out/<tag>/ ; copy the ones to keep into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-other-gemm --no-side-configs > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-other-gemm --no-side-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-other-gemm --no-side-configs > /dev/null 2>&1
python "$R/tools/pmc_summary.py" "$O"/fetch/*/*counter_collection.csv "$O"/write/*/*counter_collection.csv > "$O/pmc_traffic_summary.json"
find "$O" -name '*kernel_trace.csv' -delete
find "$O" -name '*counter_collection.csv' -delete
find "$O" -name '*.db' -delete
timeout 300 python "$R/bench.py" > "$O/bench_plain.json" 2> /dev/null
ls -R "$O" | head -30
