#!/bin/bash
# Round-2 GPU call B: Toeplitz depthwise + DPP maxima + f16x2 default; re-run of the tests that failed in call A.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02b; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider \
  -k "matrix_pipe or goldens or split_gemms or batch_size or random_arch or alternate or real_record or config3 or config4 or executor or strongly or long_clips or five_minute or no_writes" \
  > $O/pytest_b.log 2>&1; echo "pytest rc=$?" >> $O/pytest_b.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B > $O/bench_default.json 2> $O/bench_default.err
VASR_DW_MFMA=0 $B --no-other-gemm > $O/bench_dwfma.json 2> $O/bench_dwfma.err
$B --gemm bf16x3 --no-other-gemm > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
$B --config 4 --no-other-gemm > $O/bench_c4.json 2> $O/bench_c4.err
$B --config 5 --no-other-gemm --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err
$B --config 2 --no-other-gemm > $O/bench_c2.json 2> $O/bench_c2.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-gemm > $O/bench_under_rocprof.json 2> $O/stats.err
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
cd $R; tail -8 $O/pytest_b.log
