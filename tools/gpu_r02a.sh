#!/bin/bash
# Round-2 GPU call A: new tests (configs 3/4/5, RCCL world of one, f16x2 GEMM mode, real audio, tightened tolerances),
# then bench lines for the GEMM arithmetics / tiles and the other BASELINE configs.  Everything lands in gpurun_out/r02a.
set -u
O=gpurun_out/r02a; mkdir -p $O; rm -f gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider \
  -k "goldens or split_gemms or batch_size or random_arch or alternate or real_record or configs or executor or edge or stage or no_writes or rccl" \
  > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B > $O/bench_default.json 2> $O/bench_default.err
$B --gemm f16x2 --no-other-gemm > $O/bench_f16x2.json 2> $O/bench_f16x2.err
VASR_PW3_TILE=6 $B --gemm f16x2 --no-other-gemm > $O/bench_f16x2_t6.json 2> $O/bench_f16x2_t6.err
VASR_PW3_TILE=2 $B --gemm f16x2 --no-other-gemm > $O/bench_f16x2_t2.json 2> $O/bench_f16x2_t2.err
$B --gemm f16x2 --no-other-gemm --seconds 10.3 > $O/bench_f16x2_10p3s.json 2> $O/bench_f16x2_10p3s.err
$B --gemm bf16x3 --no-other-gemm --seconds 10.3 > $O/bench_bf16x3_10p3s.json 2> $O/bench_bf16x3_10p3s.err
$B --config 4 --no-other-gemm > $O/bench_c4.json 2> $O/bench_c4.err
$B --config 4 --no-other-gemm --no-overlap > $O/bench_c4_serial.json 2> $O/bench_c4_serial.err
$B --config 5 --no-other-gemm --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err
$B --config 2 --no-other-gemm > $O/bench_c2.json 2> $O/bench_c2.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_f16x2 -o f16x2 -- python $GRAFT_REPO_ROOT/bench.py --gemm f16x2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm > $GRAFT_REPO_ROOT/$O/prof_f16x2.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_f16x2 -name "*kernel_stats.csv" -exec cp {} $O/f16x2_kernel_stats.csv \;
find $O/prof_f16x2 -type f ! -name "*stats*" -delete 2>/dev/null
tail -5 $O/pytest_a.log
