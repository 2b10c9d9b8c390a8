#!/bin/bash
# GPU box: depthwise policy comparison (auto = Toeplitz from 51 dense taps, VASR_DW_MFMA=0 packed FMAs only, =1 Toeplitz everywhere) over the bench workloads
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/dwcmp; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "matrix_pipe or goldens or bounded_memory or real_recordings" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'dw', d.get('depthwise',{}).get('ms_per_step'), d.get('depthwise',{}).get('frac'), 'gemm', d['roofline'].get('ms_per_step'))"; }
for env in "" "VASR_DW_MFMA=0" "VASR_DW_MFMA=1"; do
  for args in "" "--seconds 10.3" "--ragged" "--config 2" "--config 5 --steps 3 --warmup 1"; do
    env $env timeout 200 $B $args 2>$O/err.log | show "[$env] [$args]"
  done
done
