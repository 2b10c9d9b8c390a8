#!/bin/bash
# GPU-box tasks, one parameterised script (run through gpurun):  bash tools/gpu.sh <task> [tag] [args...]
# `bash tools/gpu.sh list` prints the tasks.  Everything a task writes goes under gpurun_out/<tag>/ (scratch); what is to be
# judged is copied into profiles/ by hand afterwards (profiles/README.md maps files to claims).
# Rounds 1-5 kept one task per experiment here (42 KB of them): `git show a54b0b9:tools/gpu.sh` has them, DESIGN_HISTORY.md cites
# them by name.  What is left is what a round needs every time.
R=${GRAFT_REPO_ROOT:-/root/repo}
export HSA_ENABLE_IPC_MODE_LEGACY=0

stats_table() {   # <kernel_stats.csv> [rows]: the per-kernel table of a rocprofv3 --stats run
python - "$1" "${2:-24}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2])]:
    name = r["Name"].replace("vasr::(anonymous namespace)::", "").replace("void ", "")[:84]
    print("%-84s calls %5s avg %10.1f us  %5.1f%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
}

clean_traces() { find "$1" -name '*kernel_trace.csv' -delete; find "$1" -name '*counter_collection.csv' -delete; find "$1" -name '*.db' -delete; }

# ---- full: the whole GPU suite + smoke + default bench, as the driver runs them at round end
task_full() {
O=$R/gpurun_out/${1:-full}; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl; cd $R
(time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cp $R/gpurun_out/parity_errors.jsonl $O/ 2>/dev/null
tail -8 $O/pytest.log; tail -3 $O/smoke.log; head -c 600 $O/bench_default.json
}

# ---- quick: selected tests ($TESTS, default the beam + golden tests) + smoke + the default bench line's class times
task_quick() {
O=$R/gpurun_out/${1:-quick}; mkdir -p $O; cd $R
timeout 1200 python -m pytest ${TESTS:-tests/test_beam.py tests/test_gpu_parity.py} -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 600 python bench.py ${BENCH_ARGS:---no-cpu-baseline --no-other-gemm --no-side-configs} 2> $O/bench.err | tee $O/bench.json | python -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('step %.3f ms = %.0fx | gemm %.3f ms frac %.3f (of measured %s) | dw %.3f ms frac %.3f | fused %.3f ms | other %s | box %s' % (j['ms_per_step'], j['value'], j['roofline']['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_of_measured'), j['depthwise']['ms_per_step'], j['depthwise']['frac'], j['fused']['ms_per_step'], j['other_ms_per_step'], j.get('box')))"
}

# ---- prof: kernel-trace stats + the two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py $BENCH_ARGS`
#      -> <tag>/kernel_stats.csv, <tag>/pmc_traffic_summary.json, <tag>/bench_under_rocprof.json
#      (gpurun refuses --pmc together with the hip / hsa / memory-copy traces: counters get runs of their own)
task_prof() {
local O=$R/gpurun_out/${1:-prof}; mkdir -p $O   # (local: task_final calls this twice and keeps its own $O)
local ARGS PARGS f
ARGS=${BENCH_ARGS:---steps 10 --warmup 2 --no-cpu-baseline --no-other-gemm --no-side-configs}
PARGS=${PMC_BENCH_ARGS:-$ARGS}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py $PARGS > /dev/null 2> $O/fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py $PARGS > /dev/null 2> $O/write.err
python $R/tools/pmc_summary.py $O/fetch/*/*counter_collection.csv $O/write/*/*counter_collection.csv > $O/pmc_traffic_summary.json
clean_traces $O
stats_table $O/kernel_stats.csv 26
}

# ---- final: the round's record: full GPU suite, smoke, kernel stats + PMC traffic of the headline bench AND of the configs[4]
#      shard (512 x 30 s: tensors 12 x the Infinity Cache -- HBM-resident), the default line, configs 2 / 4 / 5 on their own,
#      serving-shape latencies + their per-kernel table
task_final() {
TAG=${1:-r06}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl; cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cp $R/gpurun_out/parity_errors.jsonl $O/parity_errors.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
task_prof $TAG/headline > $O/prof_headline.log 2>&1
BENCH_ARGS="--config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-other-gemm" task_prof $TAG/c5 > $O/prof_c5.log 2>&1
cd $R
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for c in 2 4 5; do timeout 300 python bench.py --config $c --steps $([ $c = 5 ] && echo 5 || echo 20) --warmup 3 --no-other-gemm --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err; done
python tools/b1_serving.py > $O/b1_vi12x1.json 2> /dev/null
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam > $O/b1_15x5.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1 -- python $R/tools/b1_serving.py --calls 30 > /dev/null 2> $O/stats_b1.err
cp $(find $O/stats_b1 -name '*kernel_stats.csv' | head -1) $O/b1_kernel_stats.csv; clean_traces $O
cd $R; tail -3 $O/pytest.log; tail -3 $O/smoke.log; ls $O $O/headline $O/c5
}

# ---- soak: the two beam-search kernels against each other and against themselves (run-to-run), seeds the tests do not use,
#      without an LM and with one in both of pyctcdecode's behaviours: soak <tag> [cases] [seed0] [max seconds]
task_soak() {
O=$R/gpurun_out/${1:-soak}; mkdir -p $O; cd $R
timeout 900 python tools/soak_beam.py ${2:-2000} ${3:-500000} 4 ${4:-400} 2> $O/soak.err | tee $O/soak.json
grep -v amdgpu $O/soak.err | tail -5
}

# ---- b1: the reference's serving shape (batch 1): latencies + the per-kernel table of a batch-1 call
task_b1() {
O=$R/gpurun_out/${1:-b1}; mkdir -p $O; cd $R
python tools/b1_serving.py 2> /dev/null | tee $O/b1_vi12x1.json
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam 2> /dev/null | tee $O/b1_15x5.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam --calls 30 > /dev/null 2> $O/stats.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; clean_traces $O; stats_table $O/kernel_stats.csv 20
}

task=${1:-list}; shift || true
if [ "$task" = list ]; then grep -E "^# ---- " "$0" | sed "s/^# ---- //"; exit 0; fi
if ! declare -F "task_$task" > /dev/null; then echo "unknown task $task (try: list)" >&2; exit 2; fi
"task_$task" "$@"
