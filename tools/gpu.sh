#!/bin/bash
# GPU-box tasks, one parameterised script (run through gpurun):  bash tools/gpu.sh <task> [args...]
# Every task was a one-off tools/gpu_<task>.sh in rounds 1-3 (DESIGN.md cites them as `tools/gpu.sh <task>`); the bodies
# are unchanged.  `bash tools/gpu.sh list` prints the tasks with their one-line purpose.
R=${GRAFT_REPO_ROOT:-/root/repo}

# ---- abl: run a timing tool against the default library and every ablation build: tools/gpu_abl.sh <tag> <MACRO> "<tool + args>"
task_abl() {
# run a timing tool against the default library and every ablation build: tools/gpu_abl.sh <tag> <MACRO> "<tool + args>"
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; MACRO=$2; TOOL=$3
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
echo "== default" > $O/abl.txt; $TOOL >> $O/abl.txt 2>&1
for f in $R/viet-asr_amd/lib/abl_${MACRO}_*.so; do echo "== $(basename $f)" >> $O/abl.txt; VASR_LIB_PATH=$f $TOOL >> $O/abl.txt 2>&1; done
cat $O/abl.txt
}

# ---- b1: 
task_b1() {
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-b1}; mkdir -p $O; cd $R
python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-other-gemm --no-side-configs > $O/b1.json 2>$O/b1.err
python - <<PY
import json
j=json.loads([l for l in open("$O/b1.json").read().splitlines() if l.startswith("{")][-1])
print("B=1: %.3f ms/step gemm %.3f (%d launches) dw %.3f (%d) fused %.3f other %s" % (j["ms_per_step"], j["roofline"]["ms_per_step"], j["roofline"]["launches_per_step"], j["depthwise"]["ms_per_step"], j["depthwise"]["launches_per_step"], j["fused"]["ms_per_step"], j["other_ms_per_step"]))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-other-gemm --no-side-configs > /dev/null 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:25]:
    print("%-90s calls %6s avg %8.1f ns  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), 100*float(r["TotalDurationNs"])/tot))
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
}

# ---- c5prof: 
task_c5prof() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-c5prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config ${2:-5} --steps 3 --warmup 1 --no-cpu-baseline --no-other-gemm > $O/bench.json 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-84s calls %5s avg %10.1f us  %5.1f%%" % (r["Name"].replace("vasr::(anonymous namespace)::","").replace("void ","")[:84], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
}

# ---- dwbig: 
task_dwbig() {
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1 B=512 T=1501
for f in dev $R/viet-asr_amd/lib/var_*.so; do
  [ $f = dev ] && export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so || export VASR_LIB_PATH=$f
  for upw in 0 1 2 4; do
    [ $upw = 0 ] && unset VASR_DW_UPW || export VASR_DW_UPW=$upw
    echo "== $(basename $f) upw=$upw"; python tools/bench_dw.py 51 75 2>&1 | grep -v amdgpu
    [ $f != dev ] && break
  done
done
}

# ---- dwcmp: GPU box: depthwise policy comparison (auto = Toeplitz from 51 dense taps, VASR_DW_MFMA=0 packed FMAs only, =1 Toeplitz everywhere) over the bench workloads
task_dwcmp() {
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
}

# ---- epi: 
task_epi() {
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/bench_pw.py 512 512 256 256 2>/dev/null | grep -v amdgpu
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm --no-side-configs > gpurun_out/epi_bench.json 2> gpurun_out/epi_bench.err
python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/epi_bench.json").read().splitlines() if l.startswith("{")][-1])
print("bench: %.0fx %.3f ms pw %.3f (frac %.3f) dw %.3f fused %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["ms_per_step"], j["roofline"]["frac"], j["depthwise"]["ms_per_step"], j["fused"]["ms_per_step"]))
PY
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "golden or fused or alternate" 2>&1 | tail -2
}

# ---- final_r03: Round-3 profile run: kernel stats, PMC traffic, bench lines (default with all configs, 10.3 s), SQ counters of the fused kernel
task_final_r03() {
# Round-3 profile run: kernel stats, PMC traffic, bench lines (default with all configs, 10.3 s), SQ counters of the fused kernel
# and of the 512-channel GEMM inside the bench workload.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
python bench.py --seconds 10.3 --no-cpu-baseline --no-other-gemm --no-side-configs > $O/bench_10p3s.json 2> /dev/null
for c in 2 4 5; do timeout 300 python bench.py --config $c --steps $([ $c = 5 ] && echo 5 || echo 20) --warmup 3 --no-other-gemm --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err; done
cd /tmp && export TMPDIR=/tmp
BQ="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-gemm --no-side-configs"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/sq1 -- $BQ > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/sq2 -- $BQ > $O/sq2.log 2>&1
python - <<PY > $O/sq_counters.txt 2>&1
import csv, glob, collections
for tag in ("sq1", "sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for key in ("pw_gemm_split_kernel", "dwpw_fused_kernel", "dw_toeplitz_kernel"):
                if key in k:
                    name = key + "<" + k.split(key + "<")[1].split(">")[0] + ">"
                    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    agg[name]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name, c in sorted(agg.items()):
        ncnt = max(1, len(c) - 1)
        print(tag, name, "dispatches", len(c["_dur_us"]) // ncnt)
        for cn, v in sorted(c.items()):
            print("    %-28s mean %.4g" % (cn, sum(v) / len(v)))
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
cd $R; ls $O; head -40 $O/sq_counters.txt
}

# ---- full: the whole GPU suite + smoke + default bench, as the driver runs them at round end
task_full() {
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
}

# ---- fvar: fused-kernel variants: default library and every viet-asr_amd/lib/var_*.so through bench.py; prints the fused / gemm / dw class times
task_fvar() {
# fused-kernel variants: default library and every viet-asr_amd/lib/var_*.so through bench.py; prints the fused / gemm / dw class times
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-fvar}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-gemm"
for f in default $R/viet-asr_amd/lib/var_*.so; do
  n=$(basename $f .so); [ $f = default ] && unset VASR_LIB_PATH || export VASR_LIB_PATH=$f
  $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("%-12s %.3f ms/step  gemm-family %.3f  dw %.3f  fused %.3f ms = %.1f us/launch" % ("$n", j["ms_per_step"], j["roofline"]["ms_per_step"], j["depthwise"]["ms_per_step"], j["fused"]["ms_per_step"], 1e3*j["fused"]["ms_per_step"]/max(1,j["fused"]["launches_per_step"])))
except Exception as e: print("$n bench ERR", e)
PY
done
}

# ---- phase: phase-shift experiment: 256 x 128 tiles on four wavefronts (two workgroups per CU), second arrival delayed
task_phase() {
# phase-shift experiment: 256 x 128 tiles on four wavefronts (two workgroups per CU), second arrival delayed
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
echo "== default tile"; python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu
for d in ${DELAYS:-0 200 400 600 800}; do
  echo "== tile 6, delay $d x 10 ns"; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu
done
}

# ---- phase2: phase shift again, with the activation staging compiled out (what an LDS-DMA of pre-split activations would leave of it)
task_phase2() {
# phase shift again, with the activation staging compiled out (what an LDS-DMA of pre-split activations would leave of it)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for f in $R/viet-asr_amd/lib/libvasr_hip_dev.so $R/viet-asr_amd/lib/var_*.so; do
  export VASR_LIB_PATH=$f; echo "== $(basename $f)"
  echo -n "tile 1: "; VASR_PW3_TILE=1 python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60
  for d in 0 300 500 700; do
    echo -n "tile 6 delay $d: "; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60
  done
done
}

# ---- phase3: is the epilogue of the two-workgroups-per-CU tile bandwidth-bound (half the workgroups store in half the time) or not?
task_phase3() {
# is the epilogue of the two-workgroups-per-CU tile bandwidth-bound (half the workgroups store in half the time) or not?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1
for f in $R/viet-asr_amd/lib/libvasr_hip_dev.so $R/viet-asr_amd/lib/var_abl4w4b2.so; do
  export VASR_LIB_PATH=$f; echo "== $(basename $f)"
  for t in 1 6; do for e in 0 2 1; do
    [ $e = 0 ] && unset VASR_DEBUG_NO_EPILOGUE || export VASR_DEBUG_NO_EPILOGUE=$e
    echo -n "tile $t epilogue-skip $e: "; VASR_PW3_TILE=$t python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60
  done; done
  unset VASR_DEBUG_NO_EPILOGUE
  for d in 300 600; do echo -n "tile 6 delay $d: "; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60; done
  export VASR_DEBUG_NO_EPILOGUE=1
  for d in 300 600; do echo -n "tile 6 delay $d no epilogue: "; VASR_PW3_TILE=6 VASR_PW_PHASE=$d python tools/bench_pw.py 512 512 2>/dev/null | grep -v amdgpu | cut -c1-60; done
  unset VASR_DEBUG_NO_EPILOGUE
done
}

# ---- quick: quick GPU check: selected tests + default bench (+ variants given as "ENV=.. ENV=.." strings in $VARIANTS, ';' separated) + kernel stats
task_quick() {
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
}

# ---- sq_dw: GPU box: SQ wait / issue counters of the depthwise kernels on an isolated layer (tools/bench_dw.py K), two PMC passes
task_sq_dw() {
# GPU box: SQ wait / issue counters of the depthwise kernels on an isolated layer (tools/bench_dw.py K), two PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}; K=${1:-75}; O=$R/gpurun_out/sqdw; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 VASR_BENCH_KEEP_AMAX=1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/a -- python $R/tools/bench_dw.py $K > $O/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/b -- python $R/tools/bench_dw.py $K > $O/b.log 2>&1
python - <<PY
import csv, glob, collections, re
for tag in ("a", "b"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        n = row["Kernel_Name"]
        if "dw_" not in n: continue
        acc[re.search(r"dw_\w+(<[^>]*>)?", n).group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k, {c: "%.3g" % (sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
}

# ---- tzocc: Toeplitz depthwise occupancy variants: isolated layers at 64 x 10 s and 512 x 30 s, then the bench line
task_tzocc() {
# Toeplitz depthwise occupancy variants: isolated layers at 64 x 10 s and 512 x 30 s, then the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_BENCH_KEEP_AMAX=1
for f in dev $R/viet-asr_amd/lib/var_*.so; do
  [ $f = dev ] && export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so || export VASR_LIB_PATH=$f
  echo "== $(basename $f)"
  B=64 T=501 python tools/bench_dw.py 33 51 63 75 2>&1 | grep -v amdgpu
  B=512 T=1501 python tools/bench_dw.py 51 75 2>&1 | grep -v amdgpu
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm --no-side-configs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('   bench: %.0fx %.3f ms pw %.3f dw %.3f (frac %.3f) fused %.3f' % (j['value'], j['ms_per_step'], j['roofline']['ms_per_step'], j['depthwise']['ms_per_step'], j['depthwise']['frac'], j['fused']['ms_per_step']))"
done
}

# ---- var: time the default library and every viet-asr_amd/lib/var_*.so: isolated GEMM layers + the bench line + (optional) goldens
task_var() {
# time the default library and every viet-asr_amd/lib/var_*.so: isolated GEMM layers + the bench line + (optional) goldens
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-var}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export VASR_BENCH_KEEP_AMAX=1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-gemm"
for f in default $R/viet-asr_amd/lib/var_*.so; do
  n=$(basename $f .so); [ $f = default ] && unset VASR_LIB_PATH || export VASR_LIB_PATH=$f
  echo "== $n"; python tools/bench_pw.py 512 512 256 256 512 1024 2>/dev/null | grep -v amdgpu
  $B > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("   bench: %.0fx %.3f ms pw %.3f dw %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["ms_per_step"], j["depthwise"]["ms_per_step"]))
except Exception as e: print("   bench ERR", e)
PY
  if [ -n "${KEXPR:-}" ]; then timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$KEXPR" 2>&1 | tail -1; fi
done
}

# ---- r4a: round 4, call A: full GPU suite (new whole-batch flip tests), default bench line, batch-1 serving-shape profile
task_r4a() {
set -u
TAG=${1:-r4a}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cp $R/gpurun_out/parity_errors.jsonl $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/b1_serving.py > $O/b1_vi.json 2> $O/b1_vi.err
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam > $O/b1_15x5.json 2>> $O/b1_vi.err
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --batch 8 --no-beam > $O/b8_15x5.json 2>> $O/b1_vi.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1_vi -- python $R/tools/b1_serving.py --calls 30 > /dev/null 2> $O/stats_b1_vi.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1_15 -- python $R/tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam --calls 30 > /dev/null 2> $O/stats_b1_15.err
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
cd $R; tail -5 $O/pytest.log; cat $O/b1_vi.json $O/b1_15x5.json $O/b8_15x5.json
}

# ---- beamlat: beam-search latency of the default dev library and every var_*.so (tools/probes/beam_lat.py)
task_beamlat() {
cd $R
for f in dev $R/viet-asr_amd/lib/var_*.so; do
  [ $f = dev ] && export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so || export VASR_LIB_PATH=$f
  case $f in
    *prof*|*p.so) echo "== $(basename $f) (section cycle counters, one launch per case)"
        ONCE=1 BATCHES=1 WIDTHS=${PROF_WIDTHS:-50,100} python tools/probes/beam_lat.py 2>&1 | grep -v amdgpu | grep -E "prof|/B1/" ;;
    *) echo "== $(basename $f)"; python tools/probes/beam_lat.py 2>&1 | grep -v amdgpu ;;
  esac
done
unset VASR_LIB_PATH
if [ -n "${FUZZ:-}" ]; then
  for f in $R/viet-asr_amd/lib/var_*.so; do
    case $f in *prof*|*p.so) continue ;; esac
    echo "== fuzz $(basename $f)"; VASR_LIB_PATH=$f timeout 600 python tests/devtools/fuzz_beam.py $FUZZ 0 2>&1 | tail -1
  done
fi
}

# ---- r4b: round 4, call B: full GPU suite, default bench, serving-shape latencies, depthwise walk variants at configs[1]
task_r4b() {
set -u
TAG=${1:-r4b2}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cp $R/gpurun_out/parity_errors.jsonl $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/b1_serving.py > $O/b1_vi.json 2> $O/b1_vi.err
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam > $O/b1_15x5.json 2>> $O/b1_vi.err
for u in 0 1 2 4 8; do
  echo "== VASR_DW_UPW=$u"; VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so VASR_DW_UPW=$u python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --no-other-gemm 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(j['ms_per_step'], 'dw', j['depthwise']['ms_per_step'], j['depthwise']['frac'], 'gemm', j['roofline']['ms_per_step'])"
done > $O/dw_upw.txt 2>&1
tail -5 $O/pytest.log; cat $O/b1_vi.json $O/b1_15x5.json $O/dw_upw.txt
}

# ---- fused64: the fused depthwise + pointwise kernel on 64-frame tiles (round 4): parity (fuzz on both tiles, goldens forced through
#      the 64-frame form), then ms per step of 12x1_vi / 15x5 at batches between the fill rules, per kernel choice
task_fused64() {
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/fused64; mkdir -p $O
if [ "${1:-all}" != time ]; then
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x \
  -k "fused_depthwise_pointwise_kernel or (alternate_kernel_paths and FUSED)" 2>&1 | tail -15
fi
[ "${1:-all}" = test ] && return
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
line() { python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f ms  fused %.4f (%s)  dw %.4f  gemm %.4f' % (j['ms_per_step'], j['fused']['ms_per_step'], j['fused'].get('launches_per_step'), j['depthwise']['ms_per_step'], j['roofline']['ms_per_step']))"; }
for cfg in ${FUSED64_CASES:-2:24 2:32 2:48 3:24 3:32 3:40}; do
  c=${cfg%%:*}; b=${cfg##*:}
  for v in off t128 t64 rule; do
    case "${FUSED64_V:-off t128 t64 rule}" in *$v*) ;; *) continue ;; esac
    case $v in off) e="VASR_FUSED=0" ;; t128) e="VASR_FUSED_MIN_TILES=1 VASR_FUSED_TILE=128" ;; t64) e="VASR_FUSED_MIN_TILES=1 VASR_FUSED_TILE=64" ;; rule) e="VASR_NONE=1" ;; esac
    echo -n "config $c batch $b  $v: "
    env $e python bench.py --config $c --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-other-gemm --no-side-configs 2>/dev/null | line
  done
done | tee $O/matrix.txt
}

# ---- pwlat: the small-batch latency GEMM (encoder_pw_lat.hip): bit-identity and golden tests, then batch-1 / batch-4 / batch-8
#      call latencies with the kernel off (VASR_PW_LAT=0), on (default) and extended to the 128 x 64 tile's batches (=2)
task_pwlat() {
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/pwlat; mkdir -p $O
if [ "${1:-all}" != time ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider -x \
  -k "results_do_not_depend or fused_path_matches_reference_goldens or row_independent or neural_module_dag or edge_cases or random_architectures or split_gemms" 2>&1 | tail -6
fi
[ "${1:-all}" = test ] && return
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
for v in 0 1 2; do
  echo "== VASR_PW_LAT=$v"
  VASR_PW_LAT=$v python tools/b1_serving.py --no-beam 2>/dev/null | tail -1
  VASR_PW_LAT=$v python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam 2>/dev/null | tail -1
  for b in 1 2 4 8; do
    echo -n "15x5 batch $b: "; VASR_PW_LAT=$v python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-other-gemm --no-side-configs 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f ms  gemm %.4f (%s)  dw %.4f' % (j['ms_per_step'], j['roofline']['ms_per_step'], j['roofline'].get('launches_per_step'), j['depthwise']['ms_per_step']))"
  done
done 2>&1 | tee $O/lat.txt
}

# ---- final_r04c: HEAD after the latency GEMM and the four-thread log-softmax: full GPU suite, default bench line, the serving-shape
#      latencies, configs[1]
task_final_r04c() {
set -u
O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cp $R/gpurun_out/parity_errors.jsonl $O/parity_errors.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python tools/b1_serving.py > $O/b1_vi.json 2> $O/b1_vi.err
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam > $O/b1_15x5.json 2>> $O/b1_vi.err
timeout 300 python bench.py --config 2 --steps 20 --warmup 3 --no-other-gemm --no-cpu-baseline > $O/bench_c2.json 2> /dev/null
tail -4 $O/pytest.log; tail -2 $O/smoke.log; tail -1 $O/b1_vi.json; tail -1 $O/b1_15x5.json
python - <<PY
import json
for n in ("bench_n1","bench_c2"):
    j=json.loads([l for l in open("$O/%s.json"%n).read().splitlines() if l.startswith("{")][-1])
    print(n, j["ms_per_step"], j["value"], "fused", j["fused"]["ms_per_step"], "roofline", j["roofline"]["frac"], "dw", j["depthwise"]["frac"])
    if "latency" in j: print({k:v for k,v in j["latency"].items() if not isinstance(v,dict)}); print(j["latency"]["vi12x1_b1"]["vi12x1_b1_6.6s"])
PY
}

# ---- final_r04b: after the 64-frame fused kernel: the two re-sized tests, default bench line, configs[1] on its own + its kernel stats
task_final_r04b() {
set -u
O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "full_size_properties or results_do_not_depend" 2>&1 | tail -3
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python bench.py --config 2 --steps 20 --warmup 3 --no-other-gemm --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --config 3 --batch 24 --steps 20 --warmup 3 --no-other-gemm --no-cpu-baseline --no-side-configs > $O/bench_15x5_b24.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-gemm --no-side-configs > /dev/null 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp "$f" $O/c2_kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete; rm -rf $O/stats
python - <<PY
import json,csv
for n in ("bench_n1","bench_c2","bench_15x5_b24"):
    j=json.loads([l for l in open("$O/%s.json"%n).read().splitlines() if l.startswith("{")][-1])
    print(n, j["ms_per_step"], j["value"], "fused", j["fused"]["ms_per_step"], "roofline", j["roofline"]["frac"], "dw", j["depthwise"]["frac"])
    if "configs" in j: print({k:(v["ms_per_step"], v.get("fused_ms_per_step")) for k,v in j["configs"].items()})
for r in list(csv.DictReader(open("$O/c2_kernel_stats.csv")))[:12]:
    print("%-100s calls %5s avg %8.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
}

# ---- final_r04: round-4 record: full GPU suite, kernel stats + PMC traffic of the bench, bench lines (default with every config,
#      10.3 s, configs 2 / 4 / 5 on their own), kernel stats of the reference's serving shape (batch 1, 12x1_vi, greedy + beam)
task_final_r04() {
set -u
TAG=${1:-r04}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cp $R/gpurun_out/parity_errors.jsonl $O/parity_errors.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
python bench.py --seconds 10.3 --no-cpu-baseline --no-other-gemm --no-side-configs > $O/bench_10p3s.json 2> /dev/null
for c in 2 4 5; do timeout 300 python bench.py --config $c --steps $([ $c = 5 ] && echo 5 || echo 20) --warmup 3 --no-other-gemm --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err; done
python tools/b1_serving.py > $O/b1_vi12x1.json 2> /dev/null
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam > $O/b1_15x5.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1 -- python $R/tools/b1_serving.py --calls 30 > /dev/null 2> $O/stats_b1.err
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
cd $R; tail -3 $O/pytest.log; cat $O/smoke.log | tail -2; ls $O
}

# ---- final_r05: round-5 record: full GPU suite, smoke, kernel stats + PMC traffic of the bench, the default line (every config, latency,
#      sol, box normalisers), configs 2 / 4 / 5 on their own, serving-shape latencies + their per-kernel table
task_final_r05() {
set -u
TAG=${1:-r05}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cp $R/gpurun_out/parity_errors.jsonl $O/parity_errors.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
for c in 2 4 5; do timeout 300 python bench.py --config $c --steps $([ $c = 5 ] && echo 5 || echo 20) --warmup 3 --no-other-gemm --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err; done
python tools/b1_serving.py > $O/b1_vi12x1.json 2> /dev/null
python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam > $O/b1_15x5.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1 -- python $R/tools/b1_serving.py --calls 30 > /dev/null 2> $O/stats_b1.err
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*.db' -delete
cd $R
# the tile a CTC head folded into the 512 -> 1024 GEMM would need (1024 x 64 on 8 wavefronts: spills), against the shipped rule, kernel-only
{ echo "== 512 -> 1024, tile rule"; VASR_BENCH_KEEP_AMAX=1 python tools/bench_pw.py 512 1024 2>&1 | grep -v amdgpu
  echo "== 512 -> 1024, 1024 x 64 tile (VASR_PW3_TILE=9)"; VASR_BENCH_KEEP_AMAX=1 VASR_PW3_TILE=9 python tools/bench_pw.py 512 1024 2>&1 | grep -v amdgpu; } > $O/head_fold_tile.txt 2>&1
tail -3 $O/pytest.log; cat $O/smoke.log | tail -2; cat $O/head_fold_tile.txt; ls $O
}

# ---- probes: build the HIP probes from their sources (the binaries are not tracked) and run them
task_probes() {
cd $R/tools/probes
for p in place_probe tr_probe; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $p.hip -o $p && ./$p; done
}

# ---- r5probe: round 5, kill criterion of the fused 512-channel kernel: (a) matrix / vector wavefronts sharing a SIMD, register-resident
#      (tools/probes/coissue_probe.hip); (b) the 512 x 64 GEMM tile with 4 extra wavefronts issuing a depthwise producer's FMAs
task_r5probe() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5probe}; mkdir -p $O; cd $R/tools/probes
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 coissue_probe.hip -o coissue_probe && ./coissue_probe > $O/coissue.txt 2>&1
cd $R
{
echo "== default library, tile rule (512 x 128)"; python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
echo "== default library, 512 x 64 tile (VASR_PW3_TILE=8)"; VASR_PW3_TILE=8 python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
for f in $R/viet-asr_amd/lib/var_x*.so; do
  echo "== $(basename $f), 512 x 64 tile"; VASR_LIB_PATH=$f VASR_PW3_TILE=8 python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
done
echo "== x0_ns (no activation staging), 512 x 128 tile"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_x0_ns.so python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
echo "== depthwise alone"; python tools/bench_dw.py 51 63 75 2>&1 | grep -v amdgpu
echo "== normalisers"; python tools/mfma_sustained.py 2>&1 | grep -v amdgpu; python tools/copy_bw.py 2>&1 | grep -v amdgpu
} > $O/probe.txt 2>&1
cat $O/coissue.txt $O/probe.txt
}

# ---- r5b: round 5, call B: the four-wavefront beam search (beam_group.hip): device-vs-oracle tests, fuzz, A/B against the one-wavefront
#      kernel, latencies at the reference's serving shape for VASR_BEAM_GROUP = 0 / 2 / 4; the band-limited goldens and the flip tests
#      (new bounds); the fused-512 probe again WITHOUT the bench entry's maxima pre-pass (kernel-only times)
task_r5b() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5b}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_beam.py -x -q -m gpu > $O/pytest_beam.log 2>&1; tail -5 $O/pytest_beam.log
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
for g in 0 2 4; do
  echo "== VASR_BEAM_GROUP=$g: beam_lat (B = 1)"; VASR_BEAM_GROUP=$g BATCHES=1 python tools/probes/beam_lat.py 2>&1 | grep -v amdgpu
  echo "== VASR_BEAM_GROUP=$g: serving shape"; VASR_BEAM_GROUP=$g python tools/b1_serving.py --calls 30 2>&1 | grep -v amdgpu | tail -3
done > $O/beam_lat.txt 2>&1
unset VASR_LIB_PATH
cat $O/beam_lat.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "band_limited or goldens" > $O/pytest_band.log 2>&1; tail -5 $O/pytest_band.log
timeout 900 python -m pytest tests/test_gpu_flips.py -x -q -m gpu > $O/pytest_flips.log 2>&1; tail -5 $O/pytest_flips.log
{
export VASR_BENCH_KEEP_AMAX=1
echo "== kernel-only times (VASR_BENCH_KEEP_AMAX=1: no maxima pre-pass in the bench entry)"
echo "== default library, tile rule (512 x 128)"; python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
echo "== default library, 512 x 64 tile (VASR_PW3_TILE=8)"; VASR_PW3_TILE=8 python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
for f in $R/viet-asr_amd/lib/var_x*.so; do
  echo "== $(basename $f), 512 x 64 tile"; VASR_LIB_PATH=$f VASR_PW3_TILE=8 python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
done
echo "== x0_ns (no activation staging), 512 x 128 tile"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_x0_ns.so python tools/bench_pw.py 512 512 2>&1 | grep -v amdgpu
echo "== depthwise alone"; python tools/bench_dw.py 51 63 75 2>&1 | grep -v amdgpu
unset VASR_BENCH_KEEP_AMAX
} > $O/probe_kernel_only.txt 2>&1
cat $O/probe_kernel_only.txt
cp $R/gpurun_out/parity_errors.jsonl $O/ 2>/dev/null
}

# ---- r5prof: section cycle counters of the beam kernels at the serving shape (var_gprof.so / var_wprof.so: -DVASR_BEAM_PROF builds)
task_r5prof() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5prof}; mkdir -p $O; cd $R
{
for g in 4 2; do echo "#### group W=$g"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_gprof.so VASR_BEAM_GROUP=$g python tools/probes/beam_prof.py 2>&1 | grep -v amdgpu; done
echo "#### wave"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_wprof.so VASR_BEAM_GROUP=0 python tools/probes/beam_prof.py 2>&1 | grep -v amdgpu
} > $O/beam_prof.txt 2>&1
cat $O/beam_prof.txt
}

# ---- r5c: beam_group.hip iteration: beam tests, serving-shape latency per VASR_BEAM_GROUP in $BGROUPS, section counters
task_r5c() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5c}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_beam.py -x -q -m gpu > $O/pytest_beam.log 2>&1; tail -5 $O/pytest_beam.log
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
for g in ${BGROUPS:-0 4 8}; do
  echo "== VASR_BEAM_GROUP=$g: serving shape"; VASR_BEAM_GROUP=$g python tools/b1_serving.py --calls 30 2>&1 | grep -v amdgpu | tail -1
  echo "== VASR_BEAM_GROUP=$g: beam_lat (B = 1)"; VASR_BEAM_GROUP=$g BATCHES=1 python tools/probes/beam_lat.py 2>&1 | grep -v amdgpu
done > $O/beam_lat.txt 2>&1
unset VASR_LIB_PATH
cat $O/beam_lat.txt
for g in ${BGROUPS:-0 4 8}; do [ $g = 0 ] && continue; echo "#### group W=$g"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_gprof.so VASR_BEAM_GROUP=$g WIDTHS=100 python tools/probes/beam_prof.py 2>&1 | grep -v amdgpu; done > $O/beam_prof.txt 2>&1
cat $O/beam_prof.txt
}

# ---- r5d: beam_group.hip with a larger pass (var_g716.so: 716 pairs per pass instead of 358): fuzz against the oracle AND against the
#      one-wavefront kernel (bit equality), serving-shape latency, section counters -- after the same for the default build
task_r5d() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5d}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_beam.py -x -q -m gpu > $O/pytest_beam.log 2>&1; tail -3 $O/pytest_beam.log
{
echo "== default build, serving shape"; VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so python tools/b1_serving.py --calls 30 2>&1 | grep -v amdgpu | tail -1
echo "== default build, section counters"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_gprof.so VASR_BEAM_GROUP=4 WIDTHS=50,100 python tools/probes/beam_prof.py 2>&1 | grep -v amdgpu
echo "== g716: fuzz (oracle + bit equality with the one-wavefront kernel)"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_g716.so timeout 600 python tests/devtools/fuzz_beam.py 400 0 2>&1 | grep -v amdgpu | tail -8
echo "== g716, serving shape"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_g716.so python tools/b1_serving.py --calls 30 2>&1 | grep -v amdgpu | tail -1
echo "== g716, section counters"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_gprof716.so VASR_BEAM_GROUP=4 WIDTHS=100 python tools/probes/beam_prof.py 2>&1 | grep -v amdgpu
} > $O/beam.txt 2>&1
cat $O/beam.txt
}

# ---- r5e: beam_group.hip as shipped (716 pairs per pass, W = 4): the beam tests, 2 000 fuzz cases, serving-shape latency, section counters
task_r5e() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5e}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_beam.py -x -q -m gpu > $O/pytest_beam.log 2>&1; tail -3 $O/pytest_beam.log
{
echo "== fuzz: 2 000 cases against the oracle and (bit equality) against the one-wavefront kernel"; timeout 900 python tests/devtools/fuzz_beam.py 2000 400 2>&1 | grep -v amdgpu | tail -5
echo "== serving shape, four-wavefront kernel (default below 16 utterances)"; python tools/b1_serving.py --calls 30 2>&1 | grep -v amdgpu | tail -1
echo "== serving shape, one-wavefront kernel (VASR_BEAM_GROUP=0, devtools build)"; VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so VASR_BEAM_GROUP=0 python tools/b1_serving.py --calls 30 2>&1 | grep -v amdgpu | tail -1
echo "== beam_lat, four-wavefront kernel"; BATCHES=1,8 python tools/probes/beam_lat.py 2>&1 | grep -v amdgpu
echo "== beam_lat, one-wavefront kernel"; VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so VASR_BEAM_GROUP=0 BATCHES=1,8 python tools/probes/beam_lat.py 2>&1 | grep -v amdgpu
echo "== section counters (var_gprof.so)"; VASR_LIB_PATH=$R/viet-asr_amd/lib/var_gprof.so VASR_BEAM_GROUP=4 WIDTHS=50,100 python tools/probes/beam_prof.py 2>&1 | grep -v amdgpu
} > $O/beam.txt 2>&1
cat $O/beam.txt
}

# ---- r5f: front-end iteration: the front-end / golden / edge-case tests, the front-end fuzz, then the default bench line's class times
task_r5f() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5f}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "golden or stage or edge or stft or front or independent or batch_size" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tests/devtools/fuzz_frontend.py 200 2>&1 | grep -v amdgpu | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-other-gemm --no-side-configs 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('step %.3f ms  other %s  box %s' % (j['ms_per_step'], j['other_ms_per_step'], j['box']))"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-gemm --no-side-configs > /dev/null 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); grep -E "stft|normalize" $f | cut -c1-160
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
}

# ---- r5soak: the two beam-search kernels against each other and against themselves (run-to-run), seeds the tests do not use
task_r5soak() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5soak}; mkdir -p $O; cd $R
timeout 400 python tools/soak_beam.py ${2:-2000} ${3:-500000} 4 ${4:-170} 2> $O/soak.err | tee $O/soak.json
grep -v amdgpu $O/soak.err | tail -5
}

# ---- r5g: tie-break by key in both beam kernels: the beam tests, the soak, serving-shape latency, configs[3]
task_r5g() {
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5g}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_beam.py -x -q -m gpu -p no:cacheprovider > $O/pytest_beam.log 2>&1; tail -3 $O/pytest_beam.log
timeout 300 python tools/soak_beam.py 2000 500000 4 170 2> $O/soak.err | tee $O/soak.json
python tools/b1_serving.py 2> /dev/null | tee $O/b1_vi12x1.json
timeout 200 python bench.py --config 4 --steps 20 --warmup 3 --no-other-gemm --no-cpu-baseline 2> /dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('config 4: %.3f ms/step, beam %s' % (j['ms_per_step'], j.get('beam')))"
}

task=${1:-list}; shift || true
if [ "$task" = list ]; then grep -E "^# ---- " "$0" | sed "s/^# ---- //"; exit 0; fi
if ! declare -F "task_$task" > /dev/null; then echo "unknown task $task (try: list)" >&2; exit 2; fi
"task_$task" "$@"
