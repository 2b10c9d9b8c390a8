#!/bin/bash
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
