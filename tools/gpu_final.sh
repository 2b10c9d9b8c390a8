#!/bin/bash
# Final round-2 run: full GPU suite + smoke, round profile (kernel stats, PMC traffic, bench lines), SQ wait counters of the
# GEMM and depthwise kernels on isolated layers.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r02; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; rm -f $R/gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=8 > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
for c in 2 4 5; do timeout 300 python bench.py --config $c --steps $([ $c = 5 ] && echo 5 || echo 20) --warmup 3 --no-other-gemm > $O/bench_c$c.json 2> $O/bench_c$c.err; done
python bench.py --seconds 10.3 --no-cpu-baseline --no-other-gemm > $O/bench_10p3s.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
export VASR_BENCH_KEEP_AMAX=1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/sq1 -- python $R/tools/bench_pw.py 512 512 > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/sq2 -- python $R/tools/bench_pw.py 512 512 > $O/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/sq3 -- python $R/tools/bench_dw.py 75 > $O/sq3.log 2>&1
python - <<PY > $O/sq_summary.txt 2>&1
import csv, glob, collections
for tag in ("sq1", "sq2", "sq3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for key in ("pw_gemm_split_kernel", "dw_pair_kernel", "dw_toeplitz_kernel"):
                if key in k:
                    name = key + "<" + k.split(key + "<")[1].split(">")[0] + ">"
                    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    agg[name]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name, c in agg.items():
        print(tag, name, "launches", len(c["_dur_us"]) // max(1, len(c) - 1))
        for cn, v in sorted(c.items()):
            print("    %-28s mean %.4g" % (cn, sum(v) / len(v)))
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
cd $R; tail -4 $O/pytest_full.log; tail -2 $O/smoke.log; cat $O/sq_summary.txt | head -60
