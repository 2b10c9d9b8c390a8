#!/bin/bash
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
