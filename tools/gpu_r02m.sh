#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02m; mkdir -p $O; cd $R
VARIANTS="VASR_DW_MFMA=1;VASR_DW_MFMA=1 VASR_DW_PPW=2;VASR_DW_MFMA=1 VASR_DW_PPW=1" bash tools/gpu_quick.sh r02m "matrix_pipe or goldens or config3 or real_record"
VASR_BENCH_KEEP_AMAX=1 python tools/bench_dw.py 51 63 75 87 > $O/dw.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clock -- python $R/tools/clock_probe.py 20000 > $O/clock.txt 2>&1
python - <<PY >> $O/clock.txt 2>&1
import csv, glob
for f in glob.glob("$O/clock/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sustained" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print("dispatch %s: GRBM_GUI_ACTIVE %.0f over %.3f ms -> effective clock %.3f GHz" % (r["Dispatch_Id"], float(r["Counter_Value"]), d / 1e6, float(r["Counter_Value"]) / d))
PY
find $O/clock -name "*.csv" ! -name "*counter_collection*" -delete
cat $O/clock.txt | tail -6; cat $O/dw.txt
