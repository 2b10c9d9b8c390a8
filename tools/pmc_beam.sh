#!/bin/bash
# SQ counters of the beam-search kernel on the reference's serving shape (batch 1, 12x1_vi, tools/b1_serving.py): two --pmc passes
# (kernel-trace only), reduced to per-launch means by beam width / posterior kind (the launches appear in the script's order).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-pmc_beam}; rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/sq1 -- python $R/tools/b1_serving.py --calls 10 > $O/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $O/sq2 -- python $R/tools/b1_serving.py --calls 10 > $O/sq2.log 2>&1
python - <<PY > $O/beam_sq_counters.txt 2>&1
import csv, glob, collections
for tag in ("sq1", "sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % tag):
        rows = list(csv.DictReader(open(f)))
        # launches of the beam kernel in dispatch order; b1_serving.py runs widths 20, 50, 100, 128, each first on the model's
        # posteriors then on CTC-like ones, 5 warm-ups + 10 timed calls per case = 15 launches per case
        disp = collections.OrderedDict()
        for r in rows:
            if "beam_wave_kernel" not in r["Kernel_Name"]: continue
            disp.setdefault(r["Dispatch_Id"], []).append(r)
        ids = list(disp)
        for k, d in enumerate(ids):
            case = k // 15
            name = "beam%s/%s" % ((20, 50, 100, 128)[min(case // 2, 3)], ("model", "ctc-like")[case % 2])
            for r in disp[d]:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            r0 = disp[d][0]
            agg[name]["_dur_us"].append((int(r0["End_Timestamp"]) - int(r0["Start_Timestamp"])) / 1e3)
    for name, c in agg.items():
        print(tag, name, "launches", len(c["_dur_us"]))
        for cn, v in sorted(c.items()):
            print("    %-24s mean %.5g" % (cn, sum(v) / len(v)))
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
cat $O/beam_sq_counters.txt | head -80
