#!/bin/bash
# GPU box: stall-breakdown PMC passes over one isolated pointwise GEMM (tools/pmc_gemm.py cin cout mode).
set -u
CIN=${1:-512}; COUT=${2:-512}; MODE=${3:-bf16x3}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_gemm_${CIN}_${COUT}_${MODE}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/$tag" -- python "$R/tools/pmc_gemm.py" $CIN $COUT $MODE > /dev/null 2> "$O/$tag.err"
done
python "$R/tools/pmc_summary.py" $(find "$O" -name '*counter_collection.csv') > "$O/summary.json" 2> "$O/summary.err"
find "$O" -name '*kernel_trace.csv' -delete; find "$O" -name '*counter_collection.csv' -delete; find "$O" -name '*.db' -delete; find "$O" -name '*agent_info.csv' -delete
python - <<P
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "pw_gemm" in k: print(k); [print("  ",a,b) for a,b in sorted(v.items())]
P
