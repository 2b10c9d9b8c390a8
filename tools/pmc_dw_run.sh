#!/bin/bash
# GPU box: PMC passes over the isolated depthwise kernel (tools/pmc_dw.py C K): LDS conflicts, VALU/LDS busy, HBM bytes.
set -u
C=${1:-512}; K=${2:-75}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_dw_${C}_${K}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/$tag" -- python "$R/tools/pmc_dw.py" $C $K > /dev/null 2> "$O/$tag.err"
done
python "$R/tools/pmc_summary.py" $(find "$O" -name '*counter_collection.csv') > "$O/summary.json" 2> "$O/summary.err"
find "$O" -name '*kernel_trace.csv' -delete; find "$O" -name '*counter_collection.csv' -delete; find "$O" -name '*.db' -delete
cat "$O/summary.json"
