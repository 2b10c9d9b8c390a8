#!/bin/bash
# run a timing tool against the default library and every ablation build: tools/gpu_abl.sh <tag> <MACRO> "<tool + args>"
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; MACRO=$2; TOOL=$3
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
echo "== default" > $O/abl.txt; $TOOL >> $O/abl.txt 2>&1
for f in $R/viet-asr_amd/lib/abl_${MACRO}_*.so; do echo "== $(basename $f)" >> $O/abl.txt; VASR_LIB_PATH=$f $TOOL >> $O/abl.txt 2>&1; done
cat $O/abl.txt
