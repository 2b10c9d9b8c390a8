#!/bin/bash
# build viet-asr_amd/lib/var_<tag>.so (a DEVTOOLS build) with extra compile flags for ONE source: tools/build_variant.sh <source.hip> <tag> "<flags>"
set -e
R=$(cd $(dirname $0)/.. && pwd); SRC=$1; TAG=$2; FLAGS=$3
mkdir -p $R/build/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/viet-asr_amd/csrc -ffp-contract=fast -DVASR_DEVTOOLS $FLAGS -c $R/viet-asr_amd/csrc/$SRC -o $R/build/abl/${SRC}_$TAG.o -Rpass-analysis=kernel-resource-usage 2> $R/build/abl/${SRC}_$TAG.rpt
echo "$TAG: kernels with scratch: $(grep -c 'ScratchSize \[bytes/lane\]: [1-9]' $R/build/abl/${SRC}_$TAG.rpt)"
OBJS=$(ls $R/build/csrc_dev/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/build/abl/${SRC}_$TAG.o -o $R/viet-asr_amd/lib/var_$TAG.so
