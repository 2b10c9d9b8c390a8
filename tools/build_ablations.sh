#!/bin/bash
# build/abl/libvasr_N.so for the -DNAME=N ablation values given: tools/build_ablations.sh <source.hip> <MACRO> n1 n2 ...
set -e
R=$(cd $(dirname $0)/.. && pwd); SRC=$1; MACRO=$2; shift 2
mkdir -p $R/build/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/viet-asr_amd/csrc -ffp-contract=fast -D$MACRO=$n -c $R/viet-asr_amd/csrc/$SRC -o $R/build/abl/${SRC}_$n.o
  OBJS=$(ls $R/build/csrc/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/build/abl/${SRC}_$n.o -o $R/viet-asr_amd/lib/abl_${MACRO}_$n.so
done
ls -la $R/viet-asr_amd/lib/
