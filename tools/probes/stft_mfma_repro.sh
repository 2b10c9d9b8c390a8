#!/bin/bash
# Builds and runs tools/probes/stft_mfma_repro.hip: this library's front-end launches (csrc/frontend.hip linked in) next to a synthetic 16-bit
# MFMA kernel from a second host thread.  frontend.hip is compiled with the LIBRARY's flags (csrc/Makefile), with and without -fno-slp-vectorize.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=${TMPDIR:-/tmp}
MAIN="--offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/viet-asr_amd/csrc -ffp-contract=fast -Wno-unused-value"
LIBF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/viet-asr_amd/csrc -Wall -Wno-unused-function -ffp-contract=fast ${VIS--fvisibility=hidden -fvisibility-inlines-hidden}"
/opt/rocm/bin/hipcc $MAIN -c $R/tools/probes/stft_mfma_repro.hip -o $O/repro_main.o 2>/dev/null || exit 1
for v in slp noslp; do
  X=""; [ $v = noslp ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $LIBF $X -c $R/viet-asr_amd/csrc/frontend.hip -o $O/repro_frontend_$v.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $O/repro_main.o $O/repro_frontend_$v.o -lpthread -o $O/stft_mfma_repro_$v || exit 1
  echo "== frontend.hip built $([ $v = slp ] && echo 'WITH the SLP vectoriser (packed-FP32 instructions: until round 6)' || echo 'with -fno-slp-vectorize (as shipped)') [visibility flags: ${VIS--fvisibility=hidden -fvisibility-inlines-hidden}]"
  $O/stft_mfma_repro_$v 0 0 $1 | cut -c1-200
done
