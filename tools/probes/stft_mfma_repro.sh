#!/bin/bash
# Builds and runs tools/probes/stft_mfma_repro.hip in the two builds of csrc/frontend.hip (see the .hip file).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=${TMPDIR:-/tmp}
F="--offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/viet-asr_amd/csrc -ffp-contract=fast -Wno-unused-value"
/opt/rocm/bin/hipcc $F -c $R/tools/probes/stft_mfma_repro.hip -o $O/repro_main.o 2>/dev/null || exit 1
for v in slp noslp; do
  X=""; [ $v = noslp ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $F $X -c $R/viet-asr_amd/csrc/frontend.hip -o $O/repro_frontend_$v.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $O/repro_main.o $O/repro_frontend_$v.o -lpthread -o $O/stft_mfma_repro_$v || exit 1
  echo "== frontend.hip built $([ $v = slp ] && echo 'WITH the SLP vectoriser (packed-FP32 instructions: until round 6)' || echo 'with -fno-slp-vectorize (as shipped)')"
  for q in 0 1 3; do $O/stft_mfma_repro_$v 0 $q | sed "s/^/[attacker stream $q] /"; done
done
