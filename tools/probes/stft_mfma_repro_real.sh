R=$GRAFT_REPO_ROOT; O=/tmp
python $R/tools/probes/stft_mfma_repro_dump.py /tmp/repro_in 2>&1 | tail -1
F="--offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/viet-asr_amd/csrc -ffp-contract=fast -Wno-unused-value"
/opt/rocm/bin/hipcc $F -c $R/tools/probes/stft_mfma_repro.hip -o $O/repro_main.o 2>/dev/null
for v in slp noslp; do
  X=""; [ $v = noslp ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $F $X -c $R/viet-asr_amd/csrc/frontend.hip -o $O/repro_frontend_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $O/repro_main.o $O/repro_frontend_$v.o -lpthread -o $O/stft_mfma_repro_$v
  echo "== frontend.hip $v, the library's real inputs"; $O/stft_mfma_repro_$v 0 0 /tmp/repro_in | cut -c1-200
done
