TL=/usr/local/lib/python3.10/dist-packages/torch/lib
# torch/lib holds libamdhip64.so / libhsa-runtime64.so WITHOUT their versioned names: the loader looks for libamdhip64.so.7 and would fall
# through to the binary's RUNPATH (/opt/rocm) -- symlinks under the sonames make LD_LIBRARY_PATH really select the bundled runtime
mkdir -p /tmp/trt; ln -sf $TL/libamdhip64.so /tmp/trt/libamdhip64.so.7; ln -sf $TL/libhsa-runtime64.so /tmp/trt/libhsa-runtime64.so.1
R=$GRAFT_REPO_ROOT; O=/tmp
F="--offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/viet-asr_amd/csrc -ffp-contract=fast -Wno-unused-value"
/opt/rocm/bin/hipcc $F -c $R/tools/probes/stft_mfma_repro.hip -o $O/repro_main.o 2>/dev/null
for v in slp noslp; do
  X=""; [ $v = noslp ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $F $X -c $R/viet-asr_amd/csrc/frontend.hip -o $O/repro_frontend_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $O/repro_main.o $O/repro_frontend_$v.o -lpthread -o $O/stft_mfma_repro_$v
  echo "== frontend.hip $v | system runtime (/opt/rocm, 7.2)"; $O/stft_mfma_repro_$v 0 0 | cut -c1-190
  echo "== frontend.hip $v | torch's bundled runtime (LD_LIBRARY_PATH=torch/lib)"; LD_LIBRARY_PATH=/tmp/trt:$TL ldd $O/stft_mfma_repro_$v | grep -i "amdhip"; LD_LIBRARY_PATH=/tmp/trt:$TL $O/stft_mfma_repro_$v 0 0 | cut -c1-190
done
