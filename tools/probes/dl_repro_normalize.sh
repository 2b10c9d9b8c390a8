#!/bin/bash
# The FP64 normalisation kernels under the same harness: the product build (they ask for 152 KB of LDS: alone on their compute unit) against
# a variant in which they run with their 64 bytes as before (kAloneLds = 0), per-feature normalisation on, attackers with 24 KB / 0 bytes of LDS.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=${TMPDIR:-/tmp}/vasr_dl_norm; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/probes/mfma_attacker.hip -o $O/attacker.so 2>/dev/null || exit 1
g++ -O2 -std=c++17 -w -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ $R/tools/probes/dl_repro.cpp -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o $O/dl_repro || exit 1
python $R/tools/probes/stft_mfma_repro_dump.py $O/in > /dev/null 2>&1
S=$O/src; rm -rf $S; mkdir -p $S/viet-asr_amd; cp -r $R/include $S/include; cp -r $R/viet-asr_amd/csrc $S/viet-asr_amd/csrc
sed -i 's/constexpr int kAloneLds = 152 \* 1024;/constexpr int kAloneLds = 0;/' $S/viet-asr_amd/csrc/frontend.hip; grep -c "kAloneLds = 0" $S/viet-asr_amd/csrc/frontend.hip
make -C $S/viet-asr_amd/csrc -j16 OUT=$O/lib OBJ=$O/obj OBJD=$O/objd $O/lib/libvasr_hip.so > $O/build.log 2>&1 || { tail -3 $O/build.log; exit 1; }
for lds in 24576 0; do
  echo "== normalisation kernels with 64 bytes of LDS (as before)"; $O/dl_repro $O/lib/libvasr_hip.so $O/attacker.so $O/in 1 $lds | grep attacker
  echo "== product (152 KB: alone on their compute unit)";          $O/dl_repro $R/viet-asr_amd/lib/libvasr_hip.so $O/attacker.so $O/in 1 $lds | grep attacker
done
