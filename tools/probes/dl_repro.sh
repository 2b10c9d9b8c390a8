#!/bin/bash
# Stand-alone C++ reproducer of profiles/r06_concurrency.txt (no Python, no torch):
#   1. builds the library a second time into /tmp with frontend.hip compiled WITH the SLP vectoriser (packed-FP32 instructions), as it was
#      built until round 6 -- `make DWFLAGS_frontend.hip=` overrides the one flag of csrc/Makefile --, next to the product build;
#   2. builds the synthetic attacker (mfma_attacker.hip: nothing but v_mfma_f32_32x32x16_f16 in a loop) and the harness (dl_repro.cpp: dlopen()s a
#      library and the attacker, runs vasr_melspec_f32 on 64 x 10 s from one host thread and the attacker from another, two streams, compares every
#      result with the idle-device one);
#   3. runs the harness on both libraries.  Expected on MI355X: the variant wrong in ~100 % of the calls next to the attacker, the product in none.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=${TMPDIR:-/tmp}/vasr_dl_repro; mkdir -p $O
make -C $R/viet-asr_amd/csrc -j16 DWFLAGS_frontend.hip= OUT=$O OBJ=$O/obj OBJD=$O/objd $O/libvasr_hip.so > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/probes/mfma_attacker.hip -o $O/attacker.so 2>/dev/null || exit 1
g++ -O2 -std=c++17 -w -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ $R/tools/probes/dl_repro.cpp -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o $O/dl_repro || exit 1
python $R/tools/probes/stft_mfma_repro_dump.py $O/in 2>&1 | tail -1      # (the inputs: synthetic audio, hann window, Slaney filterbank as raw float32 files)
echo "== frontend.hip WITH packed-FP32 instructions (the build until round 6)"; $O/dl_repro $O/libvasr_hip.so $O/attacker.so $O/in
echo "== the product library (frontend.hip with -fno-slp-vectorize)";           $O/dl_repro $R/viet-asr_amd/lib/libvasr_hip.so $O/attacker.so $O/in
