#!/bin/bash
# Is the fix a matter of code LAYOUT?  The product build (frontend.hip without packed-FP32 instructions) with its code object perturbed -- a dummy
# kernel of 1 / 7 / 40 KB of straight-line code in front of the STFT kernel -- and the vulnerable build (SLP on) perturbed the same way.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=${TMPDIR:-/tmp}/vasr_dl_layout; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/probes/mfma_attacker.hip -o $O/attacker.so 2>/dev/null || exit 1
g++ -O2 -std=c++17 -w -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ $R/tools/probes/dl_repro.cpp -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o $O/dl_repro || exit 1
python $R/tools/probes/stft_mfma_repro_dump.py $O/in > /dev/null 2>&1
for pad in 60 400 2400; do
  S=$O/src_$pad; rm -rf $S; mkdir -p $S/viet-asr_amd; cp -r $R/include $S/include; cp -r $R/viet-asr_amd/csrc $S/viet-asr_amd/csrc
  python3 - $S/viet-asr_amd/csrc/frontend.hip $pad <<'PY'
import sys
p, n = sys.argv[1], int(sys.argv[2])
s = open(p).read()
body = "\n".join(f"  v = v * 1.0001f + {i}.5f; if (v == 12345.0f) out[{i}] = v;" for i in range(n))
dummy = "__global__ void layout_pad_kernel(float* out, float v) {\n" + body + "\n}\n\n"
mark = "template <int kFramesPerBlock, typename S>\n__global__ __launch_bounds__(kThreads, 4) void stft_logmel_kernel"
assert mark in s
open(p, "w").write(s.replace(mark, dummy + mark, 1))
PY
  for slp in 0 1; do
    X="DWFLAGS_frontend.hip=-fno-slp-vectorize"; [ $slp = 1 ] && X="DWFLAGS_frontend.hip="
    make -C $S/viet-asr_amd/csrc -j16 "$X" OUT=$O/lib_${pad}_$slp OBJ=$O/obj_${pad}_$slp OBJD=$O/objd_${pad}_$slp $O/lib_${pad}_$slp/libvasr_hip.so > $O/build.log 2>&1 || { tail -3 $O/build.log; exit 1; }
    echo "== dummy kernel of $pad statements in front of the STFT kernel, packed-FP32 instructions: $([ $slp = 1 ] && echo yes || echo no)"
    $O/dl_repro $O/lib_${pad}_$slp/libvasr_hip.so $O/attacker.so $O/in | grep attacker
  done
done
