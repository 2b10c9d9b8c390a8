#!/bin/bash
# Kill-criterion probe for VERDICT r05 item 8 (batch-1 acoustic pass with the depthwise computed inside the latency GEMM's staging
# wavefronts).  Builds viet-asr_amd/lib/var_b1fill<K>.so: libvasr_hip_dev.so with ONE change -- every staging thread of
# pw_gemm_latency_kernel issues, per item (8 k rows x 4 columns) and before converting it, the packed FMAs a K-tap depthwise
# producer would need for those 32 outputs (32 * K / 2 v_pk_fma_f32 on register operands: no window loads, no LDS traffic, no
# masking -- the cheapest such a producer could be).  Results stay correct (the filler's sum is never stored).  Then
#   VASR_LIB_PATH=.../var_b1fill51.so python tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam
# under rocprofv3 --kernel-trace --stats gives the latency GEMM's duration WITH the producer's issue load; compare with the
# shipped GEMM + dw_toeplitz_kernel durations of the same call (profiles/r06_b1_fused_probe.txt).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; K=${1:-51}
T=$(mktemp -d); mkdir -p $T/viet-asr_amd; cp -r $R/include $T/include; cp -r $R/viet-asr_amd/csrc $T/viet-asr_amd/csrc
python3 - "$T/viet-asr_amd/csrc/encoder_pw_lat.hip" "$K" <<'PY'
import sys
p, K = sys.argv[1], int(sys.argv[2])
s = open(p).read()
mark = "#pragma unroll\n    for (int i = 0; i < PI; ++i) {\n      const int g = g0 + 32 * i;\n"
assert s.count(mark) == 1
fill = f"""    v2f fa[8], fx[8], fw[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {{ fa[j] = v2f{{0.f, 0.f}}; fx[j] = v2f{{(float)(tid + j), 1e-3f * lane}}; }}
#pragma unroll
    for (int k = 0; k < 4; ++k) fw[k] = v2f{{1e-3f * (k + 1), 1e-4f * tid}};
""" + mark + f"""      for (int it = 0; it < {32 * K // 2 // 8}; ++it) {{
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(fa[j]) : "v"(fw[j & 3]), "v"(fx[j]));
      }}
"""
s = s.replace(mark, fill)
tail = "    __syncthreads();     // (the multipliers' epilogue reuses the image)\n    return;\n"
assert s.count(tail) == 1
s = s.replace(tail, "    { float fs = 0.f;\n#pragma unroll\n      for (int j = 0; j < 8; ++j) fs += fa[j].x + fa[j].y;\n      if (fs == 12345.678f) a.y[0] = fs; }\n" + tail)
open(p, "w").write(s)
PY
make -C $T/viet-asr_amd/csrc -j8 $T/viet-asr_amd/lib/libvasr_hip_dev.so > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $T/viet-asr_amd/lib/libvasr_hip_dev.so $R/viet-asr_amd/lib/var_b1fill$K.so
echo "built viet-asr_amd/lib/var_b1fill$K.so"
