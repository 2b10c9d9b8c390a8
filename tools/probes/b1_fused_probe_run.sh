O=gpurun_out/r6g; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in shipped var_b1fill51 var_b1fill75; do
  if [ $v = shipped ]; then L=$R/viet-asr_amd/lib/libvasr_hip_dev.so; else L=$R/viet-asr_amd/lib/$v.so; fi
  echo "#### $v"
  VASR_LIB_PATH=$L python $R/tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam --calls 60 2>/dev/null
  VASR_LIB_PATH=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/st_$v -- python $R/tools/b1_serving.py --model quartznet15x5 --seconds 10 --no-beam --calls 60 > /dev/null 2>&1
  f=$(find $R/$O/st_$v -name '*kernel_stats.csv' | head -1)
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].replace("vasr::(anonymous namespace)::", "").replace("void ", "")
    if any(k in n for k in ("pw_gemm_latency", "dw_toeplitz", "dwpw_fused", "pw_gemm_split", "dw_conv")):
        print("  %-70s calls %6s avg %8.2f us total %9.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
  find $R/$O -name '*kernel_trace.csv' -delete; find $R/$O -name '*.db' -delete
done
