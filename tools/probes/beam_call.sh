export HSA_ENABLE_IPC_MODE_LEGACY=0; D=$PWD/viet-asr_amd/lib/libvasr_hip_dev.so
(timeout 300 python -m pytest tests/test_beam.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4)
echo "== fuzz wave"; timeout 300 python tests/devtools/fuzz_beam.py ${FUZZ:-400} 0 2>&1 | tail -3
echo "== wave kernel"; VASR_LIB_PATH=$D timeout 200 python tools/probes/beam_lat.py 2>&1 | grep -v amdgpu
if [ -f viet-asr_amd/lib/var_wprof.so ]; then echo "== prof"; VASR_LIB_PATH=$PWD/viet-asr_amd/lib/var_wprof.so ONCE=1 BATCHES=1 WIDTHS=50,100 timeout 100 python tools/probes/beam_lat.py 2>&1 | grep -E "prof" | head -12; fi
