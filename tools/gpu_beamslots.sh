#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
for s in 512 1024 2048 auto; do
  [ $s = auto ] && unset VASR_BEAM_SLOTS || export VASR_BEAM_SLOTS=$s
  python tools/probes/beam_slots.py 2>&1 | grep -v amdgpu | tail -1
done
unset VASR_BEAM_SLOTS VASR_LIB_PATH
timeout 900 python -m pytest tests/test_beam.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
