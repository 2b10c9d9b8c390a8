#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/libvasr_hip_dev.so
MODE=base python tools/probes/beam_group.py 2>&1 | grep -v amdgpu | tail -1
export VASR_LIB_PATH=$R/viet-asr_amd/lib/var_t512s1024.so
for m in base group group_nomask half; do MODE=$m python tools/probes/beam_group.py 2>&1 | grep -v amdgpu | tail -1; done
MODE=half NCU=32 python tools/probes/beam_group.py 2>&1 | grep -v amdgpu | tail -1
MODE=group NCU=128 python tools/probes/beam_group.py 2>&1 | grep -v amdgpu | tail -1
