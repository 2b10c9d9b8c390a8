#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VASR_LIB_PATH=$R/viet-asr_amd/lib/var_prof.so
python tools/probes/beam_slots.py 2>&1 | grep -v amdgpu | grep "beam prof" | awk 'NR==61||NR==93||NR==96||NR==77||NR==29'
