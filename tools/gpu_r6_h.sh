#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_cat3_dw.py -x -q 2>&1 | tail -3
for rep in 300 1; do
REP=$rep timeout 300 python tools/cat3_fused_time.py 2>&1 | grep -E "backward with dW"
CW_PROF=1 LIB=tvqaplus_amd/libstage_hip_prof.so REP=$rep timeout 300 python tools/cat3_fused_time.py 2>&1 | grep -E "wave"
done
