#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_cat3_dw.py -x -q 2>&1 | tail -30 > gpurun_out/r6_c_dw_tests.txt
tail -5 gpurun_out/r6_c_dw_tests.txt
REP=300 timeout 300 python tools/cat3_fused_time.py 2>&1 | tail -12
REP=1 timeout 300 python tools/cat3_fused_time.py 2>&1 | tail -12
