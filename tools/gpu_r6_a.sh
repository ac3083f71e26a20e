#!/bin/bash
# round 6, first contact of csrc/cat3_bwd_dw.hip with the hardware: its own tests, per-launch times, then the group / ragged suites
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_cat3_dw.py -x -q 2>&1 | tail -40 > gpurun_out/r6_a_dw_tests.txt
tail -5 gpurun_out/r6_a_dw_tests.txt
REP=300 timeout 300 python tools/cat3_fused_time.py > gpurun_out/r6_a_time_rep300.txt 2>&1; tail -14 gpurun_out/r6_a_time_rep300.txt
REP=1 timeout 300 python tools/cat3_fused_time.py > gpurun_out/r6_a_time_rep1.txt 2>&1; tail -14 gpurun_out/r6_a_time_rep1.txt
