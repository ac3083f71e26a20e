#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
BSZ=2 timeout 200 python tools/host_calls.py 30 > gpurun_out/r5b_calls_b2.log 2>&1
BSZ=2 STREAMS=0 timeout 200 python tools/host_calls.py 30 > gpurun_out/r5b_calls_b2_s0.log 2>&1
BSZ=16 timeout 200 python tools/host_calls.py 20 > gpurun_out/r5b_calls_b16.log 2>&1
TRACE_PAT=col bash tools/trace_bench.sh r5b_s0 --streams 0
STAGE_NO_COLFOLD=1 TRACE_PAT=col bash tools/trace_bench.sh r5b_s0_nofold --streams 0
timeout 900 python -m pytest tests/test_parallel_hip.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5b_tests.log
cat gpurun_out/r5b_calls_b2.log gpurun_out/r5b_calls_b2_s0.log gpurun_out/r5b_tests.log
