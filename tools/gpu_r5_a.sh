#!/bin/bash
# round 5, batch A: new tests first, then the full GPU suite, bench, kernel trace with the reduction launches listed, host profile
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dropout_parity.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r5a_newtests.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5a_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no_children --no_cpu_baseline > gpurun_out/r5a_bench.log 2>&1
TRACE_PAT=col bash tools/trace_bench.sh r5a
BSZ=2 timeout 200 python tools/host_profile.py 20 > gpurun_out/r5a_hostprof_b2.log 2>&1
timeout 200 python tools/host_profile.py 10 > gpurun_out/r5a_hostprof_b16.log 2>&1
cat gpurun_out/r5a_newtests.log gpurun_out/r5a_tests.log
