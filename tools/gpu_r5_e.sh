#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_stage.py tests/test_hip_bf16.py tests/test_param_gate.py tests/test_hip_groups.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5e_tests.log
timeout 300 python bench.py --config stress --steps 3 --warmup 2 --no_children --no_cpu_baseline --no_device_time 2>&1 | tail -1 | cut -c1-1500 > gpurun_out/r5e_stress.log
STAGE_NO_PARAM_GATE=1 timeout 300 python bench.py --config stress --steps 3 --warmup 2 --no_children --no_cpu_baseline --no_device_time --no_roofline 2>&1 | tail -1 | cut -c1-300 > gpurun_out/r5e_stress_nogate.log
timeout 300 python bench.py --heads 4 --steps 10 --warmup 3 --no_children --no_cpu_baseline --no_roofline 2>&1 | tail -1 | cut -c1-1200 > gpurun_out/r5e_heads4.log
cat gpurun_out/r5e_*.log
