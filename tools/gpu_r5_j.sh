#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "mha" 2>&1 | tail -4 > gpurun_out/r5j_tests_ops.log
timeout 1500 python -m pytest tests/test_hip_stage.py tests/test_hip_bf16.py tests/test_hip_ragged.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r5j_tests_model.log
timeout 300 python bench.py --heads 4 --steps 10 --warmup 3 --no_children --no_cpu_baseline --no_roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('heads4 fused', d['ms_per_step'], d['launches_per_step'], d['device_ms_per_step'])" > gpurun_out/r5j_heads4.log
STAGE_NO_FUSED_QKV=1 timeout 300 python bench.py --heads 4 --steps 10 --warmup 3 --no_children --no_cpu_baseline --no_roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('heads4 separate', d['ms_per_step'], d['launches_per_step'], d['device_ms_per_step'])" >> gpurun_out/r5j_heads4.log
cat gpurun_out/r5j_*.log
