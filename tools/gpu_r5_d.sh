#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python bench.py --bsz 2 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time 2>&1 | tail -1 | cut -c1-300 > gpurun_out/r5d_b2.log
timeout 200 python bench.py --bsz 2 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time --clip torch 2>&1 | tail -1 | cut -c1-300 > gpurun_out/r5d_b2_torchclip.log
timeout 200 python bench.py --bsz 4 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time 2>&1 | tail -1 | cut -c1-300 > gpurun_out/r5d_b4.log
STAGE_RAGGED_MIN_ROWS=0 timeout 200 python bench.py --bsz 4 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time 2>&1 | tail -1 | cut -c1-300 > gpurun_out/r5d_b4_rag.log
BSZ=2 timeout 200 python tools/step_host_timeline.py > gpurun_out/r5d_tl_b2.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time 2>&1 | tail -1 | cut -c1-300 > gpurun_out/r5d_default.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5d_tests.log
cat gpurun_out/r5d_*.log
