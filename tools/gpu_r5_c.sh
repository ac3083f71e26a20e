#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
BSZ=2 timeout 200 python tools/step_host_timeline.py > gpurun_out/r5c_tl_b2.log 2>&1
BSZ=2 STAGE_NO_RAGGED=1 timeout 200 python tools/step_host_timeline.py > gpurun_out/r5c_tl_b2_dense.log 2>&1
for s in 0 2 4; do timeout 200 python bench.py --bsz 2 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time --streams $s 2>&1 | tail -1 | cut -c1-420 > gpurun_out/r5c_b2_s$s.log; done
STAGE_NO_RAGGED=1 timeout 200 python bench.py --bsz 2 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time 2>&1 | tail -1 | cut -c1-420 > gpurun_out/r5c_b2_dense.log
timeout 300 python bench.py --steps 20 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time --reference_batch 2>&1 | tail -1 | cut -c1-600 > gpurun_out/r5c_refbatch.log
timeout 300 python bench.py --steps 20 --warmup 5 --no_children --no_cpu_baseline --no_roofline --no_device_time 2>&1 | tail -1 | cut -c1-600 > gpurun_out/r5c_default.log
cat gpurun_out/r5c_tl_b2.log gpurun_out/r5c_tl_b2_dense.log gpurun_out/r5c_b2_s*.log gpurun_out/r5c_b2_dense.log gpurun_out/r5c_refbatch.log gpurun_out/r5c_default.log
