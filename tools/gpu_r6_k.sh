#!/bin/bash
# steady-state idle gaps of the multi-stream step: kernel trace of 12 steps, last ~3 steps listed
export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/tl2_prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl2_prof -- python bench.py --steps 12 --warmup 3 --no_cpu_baseline --no_roofline --no_children --no_pmc --no_device_time > gpurun_out/tl2_prof.log 2>&1
DB=$(find gpurun_out/tl2_prof -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB 900 > gpurun_out/r06_timeline3.txt 2>&1
rm -rf gpurun_out/tl2_prof
tail -1 gpurun_out/tl2_prof.log | cut -c1-150
tail -1 gpurun_out/r06_timeline3.txt
