#!/bin/bash
# Kernel timeline of the last bench step.  Usage (via gpurun): bash tools/timeline.sh TAG [n_last]
TAG=${1:-tl}
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_children --no_pmc --no_device_time > gpurun_out/${TAG}_prof.log 2>&1
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB ${2:-700} > gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
tail -3 gpurun_out/${TAG}_timeline.txt
