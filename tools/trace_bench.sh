#!/bin/bash
# Kernel trace of a bench command -> per-kernel table.  bash tools/trace_bench.sh TAG [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -- python bench.py --no_cpu_baseline --no_roofline --no_children --steps 10 --warmup 3 "$@" > gpurun_out/${TAG}_prof.log 2>&1
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
python tools/rocpd_stats.py $DB 70 $TRACE_PAT > gpurun_out/${TAG}_kernel_trace_stats.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
tail -1 gpurun_out/${TAG}_prof.log | cut -c1-200
