#!/bin/bash
# One GPU iteration: K1 parity tests, short bench, kernel trace of 4 steps.  Usage (via gpurun): bash tools/gpu_cycle.sh TAG [full]
TAG=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$2" = "full" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${TAG}_tests.log
else
  timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "str_attn or k1" 2>&1 | tail -5 > gpurun_out/${TAG}_tests.log
fi
timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline > gpurun_out/${TAG}_bench.log 2>&1
rm -rf gpurun_out/${TAG}_prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -- python bench.py --steps 4 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/${TAG}_prof.log 2>&1
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
python tools/rocpd_stats.py $DB 45 > gpurun_out/${TAG}_stats.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_tests.log gpurun_out/${TAG}_bench.log
head -30 gpurun_out/${TAG}_stats.txt
