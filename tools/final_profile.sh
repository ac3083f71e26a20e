#!/bin/bash
# Round profile: kernel trace of the default bench command + PMC passes of the K1 forward kernel.  bash tools/final_profile.sh TAG
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${TAG}_prof
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -- python bench.py --no_cpu_baseline > gpurun_out/${TAG}_prof.log 2>&1
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
python tools/rocpd_stats.py $DB 60 > gpurun_out/${TAG}_kernel_trace_stats.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
bash tools/pmc_run.sh ${TAG}_k1 str_attn_fwd_reg python bench.py --only_roofline > /dev/null 2>&1
cat gpurun_out/${TAG}_bench.json
