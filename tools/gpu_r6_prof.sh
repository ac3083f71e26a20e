#!/bin/bash
# round 6 profile set -> gpurun_out/r06_*
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python tools/step_traffic.py > gpurun_out/r06_step_traffic.txt 2> gpurun_out/r06_step_traffic.err; head -30 gpurun_out/r06_step_traffic.txt
timeout 300 bash tools/trace_bench.sh r06_bench --no_pmc > /dev/null 2>&1
TRACE_PAT=colreduce timeout 300 bash tools/trace_bench.sh r06_one_stream_bench --no_pmc --streams 0 > /dev/null 2>&1
timeout 500 bash tools/pmc_run.sh r06_cat3_instep anonymous python bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_pmc --no_children --no_roofline --no_device_time > /dev/null 2>&1
timeout 500 bash tools/pmc_run.sh r06_k1_instep str_attn python bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_pmc --no_children --no_roofline --no_device_time > /dev/null 2>&1
timeout 400 bash tools/pmc_run.sh r06_cat3_dw_rep cw_bwd python tools/cat3_fused_time.py > /dev/null 2>&1
REP=1 timeout 400 bash tools/pmc_run.sh r06_cat3_dw_flat cw_bwd python tools/cat3_fused_time.py > /dev/null 2>&1
ls -la gpurun_out | grep r06
