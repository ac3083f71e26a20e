#!/bin/bash
# Round profile set -> gpurun_out/${TAG}_*: bench line, kernel trace of the bench command, PMC passes of the K1 forward (video and
# subtitle shapes), of the fused K1 backward and of the 960000 x 384 -> 128 GEMMs (forward, weight gradient).  bash tools/round_profile.sh r02
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --no_cpu_baseline --dense > gpurun_out/${TAG}_bench_line_dense.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 bash tools/trace_bench.sh ${TAG}_bench > /dev/null 2>&1
# K1 forward PMC: bench.py --only_roofline launches the video shape then the subtitle shape
timeout 400 bash tools/pmc_run.sh ${TAG}_k1_fwd str_attn_fwd python bench.py --only_roofline > /dev/null 2>&1
timeout 400 bash tools/pmc_run.sh ${TAG}_k1_bwd_vid str_attn_bwd_fused python tools/k1_bwd_times.py > /dev/null 2>&1
LR=50 timeout 400 bash tools/pmc_run.sh ${TAG}_k1_bwd_sub str_attn_bwd_fused python tools/k1_bwd_times.py > /dev/null 2>&1
timeout 400 bash tools/pmc_run.sh ${TAG}_gemm_nt gemm_nt_stream python tools/gemm_one.py 960000 128 384 nt > /dev/null 2>&1
timeout 400 bash tools/pmc_run.sh ${TAG}_gemm_tn gemm_tn_quad python tools/gemm_one.py 960000 128 384 tn > /dev/null 2>&1
# round 3: the fused LayerNorm([a,b,a*b]) + Linear kernels (csrc/cat3_fused.hip): c2q shape (broadcast a) and concat_fc shape
timeout 400 bash tools/pmc_run.sh ${TAG}_cat3_fused_rep cf python tools/cat3_fused_time.py > /dev/null 2>&1
REP=1 timeout 400 bash tools/pmc_run.sh ${TAG}_cat3_fused_flat cf python tools/cat3_fused_time.py > /dev/null 2>&1
timeout 200 python tools/cat3_fused_time.py > gpurun_out/${TAG}_cat3_fused_times_rep.txt 2>&1
REP=1 timeout 200 python tools/cat3_fused_time.py > gpurun_out/${TAG}_cat3_fused_times_flat.txt 2>&1
timeout 300 python bench.py --config stress --steps 5 --warmup 2 --no_cpu_baseline > gpurun_out/${TAG}_bench_line_stress.json 2>> gpurun_out/${TAG}_bench.err
timeout 200 python tools/k1_bwd_times.py > gpurun_out/${TAG}_k1_bwd_times_vid.txt 2>&1
LR=50 timeout 200 python tools/k1_bwd_times.py > gpurun_out/${TAG}_k1_bwd_times_sub.txt 2>&1
ls gpurun_out | grep ${TAG}
