#!/bin/bash
# Round profile set -> gpurun_out/${TAG}_*: bench line, kernel trace of the bench command, PMC passes of the K1 forward (video and
# subtitle shapes), of the fused K1 backward and of the 960000 x 384 -> 128 GEMMs (forward, weight gradient).  bash tools/round_profile.sh r02
TAG=${1:-r05}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --no_cpu_baseline --no_children --dense > gpurun_out/${TAG}_bench_line_dense.json 2>> gpurun_out/${TAG}_bench.err
# round 4: the same step with every padded row computed (STAGE_NO_RAGGED=1), and the ragged kernels' counters inside the step
STAGE_NO_RAGGED=1 timeout 300 python bench.py --no_cpu_baseline --no_children --no_pmc --no_roofline > gpurun_out/${TAG}_bench_line_dense_rows.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --no_cpu_baseline --no_children --no_pmc --no_roofline --heads 4 > gpurun_out/${TAG}_bench_line_heads4.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 bash tools/trace_bench.sh ${TAG}_heads4_bench --heads 4 --no_children --no_pmc > /dev/null 2>&1
timeout 500 bash tools/pmc_run.sh ${TAG}_cat3_ragged_instep cf python bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_pmc --no_children --no_roofline --no_device_time > /dev/null 2>&1
timeout 500 bash tools/pmc_run.sh ${TAG}_k1_instep str_attn python bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_pmc --no_children --no_roofline --no_device_time > /dev/null 2>&1
timeout 300 bash tools/trace_bench.sh ${TAG}_bench --no_children --no_pmc > /dev/null 2>&1
# round 5: the same trace on ONE stream (--streams 0): kernel durations that are statements about the kernels -- with branch streams a
# small kernel's duration includes waiting for compute units a neighbour holds (the column reductions: 0.33 ms/step here, 1.6 in the default trace)
TRACE_PAT=colreduce timeout 300 bash tools/trace_bench.sh ${TAG}_one_stream_bench --no_children --no_pmc --streams 0 > /dev/null 2>&1
timeout 200 python bench.py --bsz 2 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline > gpurun_out/${TAG}_bench_line_bsz2.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no_children --no_cpu_baseline --no_roofline --reference_batch > gpurun_out/${TAG}_bench_line_reference_batch.json 2>> gpurun_out/${TAG}_bench.err
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
timeout 300 python bench.py --config stress --steps 5 --warmup 2 --no_cpu_baseline --no_children > gpurun_out/${TAG}_bench_line_stress.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 bash tools/trace_bench.sh ${TAG}_stress --config stress --no_children --no_pmc --steps 5 --warmup 2 > /dev/null 2>&1
timeout 500 bash tools/pmc_run.sh ${TAG}_k1_long_fwd str_attn_long_fwd python bench.py --config stress --only_roofline > /dev/null 2>&1
timeout 200 python tools/k1_bwd_times.py > gpurun_out/${TAG}_k1_bwd_times_vid.txt 2>&1
LR=50 timeout 200 python tools/k1_bwd_times.py > gpurun_out/${TAG}_k1_bwd_times_sub.txt 2>&1
ls gpurun_out | grep ${TAG}
