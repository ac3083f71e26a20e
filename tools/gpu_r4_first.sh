#!/bin/bash
# round 4, first GPU session: ragged-row tests, the full-size parity tests, one bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ragged.py -x -q -m gpu > gpurun_out/r4_ragged_tests.log 2>&1; echo "ragged tests rc=$?" | tee -a gpurun_out/r4_summary.log
timeout 1200 python -m pytest tests/test_hip_stage.py -x -q -m gpu -k "full_size or full_length or mid_ or oracle_fresh" > gpurun_out/r4_stage_tests.log 2>&1; echo "stage tests rc=$?" | tee -a gpurun_out/r4_summary.log
timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_pmc > gpurun_out/r4_bench_ragged.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/r4_summary.log
STAGE_NO_RAGGED=1 timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_pmc --no_roofline > gpurun_out/r4_bench_dense_layout.log 2>&1; echo "bench dense-layout rc=$?" | tee -a gpurun_out/r4_summary.log
tail -3 gpurun_out/r4_ragged_tests.log; tail -3 gpurun_out/r4_stage_tests.log; tail -1 gpurun_out/r4_bench_ragged.log | cut -c1-600; tail -1 gpurun_out/r4_bench_dense_layout.log | cut -c1-300
