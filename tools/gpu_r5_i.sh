#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5i_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5i_bench.json 2> gpurun_out/r5i_bench.err
timeout 500 bash tools/pmc_run.sh r05_k1_long_fwd str_attn_long_fwd python bench.py --config stress --only_roofline > /dev/null 2>&1
cat gpurun_out/r5i_tests.log; tail -c 600 gpurun_out/r5i_bench.json
