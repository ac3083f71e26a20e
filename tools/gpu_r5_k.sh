#!/bin/bash
# refresh of the round-5 profile lines that changed after the first set (stream level 3 default, fused QKV): bench trace, heads-4 line + trace
TAG=r05; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 bash tools/trace_bench.sh ${TAG}_bench --no_children --no_pmc > /dev/null 2>&1
timeout 300 python bench.py --no_cpu_baseline --no_children --no_pmc --no_roofline --heads 4 > gpurun_out/${TAG}_bench_line_heads4.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 bash tools/trace_bench.sh ${TAG}_heads4_bench --heads 4 --no_children --no_pmc > /dev/null 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no_children --no_cpu_baseline --no_roofline --reference_batch > gpurun_out/${TAG}_bench_line_reference_batch.json 2>> gpurun_out/${TAG}_bench.err
timeout 200 python bench.py --bsz 2 --steps 30 --warmup 5 --no_children --no_cpu_baseline --no_roofline > gpurun_out/${TAG}_bench_line_bsz2.json 2>> gpurun_out/${TAG}_bench.err
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/${TAG}_gpu_tests_tail.txt 2>&1
tail -4 gpurun_out/${TAG}_gpu_tests_tail.txt
