#!/bin/bash
export TMPDIR=/tmp
F="--steps 6 --warmup 3 --no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time"
for v in 0 1 0 1; do
STAGE_GEMM_F32=1 STAGE_CAT3_DW=$v timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('exact DW=$v', r['ms_per_step'], r['value'])"
done
