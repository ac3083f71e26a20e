#!/bin/bash
# in-step A/B of the [a,b,a*b] backward with dW inside (STAGE_CAT3_DW=1) vs the default
mkdir -p gpurun_out; export TMPDIR=/tmp
F="--steps 30 --warmup 8 --no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time"
for i in 1 2; do
  for v in 0 1; do
    STAGE_CAT3_DW=$v timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('DW=$v', r['ms_per_step'], r['value'])"
  done
done
