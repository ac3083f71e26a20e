#!/bin/bash
# short in-step A/B of the [a,b,a*b] backward with dW inside (STAGE_CAT3_DW=1) vs the default
mkdir -p gpurun_out; export TMPDIR=/tmp
F="--steps 40 --warmup 8 --no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time"
run() { STAGE_CAT3_DW=$1 timeout 300 python bench.py $F "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('DW=$1 ${*:2}', r['ms_per_step'], r['value'])"; }
for i in 1 2 3; do for v in 0 1; do run $v; done; done
