#!/bin/bash
export TMPDIR=/tmp
F="--steps 40 --warmup 8 --no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time"
run() { STAGE_CW_GRID=$1 timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('CW_GRID=$1', r['ms_per_step'], r['value'])"; }
for i in 1 2; do for g in 256 240 224 192; do run $g; done; done
