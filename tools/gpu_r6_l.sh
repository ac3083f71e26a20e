#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_groups.py -x -q -k "reference_loss" 2>&1 | tail -3
F="--steps 40 --warmup 8 --no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time"
run() { timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', r['ms_per_step'], r['value'], r['config'].get('final_loss'))"; }
for i in 1 2 3; do run --loss eager; run --loss fused; done
