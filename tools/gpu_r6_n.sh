#!/bin/bash
export TMPDIR=/tmp
F="--steps 40 --warmup 8 --no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time"
run() { timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('late=$STAGE_LATE_LOSSES', r['ms_per_step'], r['value'], r['host_issue_ms_per_step'])"; }
for i in 1 2 3 4; do STAGE_LATE_LOSSES=1 run; unset STAGE_LATE_LOSSES; run; done
