#!/bin/bash
export TMPDIR=/tmp
F="--no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time --dump_steps"
run() { timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', r['config'].get('final_loss'))"; }
for st in "1 0" "2 0" "3 1" "6 2"; do set -- $st; run --loss eager --steps $1 --warmup $2; run --loss fused --steps $1 --warmup $2; done
