#!/bin/bash
export TMPDIR=/tmp STAGE_RAGGED_MIN_ROWS=0
timeout 600 python -m pytest tests/test_hip_cat3_dw.py -x -q 2>&1 | tail -1
echo "heads, mixed levels"; HEADS=1 TRIALS=3000 timeout 900 python tools/experiments/step_repeat_small.py 2>&1 | grep -E "repeats that differ" | cut -c1-50
echo "groups, mixed levels"; TRIALS=3000 timeout 900 python tools/experiments/step_repeat_small.py 2>&1 | grep -E "repeats that differ" | cut -c1-50
echo "groups, level 0"; LEVELS=0 TRIALS=1500 timeout 900 python tools/experiments/step_repeat_small.py 2>&1 | grep -E "repeats that differ" | cut -c1-50
REP=300 python tools/cat3_fused_time.py 2>&1 | grep "backward with dW"; REP=1 python tools/cat3_fused_time.py 2>&1 | grep "backward with dW"
