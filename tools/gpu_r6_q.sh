#!/bin/bash
export TMPDIR=/tmp STAGE_RAGGED_MIN_ROWS=0
timeout 600 python -m pytest tests/test_hip_cat3_dw.py -x -q 2>&1 | tail -1
HEADS=1 TRIALS=10000 timeout 1200 python tools/experiments/step_repeat_small.py 2>&1 | grep -E "  step|repeats that differ" | cut -c1-120
TRIALS=10000 timeout 1200 python tools/experiments/step_repeat_small.py 2>&1 | grep -E "  step|repeats that differ" | cut -c1-120
