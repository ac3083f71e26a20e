#!/bin/bash
export TMPDIR=/tmp
STAGE_RAGGED_MIN_ROWS=0 TRIALS=300 LEVELS=0 timeout 600 python tools/experiments/step_repeat_small.py 2>&1 | grep -E "repeats that differ" | cut -c1-60
STAGE_RAGGED_MIN_ROWS=0 NOSYNC=1 TRIALS=300 timeout 600 python tools/experiments/step_repeat_small.py 2>&1 | grep -E "repeats that differ" | cut -c1-60
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest "tests/test_hip_stage.py::test_branch_streams_change_nothing_but_the_schedule" -x -q -k "True-groups" 2>&1 | grep -E "passed|failed" ; done
