#!/bin/bash
export TMPDIR=/tmp
for v in 1 0; do
  echo "== STAGE_CAT3_DW=$v"
  for i in 1 2 3 4 5 6 7 8; do
    STAGE_CAT3_DW=$v timeout 300 python -m pytest "tests/test_hip_stage.py::test_branch_streams_change_nothing_but_the_schedule" -x -q -k "True-groups" 2>&1 | grep -E "passed|failed|assert torch.equal|AssertionError|^E  " | head -4
  done
done
