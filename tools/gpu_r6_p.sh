#!/bin/bash
export TMPDIR=/tmp
for v in 1 0; do
  f=0; n=0
  for i in $(seq 1 25); do
    out=$(STAGE_CAT3_DW=$v timeout 300 python -m pytest tests/test_hip_stage.py -q -k "repeats_bit_for_bit or (branch_streams and True)" 2>&1 | grep -E "passed|failed|^FAILED" | tr '\n' ' ')
    n=$((n+1)); case "$out" in *failed*) f=$((f+1)); echo "DW=$v run $i: $out";; esac
  done
  echo "== STAGE_CAT3_DW=$v: $f failing processes of $n"
done
