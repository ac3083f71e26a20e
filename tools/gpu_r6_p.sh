#!/bin/bash
export TMPDIR=/tmp
f=0; n=0
for i in $(seq 1 30); do
  out=$(timeout 300 python -m pytest tests/test_hip_stage.py -q -k "repeats_bit_for_bit or (branch_streams and True)" 2>&1 | grep -E "passed|failed|^FAILED" | tr '\n' ' ')
  n=$((n+1)); case "$out" in *failed*) f=$((f+1)); echo "run $i: $out";; esac
done
echo "== default (dW inside on): $f failing processes of $n"
