#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for abl in $ABLS; do
  echo "== CW_ABL=$abl"; LIB=tvqaplus_amd/libstage_hip_abl$abl.so REP=${REP:-300} timeout 300 python tools/cat3_fused_time.py 2>&1 | grep "backward with dW inside"
done
