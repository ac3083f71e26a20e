#!/bin/bash
# round 6: where the time of cw_bwd_kernel goes: PMC passes on the per-launch timing script + ablation builds
mkdir -p gpurun_out; export TMPDIR=/tmp
REP=300 bash tools/pmc_run.sh r6_b_cw_rep300 cw_bwd_kernel python tools/cat3_fused_time.py > /dev/null 2>&1
cat gpurun_out/pmc_r6_b_cw_rep300.txt | cut -c1-140
for abl in 1 2 4 8 16 31; do
  
  echo "== CW_ABL=$abl"; LIB=tvqaplus_amd/libstage_hip_abl$abl.so REP=300 timeout 300 python tools/cat3_fused_time.py 2>&1 | grep "backward with dW inside"
done
