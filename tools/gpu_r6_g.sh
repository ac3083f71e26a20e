#!/bin/bash
# per-kernel in-step durations, dW-inside vs default
mkdir -p gpurun_out; export TMPDIR=/tmp
F="--steps 20 --warmup 5 --no_children --no_roofline --no_cpu_baseline --no_pmc --no_device_time"
for v in 0 1; do
  rm -rf /tmp/kt$v
  STAGE_CAT3_DW=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$v -o kt -- python bench.py $F > /tmp/kt$v.log 2>&1
  f=$(find /tmp/kt$v -name '*kernel_stats.csv' | head -1)
  echo "== DW=$v  ($f)"
  if [ -n "$f" ]; then cp "$f" gpurun_out/r6_g_kernel_stats_dw$v.csv; head -40 "$f" | cut -c1-220; else tail -5 /tmp/kt$v.log; fi
done
