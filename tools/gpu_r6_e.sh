#!/bin/bash
# round 6: full GPU suite + the driver's bench command line
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_gpu_tests_tail.txt
cat gpurun_out/r06_gpu_tests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/r06_gpu_tests_tail.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r06_bench_line.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","host_issue_ms_per_step","device_ms_per_step","launches_per_step") if k in r})
for k in ("roofline","roofline_dense","roofline_sub","roofline_bwd","roofline_sub_bwd"):
    if k in r: print(k, {x:r[k].get(x) for x in ("frac","avg_us","algorithmic_bytes","traffic","traffic_over_algorithmic")})
print("step_roofline", r.get("step_roofline"))
for k in ("heads4","cat3_dw_off","eager_loss","strong_n2_sim","one_stream","exact_f32","dense","reference_batch","stress"):
    v=r.get(k); print(k, {x:v.get(x) for x in ("ms_per_step","value","host_issue_ms_per_step","predicted_8gpu_value","error")} if isinstance(v,dict) else v)
print("cpu", r.get("cpu_baseline"))
PY
