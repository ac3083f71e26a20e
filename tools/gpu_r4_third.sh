#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ragged.py -x -q -m gpu > gpurun_out/r4_ragged_tests.log 2>&1; echo "ragged tests rc=$?"
tail -4 gpurun_out/r4_ragged_tests.log
timeout 900 python -m pytest tests/test_hip_stage.py -x -q -m gpu -k "full_size or full_length or trajectory" > gpurun_out/r4_stage_tests.log 2>&1; echo "stage tests rc=$?"
tail -3 gpurun_out/r4_stage_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_pmc --dump_steps > gpurun_out/r4_bench_ragged.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4_bench_ragged.log") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["host_issue_ms_per_step"], d["host_wait_ms_per_step"], d.get("device_ms_per_step"), d.get("launches_per_step"))
print(d.get("step_ms_all"))
for k in d:
    if k.startswith("roofline"):
        r=d[k]; print(k,{kk:r.get(kk) for kk in ("avg_us","frac","isolated_avg_us")})
PY
