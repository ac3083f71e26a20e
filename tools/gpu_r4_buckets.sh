#!/bin/bash
# round 4: context length buckets + long-row block skipping (stress config) -- tests + stress bench with / without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_hip_ragged.py tests/test_hip_ops.py tests/test_hip_bf16.py tests/test_hip_stage.py -x -q -m gpu -k "bucket or long or bf16 or stress" 2>&1 | tail -15 > gpurun_out/r4_bucket_tests.log
python bench.py --config stress --steps 5 --warmup 2 --no_cpu_baseline --no_pmc --no_children > gpurun_out/r4_stress_buckets.json 2> gpurun_out/r4_stress_buckets.err
STAGE_NO_CTX_BUCKETS=1 python bench.py --config stress --steps 5 --warmup 2 --no_cpu_baseline --no_pmc --no_children --no_roofline > gpurun_out/r4_stress_nobuckets.json 2>> gpurun_out/r4_stress_buckets.err
tail -5 gpurun_out/r4_bucket_tests.log
python - <<'P'
import json
for f in ("r4_stress_buckets", "r4_stress_nobuckets"):
    try:
        for line in open("gpurun_out/%s.json" % f):
            if line.startswith("{"):
                d = json.loads(line); print(f, d["ms_per_step"], d["value"], d["config"].get("peak_hbm_gib"), d.get("launches_per_step"), d.get("device_ms_per_step"), d.get("host_issue_ms_per_step"), json.dumps(d.get("roofline"))[:300])
    except Exception as e:
        print(f, "failed", e)
P
tail -5 gpurun_out/r4_stress_buckets.err
