#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver_cmd.log 2> gpurun_out/r4_bench_driver_cmd.err; echo "rc=$? elapsed $(( $(date +%s) - t0 )) s"
tail -3 gpurun_out/r4_bench_driver_cmd.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4_bench_driver_cmd.log") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d.get("device_ms_per_step"))
for k in ("exact_f32","dense","stress","cpu_baseline"):
    print(k, json.dumps(d.get(k))[:500])
for k in d:
    if k.startswith("roofline"):
        r=d[k]; print(k,{kk:r.get(kk) for kk in ("avg_us","frac","timed","traffic","frac_this_layout","frac_of_f16_peak")})
print(d["config"].get("ragged_rows"))
PY
