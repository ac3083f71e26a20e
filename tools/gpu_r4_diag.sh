#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python - > gpurun_out/r4_diag_tables.log 2>&1 <<'PY'
import time, torch, numpy as np
from tvqaplus_amd import ragged
from tvqaplus_amd.synth import make_batch
b = make_batch(N=16, Li=300, Lr=20, Lw=2, Lqa=40, wd_size=4, vfeat_size=4, seed=2018)
qa, fl = ragged.host_masks(b, "vid")
for _ in range(3):
    t0 = time.perf_counter(); tab = ragged.RaggedTables(qa, fl, 4); t1 = time.perf_counter()
    lay = ragged.RaggedLayout(tab, "cuda:0"); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("tables %.3f ms  layout+upload %.3f ms  U=%d S=%d Fc=%d live=%.3f" % (1e3*(t1-t0), 1e3*(t2-t1), tab.U, tab.S, tab.Fc, lay.live_fraction))
stage = lay.stage
for _ in range(5):
    t0 = time.perf_counter(); tab = ragged.RaggedTables(qa, fl, 4); lay = ragged.RaggedLayout(tab, "cuda:0", stage); t2 = time.perf_counter()
    print("both, no sync %.3f ms" % (1e3*(t2-t0)))
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_pmc --no_roofline --dump_steps > gpurun_out/r4_diag_bench.log 2>&1
tail -12 gpurun_out/r4_diag_tables.log
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4_diag_bench.log") if l.startswith("{")][-1])
print(d["ms_per_step"], d["host_issue_ms_per_step"], d["host_wait_ms_per_step"], d.get("device_ms_per_step"))
print(d.get("step_ms_all"))
PY
