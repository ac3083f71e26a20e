"""Which timed steps are slow, and does the caching allocator call hipMalloc inside them?  python tools/spike_probe.py [steps]"""
import contextlib, gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvqaplus_amd import parallel
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7)
batch = make_batch(N=16, seed=2018, att_imgs=4, att_words=3).to("cuda")
for _ in range(5):
    bench.train_step(model, batch, bucket, params, optim, 16, 1)
if os.environ.get("GC", "freeze") == "freeze":
    gc.collect(); gc.freeze()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
st = torch.cuda.current_stream()
segs, host = [], []
ev[0].record(st)
for i in range(steps):
    t0 = time.perf_counter()
    bench.train_step(model, batch, bucket, params, optim, 16, 1)
    host.append(1e3 * (time.perf_counter() - t0))
    ev[i + 1].record(st)
    ms = torch.cuda.memory_stats()
    segs.append((ms["segment.all.allocated"], ms["num_alloc_retries"], ms["reserved_bytes.all.current"] >> 20))
torch.cuda.synchronize()
d = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
med = sorted(d)[len(d) // 2]
print("median %.2f" % med)
for i, x in enumerate(d):
    flag = " <-- slow" if x > med + 1.0 else ""
    new_seg = segs[i][0] - (segs[i - 1][0] if i else segs[0][0])
    if flag or new_seg:
        print("step %3d  device %.2f ms  host %.2f ms  new segments %d  reserved %d MiB%s" % (i, x, host[i], new_seg, segs[i][2], flag))
