"""Host time of every C-ABI call of one training step (perf_counter around each ctypes call; no profiler): which library calls the
host BLOCKS in.  BSZ=2 python tools/host_calls.py [steps]"""
import contextlib, os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvqaplus_amd import parallel, _lib
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BSZ = int(os.environ.get("BSZ", "16"))
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
if os.environ.get("STREAMS"):
    model.use_streams = int(os.environ["STREAMS"])
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7, fused=True)
batch = make_batch(N=BSZ, seed=2018, att_imgs=4, att_words=3).to("cuda")
lib = _lib.load()
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
on = [False]
def wrap(name, f):
    def g(*a):
        if not on[0]:
            return f(*a)
        t = time.perf_counter(); r = f(*a); dt = time.perf_counter() - t
        e = acc[name]; e[0] += dt; e[1] += 1; e[2] = max(e[2], dt)
        return r
    return g
for name in _lib.SIGNATURES:
    setattr(lib, name, wrap(name, getattr(lib, name)))
for _ in range(5):
    bench.train_step(model, batch, bucket, params, optim, BSZ, 1)
torch.cuda.synchronize()
on[0] = True
t0 = time.perf_counter()
for _ in range(steps):
    bench.train_step(model, batch, bucket, params, optim, BSZ, 1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("bsz %d: issue %.3f ms/step, synced %.3f ms/step; C-ABI calls: total %.3f ms/step in %d calls/step" % (
    BSZ, 1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps, 1e3 * sum(v[0] for v in acc.values()) / steps, sum(v[1] for v in acc.values()) / steps))
for name, (t, n, mx) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.3f ms/step  %5.1f calls/step  avg %7.1f us  max %8.1f us  %s" % (1e3 * t / steps, n / steps, 1e6 * t / n, 1e6 * mx, name))
