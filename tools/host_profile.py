"""cProfile of the host side of the bench step (no device sync inside the profile): which Python / torch calls the interpreter
spends its time in per training step.  python tools/host_profile.py [steps]"""
import contextlib, cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvqaplus_amd import parallel
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7, fused=True)
BSZ = int(os.environ.get("BSZ", "16"))
batch = make_batch(N=BSZ, seed=2018, att_imgs=4, att_words=3).to("cuda")
for _ in range(4):
    bench.train_step(model, batch, bucket, params, optim, BSZ, 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    bench.train_step(model, batch, bucket, params, optim, BSZ, 1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print("per-step host milliseconds (tottime / cumtime), top 45 by own time; steps =", steps)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt, ct, nc, "%s:%d %s" % (os.path.basename(fn), line, name)))
rows.sort(reverse=True)
for tt, ct, nc, nm in rows[:45]:
    print("%8.3f %8.3f %7.1f  %s" % (1e3 * tt / steps, 1e3 * ct / steps, nc / steps, nm[:110]))
