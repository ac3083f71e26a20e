"""Phase timers (host) for every step: forward / backward / optimizer / event wait, to localise the periodic slow step."""
import contextlib, gc, os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import parallel
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7)
batch = make_batch(N=16, seed=2018, att_imgs=4, att_words=3).to("cuda")
waits = [0.0]
_sync = torch.cuda.Event.synchronize
def timed(self):
    t = time.perf_counter(); _sync(self); waits[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed
gc.collect(); gc.freeze()
rows = []
for i in range(steps):
    waits[0] = 0.0
    t0 = time.perf_counter()
    bucket.zero()
    (out, targets), att_loss, _, t_loss, _ = model(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") * (16.0 / len(targets)) + 0.1 * att_loss + 0.5 * t_loss
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    bucket.all_reduce()
    t3 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(params, 10.0)
    t4 = time.perf_counter()
    optim.step()
    t5 = time.perf_counter()
    rows.append([1e3 * x for x in (t1 - t0 - waits[0], waits[0], t2 - t1, t3 - t2, t4 - t3, t5 - t4)] + [len(targets)])
torch.cuda.synchronize()
print("step  fwd_issue  wait  backward  bucket  clip  adam  N_new")
for i, r in enumerate(rows):
    print("%3d  %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f  %d" % tuple([i] + r))
