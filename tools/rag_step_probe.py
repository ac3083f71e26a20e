"""Developer probe: per-step host phases, event-timed step durations and allocator segment counts of the bench step (ragged rows on/off)."""
import os, sys, time, contextlib, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import parallel
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7, fused=True)
batch = make_batch(N=16, seed=2018, att_imgs=4, att_words=3).to("cuda")
n = int(os.environ.get("STEPS", "14"))
marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
rows = []
torch.cuda.synchronize()
marks[0].record()
for i in range(n):
    t0 = time.perf_counter()
    bucket.zero()
    (out, targets), att_loss, _, t_loss, _ = model(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") * (16.0 / len(targets)) + 0.1 * att_loss + 0.5 * t_loss
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    bucket.all_reduce(); torch.nn.utils.clip_grad_norm_(params, 10.0); optim.step()
    t3 = time.perf_counter()
    marks[i + 1].record()
    st = torch.cuda.memory_stats()
    rows.append((t1 - t0, t2 - t1, t3 - t2, st["segment.all.allocated"], st["segment.all.freed"], st["reserved_bytes.all.current"] >> 20,
                 st["num_alloc_retries"]))
torch.cuda.synchronize()
for i, r in enumerate(rows):
    print("step %2d  dev %.2f ms | host fwd %.2f bwd %.2f opt %.2f | segments +%d -%d reserved %d MiB retries %d" % (
        i, marks[i].elapsed_time(marks[i + 1]), 1e3 * r[0], 1e3 * r[1], 1e3 * r[2], r[3], r[4], r[5], r[6]))
print("ragged:", model.last_ragged is not None)
