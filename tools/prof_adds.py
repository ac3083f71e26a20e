"""Which aten ops run beside the HIP kernels in one training step (shapes + stack) -- developer tool."""
import os, sys, torch, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from tvqaplus_amd import parallel
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.1)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt)
model = model.cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7)
batch = make_batch(N=16, Li=300, Lr=20, Lw=50, Lqa=40, seed=2018).to("cuda")
def step():
    bucket.zero()
    (out, targets), _, _, t_loss, _ = model(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") * (16.0 / len(targets)) + 0.5 * t_loss
    loss.backward()
    bucket.all_reduce()
    torch.nn.utils.clip_grad_norm_(params, 10.0)
    optim.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=40, max_shapes_column_width=60))
