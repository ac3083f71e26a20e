"""How long the QA branch (input MLP + input encoder over the N*5 statements: 3200 rows) occupies the GPU on its own, forward + backward --
the part of the step a second stream could hide behind the bandwidth-bound context streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
import contextlib
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
b = make_batch(N=16, Li=4, Lr=20, Lw=50, Lqa=40).to("cuda")
N, NA = 16, 5
def step():
    for p in model.parameters(): p.grad = None
    a = model.base_encoder(b.qas_bert.view(N * NA, -1, model.wd_size), b.qas_mask.view(N * NA, -1), model.bert_word_encoding_fc,
                           model.input_embedding, model.input_encoder)
    a.sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for s, e in ev:
    s.record(); step(); e.record()
torch.cuda.synchronize()
t = sorted(s.elapsed_time(e) for s, e in ev)
print("QA branch forward + backward: median %.3f ms min %.3f ms (event time incl. launch gaps)" % (t[5], t[0]))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
tot = sum(e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total for e in prof.key_averages())
n = sum(e.count for e in prof.key_averages())
print("device time %.3f ms in %d kernels" % (tot / 1e3, n))
