"""Which torch (aten) kernels run inside one training step of the bench configuration, and from where: torch.profiler over
3 steps, grouped by operator + input shapes, and by python call site.  Usage (via gpurun): python tools/torch_glue_profile.py"""
import contextlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvqaplus_amd import parallel
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
from torch.profiler import ProfilerActivity, profile

dev = torch.device("cuda", 0)
torch.manual_seed(2018)
HEADS = int(os.environ.get("HEADS", "0"))
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True, input_encoder_n_heads=HEADS, cls_encoder_n_heads=HEADS)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optimizer = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7)
batch = make_batch(N=16, Li=300, Lr=20, Lw=50, Lqa=40, seed=2018, att_imgs=4, att_words=3).to(dev)
for _ in range(3):
    bench.train_step(model, batch, bucket, params, optimizer, 16, 1)
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(STEPS):
        bench.train_step(model, batch, bucket, params, optimizer, 16, 1)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
print("%-34s %6s %10s  %s" % ("op", "n/step", "us/step", "input shapes"))
for e in ev[:45]:
    if e.self_device_time_total / STEPS < 3:
        continue
    print("%-34s %6.1f %10.1f  %s" % (e.key, e.count / STEPS, e.self_device_time_total / STEPS, str(e.input_shapes)[:150]))
print()
st = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.self_device_time_total / STEPS >= 8]
st.sort(key=lambda e: -e.self_device_time_total)
for e in st[:40]:
    frames = [f for f in e.stack if "/repo/" in f or "tvqaplus_amd" in f or "bench.py" in f][:3]
    print("%-26s %5.1f/step %8.1f us/step  %s" % (e.key, e.count / STEPS, e.self_device_time_total / STEPS, " <- ".join(x.strip()[-70:] for x in frames)))
