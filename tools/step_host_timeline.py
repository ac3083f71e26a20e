"""Host-side timeline of the bench step: time the host spends in forward / backward / optimizer calls (no sync) and the
synchronised step time.  python tools/step_host_timeline.py [--no_sup_att]"""
import os, sys, time, contextlib, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import parallel
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
sup = "--no_sup_att" not in sys.argv
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=sup, storage_dtype=os.environ.get("STORAGE", "fp32"))
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradBucket(params)
optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7, fused=True)
BSZ = int(os.environ.get("BSZ", "16"))
batch = make_batch(N=BSZ, seed=2018, att_imgs=4 if sup else 0, att_words=3).to("cuda")
acc = [0.0] * 4
def step(record):
    t0 = time.perf_counter()
    bucket.zero()
    (out, targets), att_loss, _, t_loss, _ = model(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") * (float(BSZ) / len(targets)) + 0.1 * att_loss + 0.5 * t_loss
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    bucket.all_reduce(); torch.nn.utils.clip_grad_norm_(params, 10.0); optim.step()
    t3 = time.perf_counter()
    if record:
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2)): acc[i] += v
for _ in range(3): step(False)
torch.cuda.synchronize(); T0 = time.perf_counter()
n = 10
for _ in range(n): step(True)
torch.cuda.synchronize(); T = time.perf_counter() - T0
print("sup_att", sup, "step %.2f ms | host: forward %.2f backward %.2f optimizer %.2f ms" % (1e3 * T / n, 1e3 * acc[0] / n, 1e3 * acc[1] / n, 1e3 * acc[2] / n))
# finer: time spent in selected host functions (monkeypatched timers), per step
import tvqaplus_amd.att_host as AH
tm = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); tm[key] = tm.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
wrap(AH, "build_att_pairs", "build_att_pairs"); wrap(AH, "get_att_loss", "get_att_loss")
wrap(torch.cuda.Event, "synchronize", "event.synchronize")
for nm in ("_ragged_layout", "base_encoder", "qa_ctx_attention", "classfier_head_multi_proposal", "_proposals_grouped", "get_ts_loss", "_open_gates"):
    wrap(STAGE, nm, "STAGE." + nm)
import tvqaplus_amd.groups as GR
for nm in ("concat_fc", "temporal_head", "pool_classifier", "encoder_block_rag", "encoder_block", "input_mlp", "input_mlp_rag", "qa_ctx_rag", "gate", "gt_spans", "masked_max_raw", "tscores", "att_loss", "ts_loss"):
    if hasattr(GR, nm):
        wrap(GR, nm, "groups." + nm)
wrap(bucket, "all_reduce", "bucket.all_reduce"); wrap(optim, "step", "optim.step")
_cg = torch.nn.utils.clip_grad_norm_
def _cgt(*a, **k):
    t = time.perf_counter(); r = _cg(*a, **k); tm["clip_grad_norm_"] = tm.get("clip_grad_norm_", 0.0) + time.perf_counter() - t; return r
torch.nn.utils.clip_grad_norm_ = _cgt
torch.cuda.synchronize()
for _ in range(n): step(False)
torch.cuda.synchronize()
print({k: round(1e3 * v / n, 3) for k, v in tm.items()})
