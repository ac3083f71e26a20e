"""The training step of tests/test_hip_stage.py::test_branch_streams_change_nothing_but_the_schedule[True-groups] repeated at ONE stream
level: which parameter gradients differ between repeats?"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
torch.manual_seed(11)
kw = dict(input_encoder_n_heads=4, cls_encoder_n_heads=4) if os.environ.get("HEADS") else {}
opt = make_opt(hsz=128, embedding_size=96, vfeat_size=64, dropout=0.1, add_local=True, use_sup_att=True, **kw)
model = STAGE(opt).cuda().train()
batch = make_batch(N=4, Li=48, Lr=20, Lw=30, Lqa=40, wd_size=96, vfeat_size=64, seed=3, att_imgs=3, att_words=2).to("cuda")
model.use_streams = int(os.environ.get("STREAMS", 0))
def run():
    model._seed_state = None
    for p in model.parameters(): p.grad = None
    torch.manual_seed(5)
    (out, targets), att_loss, _, t_loss, t_scores, other = model.forward_main(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") + 0.5 * t_loss + 0.1 * att_loss
    loss.backward()
    if not os.environ.get("NOSYNC"): torch.cuda.synchronize()
    return {"out": out.detach().clone(), "loss": loss.detach().clone(), **{n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}}
ref = run()
cnt = {}
nbad = 0
levels = [int(x) for x in os.environ.get("LEVELS", "2,4,3,1,0").split(",")]
for t in range(int(os.environ.get("TRIALS", 40))):
    model.use_streams = levels[t % len(levels)]
    cur = run()
    bad = [k for k in ref if not torch.equal(ref[k], cur[k])]
    if bad:
        nbad += 1
        for k in bad:
            cnt[k] = cnt.get(k, 0) + 1
            d = (ref[k] != cur[k]).flatten()
            if len(bad) <= 3: print("  step", t, "level", model.use_streams, k, "elements", int(d.sum()), "idx", d.nonzero().flatten()[:12].tolist(), "max|diff|", float((ref[k] - cur[k]).abs().max()), "of", float(ref[k].abs().max()))
print("STAGE_CAT3_DW", os.environ.get("STAGE_CAT3_DW"), "repeats that differ:", nbad, "equal everywhere:", sorted(set(ref) - set(cnt)))
print("differ:", cnt)
print("levels", levels)
