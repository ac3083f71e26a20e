"""One training step at the bench shape with branch streams 0 / 2 / 3 from the same parameters and batch: are outputs and gradients
bit-identical?  (They should be: the streams change when kernels run, not what they compute.)"""
import os, sys, contextlib, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
torch.manual_seed(2018)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
b = make_batch(N=16, Li=int(os.environ.get("LI", 300)), Lr=20, Lw=50, Lqa=40, seed=2018, att_imgs=4, att_words=3).to("cuda")
res = {}
for lv in (0, 2, 4, 3, 2, 0):
    model.use_streams = lv
    model._seed_state = None
    for p in model.parameters(): p.grad = None
    torch.manual_seed(7)
    (out, tg), att_loss, _, t_loss, _ = model(b)
    loss = F.cross_entropy(out, tg, reduction="sum") * (16 / len(tg)) + 0.5 * t_loss + 0.1 * att_loss
    loss.backward()
    torch.cuda.synchronize()
    cur = (out.detach().clone(), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    if lv in res:
        ref = res[lv]
        tag = "repeat of level %d" % lv
    elif 0 in res:
        ref = res[0]
        tag = "level %d vs level 0" % lv
    else:
        res[lv] = cur
        continue
    res.setdefault(lv, cur)
    dout = float((cur[0] - ref[0]).abs().max())
    worst = max((float((cur[2][k] - ref[2][k]).abs().max() / (ref[2][k].abs().max() + 1e-30)), k) for k in ref[2])
    nd = sum(1 for k in ref[2] if not torch.equal(cur[2][k], ref[2][k]))
    print("%-22s loss %.7f vs %.7f  max|dlogits| %.2e  gradients differing: %d of %d, worst %.2e (%s)" % (tag, cur[1], ref[1], dout, nd, len(ref[2]), worst[0], worst[1]))
