"""Several optimizer steps at the bench shape with branch streams 0 and 2 from the same parameters: where do the trajectories part?"""
import os, sys, copy, contextlib, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
opt = make_opt(hsz=128, add_local=True, dropout=0.1, use_sup_att=True)
torch.manual_seed(2018)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    base = STAGE(opt).cuda().train()
b = make_batch(N=16, Li=int(os.environ.get("LI", 300)), Lr=20, Lw=50, Lqa=40, seed=2018, att_imgs=4, att_words=3).to("cuda")
init = copy.deepcopy(base.state_dict())
hist = {}
for lv in (0, 2, 0):
    base.load_state_dict(init)
    base.use_streams = lv
    base._seed_state = None
    params = [p for p in base.parameters() if p.requires_grad]
    optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7, fused=True)
    rec = []
    for step in range(6):
        optim.zero_grad(set_to_none=True)
        torch.manual_seed(100 + step)
        (out, tg), att_loss, _, t_loss, _ = base(b)
        loss = F.cross_entropy(out, tg, reduction="sum") * (16 / len(tg)) + 0.5 * t_loss + 0.1 * att_loss
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        optim.step()
        torch.cuda.synchronize()
        rec.append((float(loss), torch.cat([p.detach().reshape(-1) for p in params]).clone()))
    key = "level %d%s" % (lv, " (again)" if lv in hist else "")
    hist.setdefault(lv, rec)
    if lv != 0 or "again" in key:
        ref = hist[0]
        for s, ((l1, p1), (l0, p0)) in enumerate(zip(rec, ref)):
            print("%-16s step %d: loss %.7f vs %.7f  max|dparam| %.3e  equal %s" % (key, s, l1, l0, float((p1 - p0).abs().max()), torch.equal(p1, p0)))
