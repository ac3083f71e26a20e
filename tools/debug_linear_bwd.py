"""Check the backward products of every Linear of one full-length training step against fp64 (given the same dy, gate)."""
import os, sys, torch, torch.nn.functional as F, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import ops
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
torch.manual_seed(2018)
opt = make_opt(hsz=128, add_local=True, dropout=0.0)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt)
with torch.no_grad():
    for p in model.parameters(): p.add_(0.05 * torch.randn_like(p))
batch = make_batch(N=1, Li=300, Lr=20, Lw=50, Lqa=40, seed=4).to("cuda")
model = model.cuda().train()
orig_bwd = ops._Linear.backward
def rel(a, b): return float((a.double() - b).abs().max() / (1 + b.abs().max()))
def bwd(ctx, dy):
    x, w2, y, mask = ctx.saved_tensors
    out = orig_bwd(ctx, dy)
    torch.cuda.synchronize()
    dx, dw, db = out[0], out[1], out[2]
    N, K = w2.shape
    M = x.numel() // K
    if M >= 4096:
        g = (y > 0).double().reshape(M, N) if y is not None else 1.0
        dyg = dy.double().reshape(M, N) * g
        rdw = dyg.t() @ x.double().reshape(M, K)
        msg = "M=%6d N=%4d K=%4d mask=%d  dW err %.2e" % (M, N, K, mask is not None, rel(dw.reshape(N, K), rdw))
        if dx is not None: msg += "  dX err %.2e" % rel(dx.reshape(M, K), dyg @ w2.double())
        if db is not None: msg += "  db err %.2e" % rel(db, dyg.sum(0))
        print(msg)
    return out
ops._Linear.backward = staticmethod(bwd)
(out, targets), _, _, t_loss, t_scores, other = model.forward_main(batch)
loss = F.cross_entropy(out, targets, reduction="sum") * (1 / len(targets)) + 0.5 * t_loss
loss.backward()
