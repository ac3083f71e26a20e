"""Check every ops.linear call of one full-length training step (forward output and the three backward products) against fp64."""
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
orig = ops.linear
calls = []
class Probe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, x, w, b, relu, idx):
        ctx.save_for_backward(x, w, y); ctx.idx = idx; ctx.relu = relu
        return y.view_as(y)
    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        calls[ctx.idx]["dy"] = dy.detach().clone()
        return dy, None, None, None, None, None
def probed(x, w, bias=None, relu=False):
    xd = x.detach().requires_grad_()
    wd = w.detach().requires_grad_()
    xb, wb = x.detach().clone(), w.detach().clone()
    y = orig(x, w, bias, relu)
    torch.cuda.synchronize()
    same_x, same_w = torch.equal(xb, x.detach()), torch.equal(wb, w.detach())
    w2 = w.detach().double().reshape(w.shape[0], -1)
    ref = x.detach().double().reshape(-1, x.shape[-1]) @ w2.t()
    if bias is not None: ref = ref + bias.detach().double()
    if relu: ref = ref.clamp(min=0)
    e = (y.detach().double().reshape(ref.shape) - ref).abs()
    rowmax = e.max(dim=1).values
    bad = (rowmax > 1e-3 * (1 + ref.abs().max(dim=1).values)).nonzero().flatten()
    calls.append(dict(M=ref.shape[0], N=ref.shape[1], K=x.shape[-1], relu=relu, fwd_max=float(e.max()), bad=bad.numel(), first=bad[:5].tolist(),
                      ymax=float(ref.abs().max()), same=(same_x, same_w)))
    return y
ops.linear = probed
import tvqaplus_amd.stage as S
(out, targets), _, _, t_loss, t_scores, other = model.forward_main(batch)
for c in calls:
    print("M=%6d N=%4d K=%4d relu=%d  fwd max err %.3e (|y| max %.2e)  bad rows %d %s" % (c["M"], c["N"], c["K"], c["relu"], c["fwd_max"], c["ymax"], c["bad"], c["first"]), c["same"])
