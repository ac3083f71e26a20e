"""Per-parameter gradient error of one full-length example vs the fp64 oracle (top 12), for bisecting kernel paths by env switch."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import rel_err
from oracle import stage_oracle as O
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
torch.manual_seed(int(os.environ.get("SEED", 2018)))
opt = make_opt(hsz=128, add_local=True, dropout=0.0)
import contextlib
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt)
with torch.no_grad():
    for p in model.parameters():
        p.add_(0.05 * torch.randn_like(p))
Li = int(os.environ.get("LI", 300))
batch = make_batch(N=1, Li=Li, Lr=20, Lw=50, Lqa=40, seed=int(os.environ.get("BSEED", 4)))
P = {k: (v.double().requires_grad_(not k.endswith(".pe")) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
b64 = type(batch)({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()})
ref = O.stage_forward(P, opt, b64, training=True)
ref_loss = O.training_loss(ref, n_examples=1)
ref_loss.backward()
model = model.cuda().train()
(out, targets), _, _, t_loss, t_scores, other = model.forward_main(batch.to("cuda"))
loss = F.cross_entropy(out, targets, reduction="sum") * (1 / len(targets)) + 0.5 * t_loss
loss.backward()
print("logits %.2e t_scores %.2e loss %.2e" % (rel_err(out, ref["logits"]), rel_err(t_scores, ref["t_scores"]), rel_err(loss, ref_loss)))
errs = []
for k, p in model.named_parameters():
    g = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
    got = p.grad if p.grad is not None else torch.zeros_like(p)
    errs.append((rel_err(got, g), k, float(g.abs().max())))
errs.sort(reverse=True)
for e, k, m in errs[:12]:
    print("%.3e  %-55s max|g| %.3e" % (e, k, m))
