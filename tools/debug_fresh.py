import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import stage_oracle as O
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt

for empty in (False, True):
    torch.manual_seed(77)
    opt = make_opt(embedding_size=80, vfeat_size=52, dropout=0.0, hsz=64, add_local=True)
    model = STAGE(opt)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.1 * torch.randn_like(p))
    batch = make_batch(N=3, Li=9, Lr=11, Lw=14, Lqa=10, wd_size=80, vfeat_size=52, seed=5, empty_frames=empty)
    P = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in model.state_dict().items()}
    ref = O.stage_forward(P, opt, batch, training=True)
    ref_loss = O.training_loss(ref, n_examples=3)
    ref_loss.backward()
    model = model.cuda().train()
    (out, targets), _, _, t_loss, t_scores, other = model.forward_main(batch.to("cuda"))
    loss = F.cross_entropy(out, targets, reduction="sum") * (3 / len(targets)) + 0.5 * t_loss
    loss.backward()
    print("empty_frames", empty, "loss", float(loss), float(ref_loss), "targets", targets.tolist(), ref["targets"].tolist())
    for k, p in model.named_parameters():
        g = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu()
        rel = float(((got - g).abs() / (1 + g.abs())).max())
        if rel > 1e-4:
            print("  %-60s |g|max %.3e abs err %.3e rel %.2e" % (k, float(g.abs().max()), float((got - g).abs().max()), rel))
