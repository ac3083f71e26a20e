"""Which fp32 gradient is closer to an fp64 evaluation: the reference's (fixture) or the HIP path's?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from conftest import Fixture
from oracle import stage_oracle as O
from tvqaplus_amd.stage import STAGE

name = sys.argv[1] if len(sys.argv) > 1 else "mid_train"
fx = Fixture(name)
opt = fx.opt
opt.mha_dropout = 0.0
P64 = {k: (v.double().requires_grad_(not k.endswith(".pe")) if v.is_floating_point() else v) for k, v in fx.group("param").items()}
b = fx.batch()
b64 = type(b)({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()})
ref = O.stage_forward(P64, opt, b64, training=True)
O.training_loss(ref, n_examples=b.target.shape[0]).backward()

model = STAGE(fx.opt)
model.load_state_dict(fx.group("param"), strict=True)
model.mha_dropout_override = 0.0
model = model.cuda().train()
batch = b.to("cuda")
(out, targets), _, _, t_loss, t_scores, other = model.forward_main(batch)
loss = F.cross_entropy(out, targets, reduction="sum") * (len(batch.qid) / len(targets)) + 0.5 * t_loss
loss.backward()
G = fx.group("grad")
print("%-66s %10s %12s %12s" % ("param", "|g|max", "ref32-f64", "hip32-f64"))
for k, p in model.named_parameters():
    g64 = P64[k].grad if P64[k].grad is not None else torch.zeros_like(P64[k])
    e_ref = float((G[k].double() - g64).abs().max())
    e_hip = float((p.grad.cpu().double() - g64).abs().max()) if p.grad is not None else 0.0
    if max(e_ref, e_hip) > 1e-4:
        print("%-66s %10.3e %12.3e %12.3e" % (k, float(g64.abs().max()), e_ref, e_hip))
