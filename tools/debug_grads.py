import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from conftest import Fixture
from tvqaplus_amd.stage import STAGE

for name in sys.argv[1:] or ["small_local_train"]:
    fx = Fixture(name)
    model = STAGE(fx.opt)
    model.load_state_dict(fx.group("param"), strict=True)
    model.mha_dropout_override = 0.0
    model = model.cuda().train()
    batch = fx.batch().to("cuda")
    (out, targets), _, _, t_loss, t_scores, other = model.forward_main(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") * (len(batch.qid) / len(targets)) + 0.5 * t_loss
    loss.backward()
    G = fx.group("grad")
    print(name, "targets", targets.tolist(), "loss", float(loss), float(fx["out/loss"]))
    for k, p in model.named_parameters():
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu()
        e = (got - G[k]).abs().max()
        print("%-70s |g|max %.3e  abs err %.3e  rel %.2e" % (k, float(G[k].abs().max()), float(e),
                                                              float(((got - G[k]).abs() / (1 + G[k].abs())).max())))
