"""Level-4 branch streams on the stress configuration (bf16 storage, hsz 256, 512-word rows; per-kernel path): do two forwards of the
same batch with the same dropout seeds give the same bits?  Reports which returned tensor differs first.  STREAMS=3|4 (4 needs
STAGE_STREAMS_UNSAFE4=1), N / LI shrink the batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("STAGE_STREAMS_UNSAFE4", "1")
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
N, Li = int(os.environ.get("N", 8)), int(os.environ.get("LI", 150))
opt = make_opt(hsz=256, add_local=True, dropout=0.1, use_sup_att=True, storage_dtype="bf16")
torch.manual_seed(0)
model = STAGE(opt).cuda().train()
model.use_streams = int(os.environ.get("STREAMS", 4))
batch = make_batch(N=N, Li=Li, Lr=20, Lw=512, Lqa=40, seed=2018, att_imgs=4, att_words=3).to("cuda")
def fwd(backward):
    model._seed_state = 12345
    model.zero_grad(set_to_none=True)
    (out, targets), att_loss, _, t_loss, t_scores, other = model.forward_main(batch)
    res = {"logits": out.float(), "t_scores": t_scores.float(), "att_loss": torch.as_tensor(att_loss).float().reshape(1), "t_loss": t_loss.float().reshape(1)}
    for k, v in other.items():
        if torch.is_tensor(v):
            res[k] = v.float()
    if backward:
        loss = torch.nn.functional.cross_entropy(out.float(), targets, reduction="sum") + 0.1 * att_loss + 0.5 * t_loss
        loss.backward()
        for n_, p_ in model.named_parameters():
            if p_.grad is not None:
                res["grad:" + n_] = p_.grad.float().clone()
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in res.items()}
bw = os.environ.get("BWD", "1") == "1"
ref = fwd(bw)
diffs = {}
for t in range(int(os.environ.get("TRIALS", 8))):
    cur = fwd(bw)
    for k in ref:
        if k in cur and cur[k].shape == ref[k].shape and not torch.equal(cur[k], ref[k]):
            diffs.setdefault(k, []).append(float((cur[k] - ref[k]).abs().max()))
print("streams", model.use_streams, "tensors that differed between repeats:", {k: (len(v), max(v)) for k, v in diffs.items()} or "none")
