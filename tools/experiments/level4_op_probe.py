"""Follow-up 2: which per-kernel op of the MAIN stream produces the first output that differs between repeats at level 4?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("STAGE_STREAMS_UNSAFE4", "1")
from tvqaplus_amd import ops
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
N, Li = int(os.environ.get("N", 8)), int(os.environ.get("LI", 150))
opt = make_opt(hsz=256, add_local=True, dropout=0.1, use_sup_att=False, storage_dtype="bf16")
torch.manual_seed(0)
model = STAGE(opt).cuda().train()
model.use_streams = int(os.environ.get("STREAMS", 4))
batch = make_batch(N=N, Li=Li, Lr=20, Lw=512, Lqa=40, seed=2018).to("cuda")
log = []
names = ["layernorm", "cat3_layernorm", "linear", "dwconv", "ln_dwconv", "l2norm", "structured_attention", "masked_max", "ln_masked_max"]
dflt = torch.cuda.default_stream()
def wrap(name):
    orig = getattr(ops, name)
    def spy(*a, **k):
        out = orig(*a, **k)
        if torch.cuda.current_stream() == dflt:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            ins = [x for x in a if torch.is_tensor(x)]
            log.append((name, [tuple(o.shape) for o in outs if torch.is_tensor(o)],
                        [o.detach().clone() for o in outs if torch.is_tensor(o)], [x.detach().clone() for x in ins]))
        return out
    setattr(ops, name, spy)
for n_ in names:
    wrap(n_)
def fwd():
    log.clear()
    model._seed_state = 12345
    with torch.no_grad():
        model.forward_main(batch)
    torch.cuda.synchronize()
    return list(log)
ref = fwd()
print("ops on the main stream per forward:", len(ref))
first = {}
for t in range(int(os.environ.get("TRIALS", 12))):
    cur = fwd()
    assert len(cur) == len(ref)
    for i, ((n0, s0, o0, i0), (n1, s1, o1, i1)) in enumerate(zip(ref, cur)):
        ieq = all(torch.equal(a, b) for a, b in zip(i0, i1))
        oeq = all(torch.equal(a, b) for a, b in zip(o0, o1))
        if not oeq or not ieq:
            det = [(j, tuple(a.shape), int((a != b).sum()), float((a.float() - b.float()).abs().max())) for j, (a, b) in enumerate(zip(i0, i1)) if not torch.equal(a, b)]
            key = (i, n0, tuple(s0), ("inputs differ " + str(det)) if not ieq else "outputs differ, inputs equal")
            first[key] = first.get(key, 0) + 1
            break
print("streams", model.use_streams, "first differing op per repeat:", first or "none")
