"""Level-4 branch streams on the per-kernel (bf16 storage) path: which forward output differs from level 0, and how often?"""
import os, sys, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tvqaplus_amd.stage as S
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
opt = make_opt(hsz=256, add_local=True, dropout=0.1, use_sup_att=True, storage_dtype="bf16")
torch.manual_seed(2018)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    model = STAGE(opt).cuda().train()
N = int(os.environ.get("NB", 8))
b = make_batch(N=N, Li=int(os.environ.get("LI", 150)), Lr=20, Lw=512, Lqa=40, seed=2018, att_imgs=4, att_words=3).to("cuda")
for k in ("vid", "sub_bert"):
    setattr(b, k, getattr(b, k).to(torch.bfloat16))
REC = {}
_orig = model.qa_ctx_attention
def _rec(qa, ctx, qm, cm, lay=None, clay=None):
    tag = "sub" if ctx.shape[2] > 64 else "vid"
    REC[tag + "_in_qa"], REC[tag + "_in_ctx"] = qa.float().clone(), ctx.float().clone()
    REC[tag + "_in_qm"], REC[tag + "_in_cm"] = qm.float().clone(), cm.float().clone()
    if os.environ.get("KEEP"):
        REC["_keep_" + tag] = ctx
    res = _orig(qa, ctx, qm, cm, lay, clay)
    REC[tag + "_out_mixed"] = res[0].float().clone()
    return res
model.qa_ctx_attention = _rec
def run(level):
    model.use_streams = level
    model._seed_state = None
    torch.manual_seed(7)
    with torch.no_grad():
        out, _, _, t_loss, t_scores, other = model.forward_main(b)
    torch.cuda.synchronize()
    out = out[0] if isinstance(out, (list, tuple)) else out
    r = {"logits": out.float().clone(), "t_scores": t_scores.float().clone()}
    for k, v in other.items():
        if torch.is_tensor(v):
            r[k] = v.float().clone()
    for k in list(REC):
        if k.startswith("_keep_"):
            REC[k[6:] + "_in_ctx_late"] = REC.pop(k).float().clone()
    r.update(REC)
    return r
import tvqaplus_amd.ops as OPS
TRACE = []
def _wrap(name):
    f = getattr(OPS, name)
    def g(*a, **k):
        out = f(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        st = torch.cuda.current_stream().cuda_stream
        which = os.environ.get("TRACE")
        if which == "1" or (which == "main") == (st == MAIN):
            TRACE.append((name, st, [o for o in outs if torch.is_tensor(o)], [x for x in a if torch.is_tensor(x)]))
        return out
    setattr(OPS, name, g)
if os.environ.get("TRACE"):
    for nm in ("linear", "ln_dwconv", "layernorm", "l2norm", "input_ln_linear", "dwconv", "structured_attention", "cat3_layernorm"):
        if hasattr(OPS, nm):
            _wrap(nm)
MAIN = main_handle = torch.cuda.current_stream().cuda_stream
ref = run(0)
REF_TRACE = [(n, s, [o.float().clone() for o in outs]) for n, s, outs, ins in TRACE] if os.environ.get("TRACE") == "1" else []
os.environ["STAGE_STREAMS_UNSAFE4"] = "1"
for trial in range(6):
    TRACE.clear()
    cur = run(4)
    if REF_TRACE:
        assert len(TRACE) == len(REF_TRACE)
        for i, ((n, st, outs, ins), (rn, rs, routs)) in enumerate(zip(TRACE, REF_TRACE)):
            assert n == rn
            d = [not torch.equal(o.float(), r) for o, r in zip(outs, routs)]
            if any(d):
                o, r = outs[d.index(True)].float(), routs[d.index(True)]
                rows = (o != r).view(-1, o.shape[-1]).any(-1).nonzero().flatten()
                print("   first differing op: #%d %s stream=%s shape=%s rows differing %d of %d: %s ..." % (
                    i, n, "main" if st == main_handle else hex(st), tuple(o.shape), rows.numel(), o.numel() // o.shape[-1], rows[:10].tolist()))
                print("   ops before it:", [(j, TRACE[j][0], "main" if TRACE[j][1] == main_handle else "side") for j in range(max(0, i - 6), i)])
                break
    bad = [k for k in ref if not torch.equal(cur[k], ref[k])]
    print("trial", trial, "differing:", bad, {k: float((cur[k] - ref[k]).abs().max()) for k in bad})
    for k in bad:
        if k.endswith("_in_ctx") or k.endswith("_in_qa"):
            d = (cur[k] != ref[k])
            rows = d.view(-1, d.shape[-1]).any(-1).nonzero().flatten()
            print("   ", k, tuple(d.shape), "rows differing:", rows.numel(), "first/last", rows[:6].tolist(), rows[-3:].tolist(),
                  "cols of first row:", d.view(-1, d.shape[-1])[rows[0]].nonzero().flatten()[:8].tolist(),
                  "nan:", bool(torch.isnan(cur[k]).any()))
