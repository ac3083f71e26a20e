"""K1 backward at the BASELINE shapes: fused single-pass kernel vs the three-kernel path (same inputs): max differences
and per-launch times (microseconds, events on the launch stream).  LR=20 video stream, LR=50 subtitle stream.
EXT=1 dense gradient on raw_s, EXT=sparse on 1 % of the valid entries."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
from tvqaplus_amd.synth import make_batch
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])   # experiment builds (tools/build_variant.sh)
lib = _lib.load()
dev = "cuda"
N, NA, Li, Lqa, Lr, D = int(os.environ.get("NB", 16)), 5, int(os.environ.get("LI", 300)), 40, int(os.environ.get("LR", 20)), 128
EXT = bool(os.environ.get("EXT"))
g = torch.Generator().manual_seed(2018)
b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=2018, ragged=os.environ.get("DENSE") is None)
C = torch.randn(N, NA, Lqa, D, generator=g).to(dev)
Q = torch.randn(N, Li, Lr, D, generator=g).to(dev)
cm, qm = b.qas_mask.to(dev).contiguous(), b.vid_mask.to(dev).contiguous()
st = torch.cuda.current_stream().cuda_stream
Cn = torch.empty_like(C); Qn = torch.empty_like(Q)
lib.stage_l2norm_fwd(C.data_ptr(), Cn.data_ptr(), None, N * NA * Lqa, D, 1e-12, 0.1, 11, st)
lib.stage_l2norm_fwd(Q.data_ptr(), Qn.data_ptr(), None, N * Li * Lr, D, 1e-12, 0.1, 12, st)
A = torch.empty(N, NA, Li, Lqa, D, device=dev); S = torch.empty(N, NA, Li, Lqa, Lr, device=dev); Sn = torch.empty_like(S)
_lib.check(lib.stage_str_attn_fwd(Cn.data_ptr(), Q.data_ptr(), cm.data_ptr(), qm.data_ptr(), A.data_ptr(), S.data_ptr(), Sn.data_ptr(),
                                  N, NA, Li, Lqa, Lr, D, 10.0, 0.1, 12, st), "fwd")
dA = torch.randn(A.shape, generator=g).to(dev)
ext = (torch.randn(S.shape, generator=g) * 0.1).to(dev) if EXT else None
if os.environ.get("EXT") == "sparse":   # as the supervised-attention loss produces it: a few labelled (valid) regions
    pair = (cm.view(N, NA, 1, Lqa, 1) * qm.view(N, 1, Li, 1, Lr))
    ext = ext * pair * (torch.rand(S.shape, generator=g).to(dev) < 0.01)
ep = ext.data_ptr() if EXT else None
del A, S
wsb = max(lib.stage_str_attn_bwd_ws_bytes(N, NA, Lqa, D), lib.stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
dS = torch.empty_like(Sn)
out = {k: [torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(C)] for k in ("old", "new")}
def old():
    o = out["old"]
    _lib.check(lib.stage_str_attn_bwd(dA.data_ptr(), ep, Cn.data_ptr(), Q.data_ptr(), Qn.data_ptr(), Sn.data_ptr(), dS.data_ptr(),
                                      o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0, ws.data_ptr(), wsb, st), "bwd")
def new():
    o = out["new"]
    _lib.check(lib.stage_str_attn_bwd_fused(dA.data_ptr(), ep, Cn.data_ptr(), Q.data_ptr(), Qn.data_ptr(), Sn.data_ptr(), qm.data_ptr(),
                                            o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0, ws.data_ptr(), wsb, st), "fused")
old(); new(); torch.cuda.synchronize()
for nm, x, y in zip(("dQraw", "dQn", "dCn"), out["old"], out["new"]):
    d = (x - y).abs()
    print("%-6s max|old-new| %.3e  rel-to-scale %.3e  max|old| %.3e  nan %d" % (nm, float(d.max()), float((d / (1 + x.abs())).max()), float(x.abs().max()), int(torch.isnan(y).sum())))
new(); torch.cuda.synchronize()
y2 = [t.clone() for t in out["new"]]
new(); torch.cuda.synchronize()
print("deterministic:", all(torch.equal(a, b) for a, b in zip(y2, out["new"])))
def timeit(fn, name, reps=20):
    for _ in range(3): fn()
    cs = torch.cuda.current_stream()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(cs); fn(); e.record(cs)
    torch.cuda.synchronize()
    t = [s.elapsed_time(e) * 1e3 for s, e in ev]
    print("%s: avg %.1f min %.1f max %.1f us" % (name, sum(t) / len(t), min(t), max(t)))
timeit(old, "three-kernel"); timeit(new, "fused")
if os.environ.get("TIM"):     # phase cycle counters (STAGE_K1_BWD_TIM): sum over waves
    tim = torch.zeros(6, dtype=torch.int64, device=dev)
    os.environ["STAGE_K1_BWD_TIM"] = str(tim.data_ptr())
    new(); torch.cuda.synchronize()
    del os.environ["STAGE_K1_BWD_TIM"]
    v = tim.cpu().tolist(); tot = float(sum(v))
    names = ["stage/skip", "wait S1", "phase 1", "wait S2", "phase 2", "slab"]
    print(" | ".join("%s %.1f%%" % (n, 100 * x / tot) for n, x in zip(names, v)), "| total wave-cycles %.3g" % tot)
