"""Per-launch times of the isolated K1 forward (video shape): sequence of 40 launches, microseconds."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
from tvqaplus_amd.synth import make_batch
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])   # experiment builds
lib = _lib.load()
dev = "cuda"
N, NA, Li, Lqa, Lr, D = 16, 5, 300, 40, int(os.environ.get("LR", 20)), 128
g = torch.Generator().manual_seed(2018)
b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=2018, ragged=os.environ.get("DENSE") is None)
Cn = F.normalize(torch.randn(N, NA, Lqa, D, generator=g), dim=-1).to(dev)
Q = torch.randn(N, Li, Lr, D, generator=g).to(dev)
if os.environ.get("CONST"):   # constant operands: the matrix cores draw far less power (DESIGN.md finding 14)
    Cn = torch.full_like(Cn, D ** -0.5); Q = torch.full_like(Q, 0.5)
cm, qm = b.qas_mask.to(dev).contiguous(), b.vid_mask.to(dev).contiguous()
A = torch.empty(N, NA, Li, Lqa, D, device=dev); S = torch.empty(N, NA, Li, Lqa, Lr, device=dev); Sn = torch.empty_like(S)
st = torch.cuda.current_stream()
p = float(os.environ.get("P", 0.0))
def launch():
    _lib.check(lib.stage_str_attn_fwd(Cn.data_ptr(), Q.data_ptr(), cm.data_ptr(), qm.data_ptr(), A.data_ptr(), S.data_ptr(), Sn.data_ptr(),
                                      N, NA, Li, Lqa, Lr, D, 10.0, p, 1, st.cuda_stream), "k1")
for _ in range(5): launch()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
gap = float(os.environ.get("GAP", 0))   # idle milliseconds in front of every launch (power / clock recovery)
for s, e in ev:
    if gap:
        torch.cuda.synchronize(); import time; time.sleep(gap * 1e-3)
    s.record(st); launch(); e.record(st)
torch.cuda.synchronize()
t = [s.elapsed_time(e) * 1e3 for s, e in ev]
print(" ".join("%.0f" % x for x in t))
print("avg %.1f min %.1f max %.1f" % (sum(t) / len(t), min(t), max(t)))
if os.environ.get("TIM"):     # phase cycle counters of the LDS-staged kernel (STAGE_K1_TIM): sum over waves
    tim = torch.zeros(6, dtype=torch.int64, device=dev)
    os.environ["STAGE_K1_TIM"] = str(tim.data_ptr())
    launch(); torch.cuda.synchronize()
    del os.environ["STAGE_K1_TIM"]
    v = tim.cpu().tolist(); tot = float(sum(v))
    names = ["stage frame", "wait Cn", "stage 1", "softmax + S stores", "stage 2 + A stores", "ticket / loop"]
    print(" | ".join("%s %.1f%%" % (n, 100 * x / tot) for n, x in zip(names, v)), "| total wave-cycles %.3g" % tot)
