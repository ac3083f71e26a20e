"""Full-size timings (us) of the streaming row kernels through the op layer: cat3 LayerNorm fwd/bwd (broadcast + plain), LayerNorm
fwd/bwd, fused LayerNorm->dwconv fwd/bwd, masked max.  LIB=<path> selects an experiment build of the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
from tvqaplus_amd import ops
torch.manual_seed(0)
dev = "cuda"
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
N, NA, Li, Lqa, D = 16, 5, 300, 40, 128
U = N * NA * Li * Lqa
res = {}
a = torch.randn(N * NA * Lqa, D, device=dev, requires_grad=True)
b = torch.randn(U, D, device=dev, requires_grad=True)
g3, b3 = torch.randn(3 * D, device=dev, requires_grad=True), torch.randn(3 * D, device=dev, requires_grad=True)
def cat3_fwd(): return ops.cat3_layernorm(a, b, g3, b3, rep=Li, inner=Lqa, p=0.1, seed=3)
res["cat3 fwd (rep)"] = t(cat3_fwd)
z = cat3_fwd(); gz = torch.randn_like(z)
res["cat3 fwd+bwd (rep)"] = t(lambda: torch.autograd.grad(cat3_fwd(), (a, b, g3, b3), gz))
del z, gz
a2 = torch.randn(U, D, device=dev, requires_grad=True)
def cat3p(): return ops.cat3_layernorm(a2, b, g3, b3, rep=1, inner=1, p=0.1, seed=3)
z = cat3p(); gz = torch.randn_like(z)
res["cat3 fwd (plain)"] = t(cat3p)
res["cat3 fwd+bwd (plain)"] = t(lambda: torch.autograd.grad(cat3p(), (a2, b, g3, b3), gz))
del z, gz, a2
g1, b1 = torch.randn(D, device=dev, requires_grad=True), torch.randn(D, device=dev, requires_grad=True)
def ln(): return ops.layernorm(b, g1, b1)[0]
y = ln(); gy = torch.randn_like(y)
res["ln fwd"] = t(ln)
res["ln fwd+bwd"] = t(lambda: torch.autograd.grad(ln(), (b, g1, b1), gy))
x3 = b.view(U // Lqa, Lqa, D)
w = torch.randn(D, 1, 5, device=dev, requires_grad=True); bc = torch.randn(D, device=dev, requires_grad=True)
def ldw(): return ops.ln_dwconv(x3, g1, b1, w, bc, p=0.1, seed=5)
o = ldw()
o0 = o[0] if isinstance(o, tuple) else o
go = torch.randn_like(o0)
res["ln_dwconv k5 fwd"] = t(ldw)
def ldw_fb():
    o = ldw(); o0 = o[0] if isinstance(o, tuple) else o
    return torch.autograd.grad(o0, (b, g1, b1, w, bc), go)
res["ln_dwconv k5 fwd+bwd"] = t(ldw_fb)
m = torch.ones(U // Lqa, Lqa, device=dev)
res["masked_max fwd"] = t(lambda: ops.masked_max(x3.detach(), m))
print(" | ".join("%s %.0f" % kv for kv in res.items()))
