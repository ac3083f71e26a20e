"""Repeat-launch determinism of stage_cat3_bwd_dw at a given shape: which outputs differ between launches, and where."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("STAGE_CAT3_DW", "1")
from tvqaplus_amd import _lib
lib = _lib.load()
D = 128
rep, inner, G = int(os.environ.get("REP", 48)), int(os.environ.get("INNER", 40)), int(os.environ.get("G", 20))
U = G * rep * inner if rep > 1 else G
p = float(os.environ.get("P", 0.1))
g = torch.Generator().manual_seed(23)
a = torch.randn((U // rep) if rep > 1 else U, D, generator=g).cuda(); b = torch.randn(U, D, generator=g).cuda()
gamma = (1 + 0.1 * torch.randn(3 * D, generator=g)).cuda(); beta = (0.1 * torch.randn(3 * D, generator=g)).cuda()
W = (0.08 * torch.randn(D, 3 * D, generator=g)).cuda()
dy = torch.randn(U, D, generator=g).cuda()
mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (D // 32, U), generator=g, dtype=torch.int64).to(torch.int32).cuda()
st = torch.cuda.current_stream().cuda_stream
z = torch.empty(U, 3 * D, device="cuda"); mean = torch.empty(U, device="cuda"); rstd = torch.empty(U, device="cuda")
_lib.check(lib.stage_cat3_layernorm_fwd(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), U, D, rep, inner, 1e-5, p, 77, st), "fwd")
assert lib.stage_cat3_bwd_dw_supported(U, D, rep, inner)
wsb = lib.stage_cat3_bwd_dw_ws_bytes(U, D, rep, inner); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
names = ("da", "db", "dgamma", "dbeta", "dW", "dc")
_noise = [torch.randn(2048, 2048, device="cuda") for _ in range(2)]
def noise(t):
    # other kernels between the launches: whatever they leave in LDS is what this launch finds there
    x = _noise[0] * (1.0 + t)
    y = (x @ _noise[1]).float()
    torch.sort(y.view(-1)[: 1 << 20])
    torch.cumsum(y, dim=1)
    return float(0)
def run():
    if os.environ.get("POISON"):
        ws.view(torch.float32).uniform_(-1e3, 1e3) if os.environ["POISON"] == "rand" else ws.fill_(0xFF)
    outs = (torch.full((a.shape[0], D), float("nan"), device="cuda"), torch.full((U, D), float("nan"), device="cuda"), torch.empty(3 * D, device="cuda"),
            torch.empty(3 * D, device="cuda"), torch.empty(D, 3 * D, device="cuda"), torch.empty(D, device="cuda"))
    _lib.check(lib.stage_cat3_bwd_dw(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     beta.data_ptr(), *[o.data_ptr() for o in outs], U, D, rep, inner, p, 77, ws.data_ptr(), wsb, st), "bwd")
    torch.cuda.synchronize()
    return outs
ref = run()
bad = {}
for t in range(int(os.environ.get("TRIALS", 200))):
    if os.environ.get("NOISE"): noise(t)
    cur = run()
    for nm, x, y in zip(names, ref, cur):
        if not torch.equal(x, y):
            d = (x != y)
            rows = d.view(d.shape[0], -1).any(-1).nonzero().flatten() if d.dim() > 1 else d.nonzero().flatten()
            bad.setdefault(nm, []).append((int(d.sum()), rows[:6].tolist(), float((x - y).abs().max()), float(x.abs().max())))
print("U", U, "rep", rep, "inner", inner, "launches that differ from the first:", {k: (len(v), v[:3]) for k, v in bad.items()} or "none")
