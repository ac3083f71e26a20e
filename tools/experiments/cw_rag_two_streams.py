"""Two instances of stage_cat3_bwd_dw_rag on two streams at once (what the model does at stream levels >= 2: the subtitle and the video
down-projection backward), each on its own buffers: do they reproduce their solo results bit for bit?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib, ragged
lib = _lib.load()
D = 128
def make(seed, full):
    rng = np.random.default_rng(seed)
    N, NA, Li, Lqa = 4, 5, 48, 40
    qa = np.zeros((N, NA, Lqa), bool)
    for n in range(N):
        for ai in range(NA):
            qa[n, ai, :(Lqa if full else rng.integers(0, Lqa + 1))] = True
    qa[0, 0, :] = True
    fl = rng.random((N, Li)) < 0.8
    lay = ragged.RaggedLayout(ragged.RaggedTables(qa, fl, Lqa if full else 4), torch.device("cuda"))
    U, Fc, G = lay.U, lay.Fc, N * NA
    g = torch.Generator().manual_seed(seed)
    t = dict(lay=lay, U=U, Fc=Fc, G=G, Li=Li, Lqa=Lqa,
             dy=torch.randn(U, D, generator=g).cuda(), W=(0.08 * torch.randn(D, 3 * D, generator=g)).cuda(), bias=torch.zeros(D).cuda(),
             a=torch.randn(G * Lqa, D, generator=g).cuda(), b=torch.randn(Fc, D, generator=g).cuda(),
             gamma=(1 + 0.1 * torch.randn(3 * D, generator=g)).cuda(), beta=(0.1 * torch.randn(3 * D, generator=g)).cuda())
    t["mean"] = torch.empty(U, device="cuda"); t["rstd"] = torch.empty(U, device="cuda"); y = torch.empty(U, D, device="cuda")
    t["mask"] = torch.zeros(D // 32, U, dtype=torch.int32, device="cuda")
    fwsb = lib.stage_cat3_ln_gemm_fwd_ws_bytes(); fws = torch.empty(fwsb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.stage_cat3_ln_gemm_fwd_rag(t["a"].data_ptr(), t["b"].data_ptr(), t["gamma"].data_ptr(), t["beta"].data_ptr(), t["W"].data_ptr(), t["bias"].data_ptr(), None,
                                              t["mean"].data_ptr(), t["rstd"].data_ptr(), y.data_ptr(), t["mask"].data_ptr(), lay.rowinfo.data_ptr(), U, G * Lqa, Fc, D, 1e-5, 0.1,
                                              4321, fws.data_ptr(), fwsb, st), "fwd")
    wsb = lib.stage_cat3_bwd_dw_rag_ws_bytes(G, Lqa); t["ws"] = torch.empty(wsb, dtype=torch.uint8, device="cuda"); t["wsb"] = wsb
    torch.cuda.synchronize()
    return t
def launch(t, stream):
    b_in = t["b"].clone()
    outs = (torch.full((t["G"] * t["Lqa"], D), float("nan"), device="cuda"), b_in, torch.empty(3 * D, device="cuda"), torch.empty(3 * D, device="cuda"),
            torch.empty(D, 3 * D, device="cuda"), torch.empty(D, device="cuda"))
    lay = t["lay"]
    _lib.check(lib.stage_cat3_bwd_dw_rag(t["dy"].data_ptr(), t["mask"].data_ptr(), t["W"].data_ptr(), t["a"].data_ptr(), b_in.data_ptr(), t["mean"].data_ptr(), t["rstd"].data_ptr(),
                                         t["gamma"].data_ptr(), t["beta"].data_ptr(), *[o.data_ptr() for o in outs], lay.gdesc.data_ptr(), lay.wtab.data_ptr(), t["U"], t["Fc"], D,
                                         t["G"], t["Li"], t["Lqa"], 0.1, 4321, t["ws"].data_ptr(), t["wsb"], stream.cuda_stream), "bwd")
    return outs
full = bool(os.environ.get("FULL"))
A, B = make(3, full), make(7, full)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1): refA = launch(A, s1)
torch.cuda.synchronize()
with torch.cuda.stream(s2): refB = launch(B, s2)
torch.cuda.synchronize()
names = ("da", "db", "dgamma", "dbeta", "dW", "dc")
bad = {}
for t in range(int(os.environ.get("TRIALS", 400))):
    with torch.cuda.stream(s1): oa = launch(A, s1)
    with torch.cuda.stream(s2): ob = launch(B, s2)
    torch.cuda.synchronize()
    for tag, ref, cur in (("A", refA, oa), ("B", refB, ob)):
        for nm, x, y in zip(names, ref, cur):
            if not torch.equal(x, y):
                bad.setdefault(tag + ":" + nm, []).append((int((x != y).sum()), float((x - y).abs().max())))
print("two concurrent instances, launches that differ from the solo run:", {k: (len(v), v[:2]) for k, v in bad.items()} or "none")
