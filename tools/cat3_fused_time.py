"""The fused backward of LayerNorm([a, b, a*b]) -> dropout -> Linear -> ReLU (csrc/cat3_fused.hip) against the two kernels it
replaces (dX GEMM + LayerNorm backward) at the bench shapes: per-launch times (events) and max differences.
REP=300 (c2q: `a` broadcast over the frames) or REP=1 (concat_fc); P=0.1 dropout."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
lib = _lib.load()
dev = "cuda"
D = 128
rep = int(os.environ.get("REP", 300)); inner = 40 if rep > 1 else 1
G = int(os.environ.get("G", 80))
U = G * rep * inner if rep > 1 else int(os.environ.get("U", 960000))
p = float(os.environ.get("P", 0.1)); seed = 1234
g = torch.Generator().manual_seed(1)
a = torch.randn((U // rep) if rep > 1 else U, D, generator=g).to(dev)
b = torch.randn(U, D, generator=g).to(dev)
gamma = (1 + 0.1 * torch.randn(3 * D, generator=g)).to(dev); beta = (0.1 * torch.randn(3 * D, generator=g)).to(dev)
W = (0.08 * torch.randn(D, 3 * D, generator=g)).to(dev); bias = torch.zeros(D, device=dev)
st = torch.cuda.current_stream().cuda_stream
z = torch.empty(U, 3 * D, device=dev); mean = torch.empty(U, device=dev); rstd = torch.empty(U, device=dev)
_lib.check(lib.stage_cat3_layernorm_fwd(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), U, D, rep, inner, 1e-5, p, seed, st), "fwd")
y = torch.empty(U, D, device=dev); mask = torch.empty(D // 32, U, dtype=torch.int32, device=dev)
_lib.check(lib.stage_gemm_nt_mask(z.data_ptr(), None, W.data_ptr(), bias.data_ptr(), y.data_ptr(), mask.data_ptr(), U, D, 3 * D, 1, st), "gemm fwd")
dy = torch.randn(U, D, generator=g).to(dev)
Wt = W.t().contiguous()
dz = torch.empty(U, 3 * D, device=dev)
da1 = torch.empty_like(a); db1 = torch.empty_like(b); dg1 = torch.empty(3 * D, device=dev); dbt1 = torch.empty(3 * D, device=dev)
da2 = torch.empty_like(a); db2 = torch.empty_like(b); dg2 = torch.empty(3 * D, device=dev); dbt2 = torch.empty(3 * D, device=dev)
wsb = max(lib.stage_cat3_layernorm_bwd_reduced_ws_bytes(U, D, rep, inner) if rep > 1 else 0, lib.stage_ln_bwd_ws_bytes(3 * D),
          lib.stage_cat3_dx_ln_bwd_ws_bytes(U, D, rep, inner))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def unfused():
    _lib.check(lib.stage_gemm_nt_mask(dy.data_ptr(), mask.data_ptr(), Wt.data_ptr(), None, dz.data_ptr(), None, U, 3 * D, D, 0, st), "dx")
    if rep > 1:
        _lib.check(lib.stage_cat3_layernorm_bwd_reduced(dz.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                   da1.data_ptr(), db1.data_ptr(), dg1.data_ptr(), dbt1.data_ptr(), U, D, rep, inner, p, seed, ws.data_ptr(), wsb, st), "ln bwd")
    else:
        _lib.check(lib.stage_cat3_layernorm_bwd(dz.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                   da1.data_ptr(), db1.data_ptr(), dg1.data_ptr(), dbt1.data_ptr(), U, D, rep, inner, p, seed, ws.data_ptr(), wsb, st), "ln bwd")
def fused():
    _lib.check(lib.stage_cat3_dx_ln_bwd(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
               gamma.data_ptr(), da2.data_ptr(), db2.data_ptr(), dg2.data_ptr(), dbt2.data_ptr(), U, D, rep, inner, p, seed, ws.data_ptr(), wsb, st), "fused")
unfused(); fused(); torch.cuda.synchronize()
# independent check of d beta (torch): sum over rows of dz * keep / (1 - p), keep = where the forward's z is non-zero
keep = (z != 0).float() / (1.0 - p) if p > 0 else torch.ones_like(z)
dz_t = torch.matmul(dy * (y > 0).float(), W) * keep
ref_b = dz_t.double().sum(0)
print("dbeta vs torch: unfused %.3e fused %.3e (scale %.3e)" % (float((dbt1.double() - ref_b).abs().max()), float((dbt2.double() - ref_b).abs().max()), float(ref_b.abs().max())))
dd = (dbt2.double() - ref_b).abs()
bad = (dd > 1e-2 * float(ref_b.abs().max())).nonzero().flatten().tolist()
print("columns with a wrong d beta:", len(bad), bad[:48])
del keep, dz_t
for nm, x, y2 in (("da", da1, da2), ("db", db1, db2), ("dgamma", dg1, dg2), ("dbeta", dbt1, dbt2)):
    print("%-7s max|diff| %.3e  of scale %.3e" % (nm, float((x - y2).abs().max()), float(x.abs().max())))
def timeit(fn, name, reps=10):
    for _ in range(2): fn()
    cs = torch.cuda.current_stream()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(cs); fn(); e.record(cs)
    torch.cuda.synchronize()
    t = [s.elapsed_time(e) * 1e3 for s, e in ev]
    print("%s: avg %.1f min %.1f us" % (name, sum(t) / len(t), min(t)))
timeit(unfused, "dX GEMM + LayerNorm backward"); timeit(fused, "fused")
# ---- forward: LayerNorm + dropout + Linear + ReLU in one pass against the two kernels it replaces ----
z2 = torch.empty_like(z); mean2 = torch.empty_like(mean); rstd2 = torch.empty_like(rstd); y2 = torch.empty_like(y); mask2 = torch.empty_like(mask)
fwsb = lib.stage_cat3_ln_gemm_fwd_ws_bytes(); fws = torch.empty(fwsb, dtype=torch.uint8, device=dev)
def fwd_unfused():
    _lib.check(lib.stage_cat3_layernorm_fwd(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), U, D, rep, inner, 1e-5, p, seed, st), "fwd")
    _lib.check(lib.stage_gemm_nt_mask(z.data_ptr(), None, W.data_ptr(), bias.data_ptr(), y.data_ptr(), mask.data_ptr(), U, D, 3 * D, 1, st), "gemm fwd")
def fwd_fused():
    _lib.check(lib.stage_cat3_ln_gemm_fwd(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), bias.data_ptr(), z2.data_ptr(), mean2.data_ptr(),
               rstd2.data_ptr(), y2.data_ptr(), mask2.data_ptr(), U, D, rep, inner, 1e-5, p, seed, fws.data_ptr(), fwsb, st), "fused fwd")
fwd_unfused(); fwd_fused(); torch.cuda.synchronize()
print("forward: max|y - y'| %.3e of %.3e, z equal up to %.3e" % (float((y - y2).abs().max()), float(y.abs().max()), float((z - z2).abs().max())))
timeit(fwd_unfused, "LayerNorm forward + GEMM"); timeit(fwd_fused, "fused forward")
os.environ.setdefault("STAGE_CAT3_DW", "1")
# ---- round 6: the backward with the Linear's gradients inside (csrc/cat3_bwd_dw.hip) and the forward without the z store ----
if lib.stage_cat3_bwd_dw_supported(U, D, rep, inner):
    da3 = torch.empty_like(a); db3 = torch.empty_like(b); dg3 = torch.empty(3 * D, device=dev); dbt3 = torch.empty(3 * D, device=dev)
    dW3 = torch.empty(D, 3 * D, device=dev); dc3 = torch.empty(D, device=dev)
    dW0 = torch.empty(D, 3 * D, device=dev); dc0 = torch.empty(D, device=dev)
    wsb3 = lib.stage_cat3_bwd_dw_ws_bytes(U, D, rep, inner); ws3 = torch.empty(wsb3, dtype=torch.uint8, device=dev)
    wsb0 = lib.stage_gemm_tn_ws_bytes(U, D, 3 * D); ws0 = torch.empty(wsb0, dtype=torch.uint8, device=dev)
    def dw_inside():
        _lib.check(lib.stage_cat3_bwd_dw(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   gamma.data_ptr(), beta.data_ptr(), da3.data_ptr(), db3.data_ptr(), dg3.data_ptr(), dbt3.data_ptr(), dW3.data_ptr(), dc3.data_ptr(),
                   U, D, rep, inner, p, seed, ws3.data_ptr(), wsb3, st), "dw inside")
    def tn_on_z():
        _lib.check(lib.stage_gemm_tn_mask(dy.data_ptr(), mask.data_ptr(), z.data_ptr(), dW0.data_ptr(), dc0.data_ptr(), U, D, 3 * D, ws0.data_ptr(), wsb0, st), "tn")
    fwd_unfused(); dw_inside(); tn_on_z(); torch.cuda.synchronize()
    for nm, x, y3 in (("da", da2, da3), ("db", db2, db3), ("dgamma", dg2, dg3), ("dbeta", dbt2, dbt3), ("dW", dW0, dW3), ("dc", dc0, dc3)):
        print("dW-inside %-7s max|diff| %.3e  of scale %.3e" % (nm, float((x - y3).abs().max()), float(x.abs().max())))
    ref = (dy * (((mask.unsqueeze(-1) >> torch.arange(32, device=dev, dtype=torch.int32)) & 1).permute(1, 0, 2).reshape(U, D).float())).double().t() @ z.double()
    print("dW vs fp64: TN on z %.3e, inside %.3e (scale %.3e)" % (float((dW0.double() - ref).abs().max()), float((dW3.double() - ref).abs().max()), float(ref.abs().max())))
    del ref
    timeit(tn_on_z, "weight-gradient GEMM on z"); timeit(dw_inside, "backward with dW inside (incl. weight image + reductions)")
    def fwd_fused_noz():
        _lib.check(lib.stage_cat3_ln_gemm_fwd(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), bias.data_ptr(), None, mean2.data_ptr(),
                   rstd2.data_ptr(), y2.data_ptr(), mask2.data_ptr(), U, D, rep, inner, 1e-5, p, seed, fws.data_ptr(), fwsb, st), "fused fwd no z")
    timeit(fwd_fused_noz, "fused forward without the z store")
    if os.environ.get("CW_PROF"):                         # developer build -DCW_PROF: cycles per phase of workgroup 0
        import ctypes
        raw = ctypes.CDLL(_lib.LIB_PATH)
        buf = (ctypes.c_ulonglong * 32)()
        dw_inside(); torch.cuda.synchronize()
        raw.stage_cw_prof(buf)
        names_w = ["prologue", "wait Ba", "z + dW", "DMA wait", "wait Bb", "geom + staging", "hashes", "DMA issue"]
        names_e = ["prologue", "geom", "wait Ba", "dX product", "LN bwd 1", "wait Bb", "LN bwd 2"]
        tiles = (U // 32 + 255) // 256
        print("W wave (cycles per tile, ~%d tiles):" % tiles, ", ".join("%s %d" % (n, buf[i] // tiles) for i, n in enumerate(names_w)), " total", sum(buf[:8]) // tiles)
        print("E wave (cycles per tile):", ", ".join("%s %d" % (n, buf[16 + i] // tiles) for i, n in enumerate(names_e)), " total", sum(buf[16:23]) // tiles)
