"""Error of stage_gemm_nt against an fp64 product (max abs error / max |y|), with a torch fp32 matmul for comparison."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(60000, 128, 128), (60000, 384, 128), (60000, 128, 384), (60000, 300, 768), (4100, 128, 128)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    _lib.check(lib.stage_gemm_nt(x.data_ptr(), None, w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st), "nt")
    ref = (x.double() @ w.double().t() + b.double())
    e = (y.double() - ref).abs()
    t = (torch.addmm(b, x, w.t()).double() - ref).abs()
    rowmax = e.max(dim=1).values
    bad = (rowmax > 1e-4).nonzero().flatten()
    print("M=%d N=%d K=%d: ours max %.3e mean %.3e | torch fp32 max %.3e mean %.3e | rows with err>1e-4: %d %s" % (
        M, N, K, float(e.max()), float(e.mean()), float(t.max()), float(t.mean()), bad.numel(), bad[:8].tolist()))
print("--- with fp32 gate (x kept where gate > 0), residual, relu ---")
for (M, N, K) in [(60000, 128, 128), (60000, 384, 128), (60000, 128, 384)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
    g = torch.randn(M, K, device="cuda"); r = torch.randn(M, N, device="cuda")
    for name, gate, res, relu in (("gate", g, None, 0), ("gate+res", g, r, 0), ("res", None, r, 0), ("relu", None, None, 1)):
        y = torch.empty(M, N, device="cuda")
        _lib.check(lib.stage_gemm_nt(x.data_ptr(), gate.data_ptr() if gate is not None else None, w.data_ptr(), None,
                                     res.data_ptr() if res is not None else None, y.data_ptr(), M, N, K, relu, st), "nt")
        xx = x.double() * (gate > 0).double() if gate is not None else x.double()
        ref = xx @ w.double().t()
        if relu: ref = ref.clamp(min=0)
        if res is not None: ref = ref + res.double()
        e = (y.double() - ref).abs()
        rowmax = e.max(dim=1).values
        bad = (rowmax > 1e-4).nonzero().flatten()
        print("M=%d N=%d K=%d %-9s max %.3e mean %.3e bad rows %d %s" % (M, N, K, name, float(e.max()), float(e.mean()), bad.numel(), bad[:6].tolist()))
# bit-mask path: forward emits the mask, dX gates with it
print("--- mask path ---")
for (M, N, K) in [(60000, 128, 128), (60000, 128, 384), (60000, 384, 128)]:
    if not lib.stage_gemm_mask_supported(M, N, K): print("mask unsupported", M, N, K); continue
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda"); mask = torch.zeros((N + 31) // 32, M, dtype=torch.int32, device="cuda")
    _lib.check(lib.stage_gemm_nt_mask(x.data_ptr(), None, w.data_ptr(), b.data_ptr(), y.data_ptr(), mask.data_ptr(), M, N, K, 1, st), "ntm")
    ref = (x.double() @ w.double().t() + b.double()).clamp(min=0)
    e = (y.double() - ref).abs()
    bits = torch.stack([(mask[j // 32] >> (j % 32)) & 1 for j in range(N)], dim=1).bool()
    mism = (bits != (y > 0)).sum().item()
    # dX through the mask: dx = (dy * [y>0]) @ w
    dy = torch.randn(M, N, device="cuda"); wt = w.t().contiguous(); dx = torch.empty(M, K, device="cuda")
    if lib.stage_gemm_mask_supported(M, K, N):
        _lib.check(lib.stage_gemm_nt_mask(dy.data_ptr(), mask.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), None, M, K, N, 0, st), "dx")
        refx = (dy.double() * (y > 0).double()) @ w.double()
        ex = (dx.double() - refx).abs()
        rowmax = ex.max(dim=1).values
        bad = (rowmax > 1e-4).nonzero().flatten()
        print("M=%d N=%d K=%d fwd max %.3e mask mismatches %d | dX max %.3e mean %.3e bad rows %d %s" % (M, N, K, float(e.max()), mism, float(ex.max()), float(ex.mean()), bad.numel(), bad[:6].tolist()))
    else:
        print("M=%d N=%d K=%d fwd max %.3e mask mismatches %d | dX mask unsupported" % (M, N, K, float(e.max()), mism))
