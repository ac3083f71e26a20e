"""Ragged shapes of the streaming NT GEMM (K, N not multiples of 128) against fp64: plain / relu+mask / dX through the mask / fp32 gate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
def rep(tag, got, ref):
    e = (got.double() - ref).abs(); rowmax = e.max(dim=1).values; bad = (rowmax > 1e-4).nonzero().flatten()
    print("   %-14s max %.3e mean %.3e bad rows %d %s" % (tag, float(e.max()), float(e.mean()), bad.numel(), bad[:6].tolist()))
for (M, N, K) in [(15000, 300, 768), (15000, 128, 300), (6000, 300, 300), (60000, 128, 300), (60000, 300, 128), (15000, 768, 300), (6000, 128, 300), (59990, 128, 128), (60001, 384, 128)]:
    print("M=%d N=%d K=%d mask_supported=%d" % (M, N, K, lib.stage_gemm_mask_supported(M, N, K)))
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    _lib.check(lib.stage_gemm_nt(x.data_ptr(), None, w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st), "nt")
    rep("plain", y, x.double() @ w.double().t() + b.double())
    g = torch.randn(M, K, device="cuda")
    _lib.check(lib.stage_gemm_nt(x.data_ptr(), g.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 0, st), "nt")
    rep("fp32 gate", y, (x.double() * (g > 0).double()) @ w.double().t() + b.double())
    if lib.stage_gemm_mask_supported(M, N, K):
        mask = torch.zeros((N + 31) // 32, M, dtype=torch.int32, device="cuda")
        _lib.check(lib.stage_gemm_nt_mask(x.data_ptr(), None, w.data_ptr(), b.data_ptr(), y.data_ptr(), mask.data_ptr(), M, N, K, 1, st), "ntm")
        rep("relu+mask fwd", y, (x.double() @ w.double().t() + b.double()).clamp(min=0))
        bits = torch.stack([(mask[j // 32] >> (j % 32)) & 1 for j in range(N)], dim=1).bool()
        print("   mask mismatches", (bits != (y > 0)).sum().item())
        dy = torch.randn(M, N, device="cuda"); wt = w.t().contiguous(); dx = torch.empty(M, K, device="cuda")
        if lib.stage_gemm_mask_supported(M, K, N) or True:
            rc = lib.stage_gemm_nt_mask(dy.data_ptr(), mask.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), None, M, K, N, 0, st)
            if rc == 0: rep("dX via mask", dx, (dy.double() * (y > 0).double()) @ w.double())
            else: print("   dX via mask: rc", rc)
