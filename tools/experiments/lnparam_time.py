"""Time of stage_gemm_nt_lnparam (LayerNorm parameter gradients inside the dX GEMM) against dX GEMM + LayerNorm backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib, ops
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
def t(f, n=10):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n
for (M, K, N, p) in ((240000, 768, 300, 0.1), (96000, 300, 300, 0.1)):
    x = torch.randn(M, K, device="cuda"); gamma = torch.randn(K, device="cuda").requires_grad_(True); beta = torch.randn(K, device="cuda").requires_grad_(True)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).requires_grad_(True); b = torch.randn(N, device="cuda").requires_grad_(True)
    g = torch.randn(M, N, device="cuda")
    def step(fused):
        if fused: h = ops.input_ln_linear(x, gamma, beta, w, b, p=p, seed=7)
        else:
            y, _ = ops.layernorm(x, gamma, beta, p=p, seed=7); h = ops.linear(y, w, b, relu=True)
        h.backward(g)
    print("M=%d K=%d N=%d: fused fwd+bwd %.0f us, unfused %.0f us" % (M, K, N, t(lambda: step(True)), t(lambda: step(False))))
