import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import ops
torch.manual_seed(0)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for (M, N, K) in [(960000, 128, 384), (960000, 128, 128), (240000, 300, 768), (240000, 128, 300), (96000, 300, 300)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * N * K
    a = t(lambda: ops.linear(x, w, b, relu=True))
    c = t(lambda: torch.relu_(torch.addmm(b, x, w.t())))
    d = t(lambda: torch.mm(dy.t(), x))          # dW via rocBLAS
    wt = w.t().contiguous()
    e_ = t(lambda: torch.mm(dy, w))               # dX via rocBLAS
    from tvqaplus_amd import _lib
    lib = _lib.load()
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f_ = t(lambda: lib.stage_gemm_tn(dy.data_ptr(), None, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st))
    dx = torch.empty(M, K, device="cuda")
    g_ = t(lambda: lib.stage_gemm_nt(dy.data_ptr(), None, wt.data_ptr(), None, None, dx.data_ptr(), M, K, N, 0, st))
    print("M=%d N=%d K=%d | fwd ours %.3f ms (%.0f TF) torch %.3f ms (%.0f TF) | dW ours %.3f (%.0f TF) torch %.3f (%.0f TF) | dX ours %.3f (%.0f) torch %.3f (%.0f)" % (
        M, N, K, a, fl / a / 1e9, c, fl / c / 1e9, f_, fl / f_ / 1e9, d, fl / d / 1e9, g_, fl / g_ / 1e9, e_, fl / e_ / 1e9))
