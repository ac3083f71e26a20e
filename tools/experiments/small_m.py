"""stage_gemm_nt / stage_gemm_tn at the QA-stream shapes (M = 3200) with the streaming kernels' row threshold from STAGE_GEMM_STREAM_MIN_M."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
out = []
def t(f, n=50):
    for _ in range(5): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n
for (M, N, K) in ((3200, 300, 768), (3200, 128, 300), (3200, 128, 128), (3200, 384, 128), (3200, 128, 384), (1600, 128, 128)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    a = t(lambda: lib.stage_gemm_nt(x.data_ptr(), None, w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st))
    b = t(lambda: lib.stage_gemm_tn(dy.data_ptr(), None, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st))
    ey = float((y.double() - x.double() @ w.double().t()).abs().max()); ew = float((dw.double() - dy.double().t() @ x.double()).abs().max())
    out.append("%dx%d->%d nt %.1f tn %.1f (err %.1e %.1e)" % (M, K, N, a, b, ey, ew))
print("MIN_M=%s" % os.environ.get("STAGE_GEMM_STREAM_MIN_M", "4096"), " | ".join(out))
