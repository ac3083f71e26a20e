"""stage_gemm_nt / stage_gemm_tn at small and medium shapes for the library named by STAGE_HIP_LIB (A/B of library builds)."""
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
for (M, N, K) in ((4800, 256, 256), (4800, 256, 768), (12800, 256, 256), (32768, 256, 768), (96000, 128, 128), (24000, 128, 384)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    a = t(lambda: lib.stage_gemm_nt(x.data_ptr(), None, w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st))
    b = t(lambda: lib.stage_gemm_tn(dy.data_ptr(), None, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st))
    out.append("%dx%d->%d nt %.1f tn %.1f" % (M, K, N, a, b))
print(os.path.basename(os.environ.get("STAGE_HIP_LIB", "default")), " | ".join(out))
