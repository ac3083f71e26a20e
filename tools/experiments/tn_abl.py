"""Times of stage_gemm_tn at the weight-gradient shapes for the library named by STAGE_HIP_LIB (ablation builds: tools/build_variant.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
out = []
for (M, N, K) in ((960000, 128, 384), (240000, 300, 768), (960000, 128, 128), (240000, 128, 300)):
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    f = lambda: lib.stage_gemm_tn(dy.data_ptr(), None, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    ref = dy.double().t() @ x.double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    out.append("%dx%dT x%d %.0f us (err %.1e)" % (M, N, K, s.elapsed_time(e) * 100, err))
print(os.environ.get("STAGE_HIP_LIB", "default"), " | ".join(out))
