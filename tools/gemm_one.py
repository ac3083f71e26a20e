"""One GEMM shape launched 13 times through the C ABI (for rocprofv3 PMC passes: bash tools/pmc_run.sh NAME KERNEL python tools/gemm_one.py M N K nt|tn)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
M, N, K, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
torch.manual_seed(0)
if kind == "nt":
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; y = torch.empty(M, N, device="cuda")
    f = lambda: lib.stage_gemm_nt(x.data_ptr(), None, w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st)
else:
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    f = lambda: lib.stage_gemm_tn(dy.data_ptr(), None, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st)
for _ in range(13): assert f() == 0
torch.cuda.synchronize()
