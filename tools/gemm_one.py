"""Run one GEMM shape a few times (for PMC / trace runs).  argv: M N K [nt|tn|dx] [gate]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])   # experiment builds
lib = _lib.load()
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
gate = len(sys.argv) > 5
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda"); dy = torch.randn(M, N, device="cuda")
g = torch.randn(M, N if kind == "tn" else K, device="cuda") if gate else None
gp = g.data_ptr() if gate else None
dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
def run():
    if kind == "nt":
        _lib.check(lib.stage_gemm_nt(x.data_ptr(), gp, w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, 1, st), "nt")
    else:
        _lib.check(lib.stage_gemm_tn(dy.data_ptr(), gp, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st), "tn")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): run()
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print("%s M=%d N=%d K=%d gate=%s: %.1f us" % (kind, M, N, K, gate, s.elapsed_time(e) * 100))
