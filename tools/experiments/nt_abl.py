"""Times of stage_gemm_nt at the 3D->D shapes for the library named by STAGE_HIP_LIB (ablation builds: tools/build_variant.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
out = []
for (M, N, K) in ((960000, 128, 384), (960000, 384, 128), (960000, 128, 128)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda")
    f = lambda: lib.stage_gemm_nt(x.data_ptr(), None, w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    out.append("%dx%d->%d %.0f us" % (M, K, N, s.elapsed_time(e) * 100))
print(os.environ.get("STAGE_HIP_LIB", "default"), " | ".join(out))
