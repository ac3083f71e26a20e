"""Event times of stage_gemm_tn_mask (the weight-gradient GEMM with the ReLU bit mask) at the bench's ragged row counts, for the library
named by STAGE_HIP_LIB (tools/build_variant.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
out = []
g = torch.Generator().manual_seed(3)
for (M, N, K) in ((630704, 128, 384), (960000, 128, 384), (630704, 128, 128), (158000, 128, 128)):
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (N // 32, M), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    f = lambda: lib.stage_gemm_tn_mask(dy.data_ptr(), mask.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st)
    for _ in range(3): assert f() == 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for s, e in ev:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    bits = ((mask.unsqueeze(-1) >> torch.arange(32, device="cuda", dtype=torch.int32)) & 1).permute(1, 0, 2).reshape(M, N)
    ref = (dy.double() * bits.double()).t() @ x.double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    out.append("%dx%dT x%d med %.0f min %.0f us (err %.1e)" % (M, N, K, t[len(t) // 2], t[0], err))
    del dy, x, mask, bits, ref
print(os.environ.get("STAGE_HIP_LIB", "default"), " | ".join(out))
