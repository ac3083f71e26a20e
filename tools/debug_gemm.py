import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
M, N, K = 4100, 128, 128
dy = torch.randn(M, N, device="cuda"); y = torch.relu(torch.randn(M, N, device="cuda")); wt = (torch.randn(K, N, device="cuda") / 11.3).contiguous()
st = torch.cuda.current_stream().cuda_stream
out = torch.empty(M, K, device="cuda")
for gate in (None, y):
    lib.stage_gemm_nt_bf16x3 = lib.stage_gemm_nt  # default path is bf16x3
    lib.stage_gemm_nt(dy.data_ptr(), None if gate is None else gate.data_ptr(), wt.data_ptr(), None, None, out.data_ptr(), M, K, N, 0, st)
    torch.cuda.synchronize()
    x = dy if gate is None else dy * (gate > 0)
    ref = x.double() @ wt.double().t()
    err = (out.double() - ref).abs()
    print("gate", gate is not None, "max err", float(err.max()), "ref max", float(ref.abs().max()))
    bad = (err > 1e-3).nonzero()
    print(" bad count", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:10], "...", sorted(set(bad[:, 0].tolist()))[-5:], "cols", sorted(set(bad[:, 1].tolist()))[:8])
