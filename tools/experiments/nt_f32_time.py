"""fp32 Linear forward / dX GEMMs (stage_gemm_nt: Y[M,N] = X[M,K] . W[N,K]^T) at the headline step's shapes: time, effective bandwidth
(algorithmic bytes = M * (N + K) * 4) and the error against an fp64 product."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
shapes = [(960000, 128, 128), (630704, 128, 384), (630704, 384, 128), (960000, 128, 384), (240000, 128, 128), (96000, 128, 300)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
    bias = torch.randn(N, device="cuda")
    gate = torch.randn(M, K, device="cuda") if os.environ.get("GATE") else None
    y = torch.empty(M, N, device="cuda")
    f = lambda: lib.stage_gemm_nt(x.data_ptr(), gate.data_ptr() if gate is not None else None, w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), M, N, K, 1, st)
    for _ in range(3): assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    n = min(M, 40000)
    xg = (x[-n:] * (gate[-n:] > 0) if gate is not None else x[-n:]).double()
    ref = (xg @ w.double().t() + bias.double()).relu()
    err = float((y[-n:].double() - ref).abs().max() / ref.abs().max())
    byt = M * (N + K) * 4 + (M * K * 4 if gate is not None else 0)
    print(f"M={M} N={N} K={K}: {ms*1e3:8.1f} us  {byt/ms/1e6:7.1f} GB/s  rel err {err:.1e}", flush=True)
