"""bf16 Linear forward / dX GEMMs (stage_gemm_nt_bf16: Y[M,N] = X[M,K] . W[N,K]^T, fp32 weight) at the stress config's shapes: time and
effective bandwidth (algorithmic bytes = M * (N + K) * 2, + M * K * 2 with a gate on X)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
shapes = [(1400000, 256, 768), (1400000, 256, 256), (1400000, 768, 256), (960000, 256, 768), (960000, 256, 256), (960000, 768, 256), (96000, 256, 2048)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda") * 0.05
    bias = torch.randn(N, device="cuda")
    gate = torch.randn(M, K, device="cuda").bfloat16() if os.environ.get("GATE") else None
    relu = int(os.environ.get("RELU", 0))
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: lib.stage_gemm_nt_bf16(x.data_ptr(), gate.data_ptr() if gate is not None else None, w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), M, N, K, relu, st)
    for _ in range(3): assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    n = min(M, 50000)
    xg = x[:n].float() * (gate[:n].float() > 0) if gate is not None else x[:n].float()
    ref = xg @ w.bfloat16().float().t() + bias
    if relu: ref = ref.relu()
    err = float((y[:n].float() - ref).abs().max() / ref.abs().max())
    byt = M * (N + K) * 2 + (M * K * 2 if gate is not None else 0)
    print(f"M={M} N={N} K={K}: {ms*1e3:8.1f} us  {byt/ms/1e6:7.1f} GB/s  {2*M*N*K/ms/1e9:6.1f} TFLOP/s  rel err {err:.1e}", flush=True)
