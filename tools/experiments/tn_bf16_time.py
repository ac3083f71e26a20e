"""bf16 weight-gradient GEMMs (stage_gemm_tn_bf16) at the stress config's shapes: time and effective bandwidth (algorithmic bytes =
M * (N + K) * 2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
shapes = [(1400000, 256, 768), (1400000, 256, 256), (960000, 256, 768), (960000, 256, 256), (960000, 128, 384), (96000, 256, 2048)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for M, N, K in shapes:
    dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, K, device="cuda").bfloat16()
    gate = torch.randn(M, N, device="cuda").bfloat16() if os.environ.get("GATE") else None
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_bf16_ws_bytes(M, N, K); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    f = lambda: lib.stage_gemm_tn_bf16(dy.data_ptr(), gate.data_ptr() if gate is not None else None, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st)
    for _ in range(3): assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    dyg = dy.float() * (gate.float() > 0) if gate is not None else dy.float()
    ref = (dyg.t() @ x.float())
    err = float((dw - ref).abs().max() / ref.abs().max())
    errb = float((db - dyg.sum(0)).abs().max() / dyg.sum(0).abs().max())
    print(f"M={M} N={N} K={K}: {ms*1e3:8.1f} us  {M*(N+K)*2/ms/1e6:7.1f} GB/s  {2*M*N*K/ms/1e9:6.1f} TFLOP/s  rel err dW {err:.1e} db {errb:.1e}", flush=True)
