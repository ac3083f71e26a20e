"""stage_str_attn_long_bwd_qm on random bf16 operands: dQraw / dQn of the 32-region bf16 kernel against fp64 products (and, with
STAGE_LONG_DQ16=1, of the 16-region fp32-MFMA kernel).  python tools/experiments/long_dq_check.py [N Li Lr Lqa]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
N, Li, Lr, Lqa = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 3, 512, 40)))
NA, D = 5, 256
g = torch.Generator(device="cuda").manual_seed(1)
bf = torch.bfloat16
dA = torch.randn(N, NA, Li, Lqa, D, device="cuda", generator=g).to(bf)
A = torch.randn(N, NA, Li, Lqa, D, device="cuda", generator=g).to(bf)
Cn = torch.randn(N, NA, Lqa, D, device="cuda", generator=g).to(bf)
Q = torch.randn(N, Li, Lr, D, device="cuda", generator=g).to(bf)
Qn = torch.randn(N, Li, Lr, D, device="cuda", generator=g).to(bf)
qm = torch.ones(N, Li, Lr, device="cuda")
lens = torch.randint(1, Lr + 1, (N, Li), device="cuda", generator=g)
qm = (torch.arange(Lr, device="cuda").view(1, 1, Lr) < lens.unsqueeze(-1)).float()
Sn = torch.softmax(torch.randn(N, NA, Li, Lqa, Lr, device="cuda", generator=g) - 1e10 * (1 - qm.view(N, 1, Li, 1, Lr)), -1) * qm.view(N, 1, Li, 1, Lr)
dS_ws = torch.empty_like(Sn); dQ = torch.empty(N, Li, Lr, D, device="cuda"); dQn = torch.empty_like(dQ); dCn = torch.empty(N, NA, Lqa, D, device="cuda")
wsb = lib.stage_str_attn_long_bwd_qm_ws_bytes(N, NA, Li, Lqa, D); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
rc = lib.stage_str_attn_long_bwd_qm(dA.data_ptr(), A.data_ptr(), None, Cn.data_ptr(), Q.data_ptr(), Qn.data_ptr(), Sn.data_ptr(), qm.data_ptr(),
                                    dS_ws.data_ptr(), dQ.data_ptr(), dQn.data_ptr(), dCn.data_ptr(), N, NA, Li, Lqa, Lr, D, 1.0, 1, ws.data_ptr(), wsb, st)
assert rc == 0, rc
torch.cuda.synchronize()
# dQraw[n,i,r,:] = sum_{a,l} P[n,a,i,l,r] dA[n,a,i,l,:] ; dQn[n,i,r,:] = sum dS[n,a,i,l,r] Cn[n,a,l,:]
ref_raw = torch.einsum("nailr,naild->nird", Sn.double(), dA.double())
ref_n = torch.einsum("nailr,nald->nird", dS_ws.double(), Cn.double())
for name, got, ref in (("dQraw", dQ, ref_raw), ("dQn", dQn, ref_n)):
    err = (got.double() - ref).abs()
    print(name, "max err", float(err.max()), "scale", float(ref.abs().max()), "nan", bool(torch.isnan(got).any()))
    bad = (err > 1e-3 * float(ref.abs().max())).nonzero()
    if bad.numel():
        print("  first bad (n,i,r,d):", bad[:6].tolist(), " got/ref:", [(float(got[tuple(b)]), float(ref[tuple(b)])) for b in bad[:3]])
        rs = sorted(set(int(b[2]) for b in bad[:2000])); ds = sorted(set(int(b[3]) for b in bad[:2000]))
        print("  bad regions (sample):", rs[:20], " bad d (sample):", ds[:20], "count", bad.shape[0], "of", err.numel())
