"""K1 backward: error of the fused kernel (fp16-pair products) and of the three-kernel path against an fp64 evaluation of the oracle."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import stage_oracle as O
from tvqaplus_amd import ops
from tvqaplus_amd.synth import make_batch
for (N, Li, Lr, Lqa) in ((2, 24, 20, 40), (2, 16, 50, 40)):
    D = 128
    g = torch.Generator().manual_seed(7)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=3)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g); Q = torch.randn(N, 1, Li, Lr, D, generator=g) * 2
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    gA = torch.randn(N, 5, Li, Lqa, D, generator=g) * torch.exp(2 * torch.randn(N, 5, Li, Lqa, 1, generator=g))
    Cc, Qc = C.double().requires_grad_(), Q.double().requires_grad_()
    Ao, So, _, _ = O.structured_attention(Cc, Qc, cm.double(), qm.double(), 10.0)
    (Ao * gA.double()).sum().backward()
    def run(unfused):
        old = ops._K1_BWD_UNFUSED; ops._K1_BWD_UNFUSED = unfused
        try:
            Cd = C.view(N, 5, Lqa, D).cuda().requires_grad_(); Qd = Q.view(N, Li, Lr, D).cuda().requires_grad_()
            A, S, _ = ops.structured_attention(Cd, Qd, cm.view(N, 5, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), 10.0)
            (A * gA.cuda()).sum().backward()
            return Cd.grad.cpu().double(), Qd.grad.cpu().double()
        finally:
            ops._K1_BWD_UNFUSED = old
    for name, unf in (("fused", False), ("three-kernel", True)):
        dC, dQ = run(unf)
        eC = float((dC.view_as(Cc.grad) - Cc.grad).abs().max() / Cc.grad.abs().max())
        eQ = float((dQ.view_as(Qc.grad) - Qc.grad).abs().max() / Qc.grad.abs().max())
        rC = float(((dC.view_as(Cc.grad) - Cc.grad) ** 2).mean().sqrt() / (Cc.grad ** 2).mean().sqrt())
        rQ = float(((dQ.view_as(Qc.grad) - Qc.grad) ** 2).mean().sqrt() / (Qc.grad ** 2).mean().sqrt())
        print("Lr=%d %-13s dC max %.2e rms %.2e | dQ max %.2e rms %.2e (relative to the gradient's max / rms, fp64 reference)" % (Lr, name, eC, rC, eQ, rQ))
