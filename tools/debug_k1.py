import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from conftest import Fixture
from tvqaplus_amd import ops
from oracle import stage_oracle as O

torch.set_printoptions(precision=4, linewidth=200)
for name in sys.argv[1:] or ["k1_small"]:
    fx = Fixture(name)
    t = lambda k: torch.from_numpy(fx[k])
    C, Q, cm, qm = t("C"), t("Q"), t("c_mask"), t("q_mask")
    N, NA, _, Lqa, D = C.shape
    _, _, Li, Lr, _ = Q.shape
    print(name, "N NA Li Lqa Lr D", N, NA, Li, Lqa, Lr, D)
    A, S, Sn = ops.structured_attention(C.view(N, NA, Lqa, D).cuda(), Q.view(N, Li, Lr, D).cuda(),
                                        cm.view(N, NA, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), float(fx["scale"]))
    torch.cuda.synchronize()
    for nm, got, exp in (("A", A, t("A")), ("S", S, t("S")), ("Sn", Sn, t("S_norm"))):
        got = got.cpu()
        nan = torch.isnan(got)
        err = (got - exp).abs() / (1 + exp.abs())
        err[nan] = 0
        print(nm, "nan count", int(nan.sum()), "of", got.numel(), "max rel err (non-nan)", float(err.max()))
        if nan.any():
            idx = torch.nonzero(nan)
            print("  first nan idx", idx[:5].tolist(), " last", idx[-3:].tolist())
        bad = torch.nonzero(err > 1e-3)
        if len(bad):
            print("  bad count", len(bad), "first", bad[:6].tolist())
            i = tuple(bad[0].tolist())
            print("  got", float(got[i]), "exp", float(exp[i]))
    print("Sn[0,0,0]:\n", Sn[0, 0, 0].cpu(), "\nexp\n", t("S_norm")[0, 0, 0])
    print("S[0,0,0]:\n", S[0, 0, 0].cpu(), "\nexp\n", t("S")[0, 0, 0])
    print("A[0,0,0,0]:\n", A[0, 0, 0, 0].cpu(), "\nexp\n", t("A")[0, 0, 0, 0])
