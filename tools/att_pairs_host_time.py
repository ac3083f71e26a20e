"""Host time of build_att_pairs on the synthetic bench batch (isolated, no device work)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd.synth import make_batch, make_opt
from tvqaplus_amd import att_host
b = make_batch(N=16, Li=300, Lr=20, Lw=2, Lqa=40, wd_size=4, vfeat_size=4, seed=2018, att_imgs=4, att_words=3)
opt = make_opt(use_sup_att=True)
for nt in (None, 1):
    if nt: torch.set_num_threads(nt)
    for _ in range(3): att_host.build_att_pairs(opt, b, None, 5)
    t = time.time()
    for _ in range(20): p, n = att_host.build_att_pairs(opt, b, None, 5)
    print("threads", torch.get_num_threads(), "build_att_pairs ms %.2f" % ((time.time() - t) / 20 * 1e3), p.shape, "cpus", os.cpu_count())
