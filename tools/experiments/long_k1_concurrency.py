"""Is the long-row K1 forward (csrc/str_attn_long.hip) deterministic when it shares the chip with another kernel?  Solo outputs are the
reference; then the video-shape and the subtitle-shape launch run on two streams at once, repeatedly, and every output is compared
bit for bit with its solo run.  (Round 6: the level-4 branch streams differ from run to run on the stress config.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import ops
torch.manual_seed(0)
N, NA, Li, Lqa, D = int(os.environ.get("N", 4)), 5, int(os.environ.get("LI", 100)), 40, 256
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
def mk(Lr):
    C = torch.randn(N, NA, Lqa, D).cuda().to(dt); Q = torch.randn(N, Li, Lr, D).cuda().to(dt)
    cm = (torch.rand(N, NA, Lqa) < 0.8).float().cuda(); qm = (torch.rand(N, Li, Lr) < 0.7).float().cuda()
    cm[:, :, 0] = 1; qm[:, :, 0] = 1
    return C, Q, cm, qm
shapes = {"vid": mk(20), "sub": mk(int(os.environ.get("LRS", 512)))}
p = float(os.environ.get("P", 0.1))
def run(name):
    C, Q, cm, qm = shapes[name]
    with torch.no_grad():
        return ops.structured_attention_long(C, Q, cm, qm, 10.0, p, 11, 12)
ref = {k: [t.clone() for t in run(k)] for k in shapes}
torch.cuda.synchronize()
# solo repeatability first
for k in shapes:
    for _ in range(3):
        out = run(k); torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, ref[k])), ("solo run differs", k)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = {"vid": 0, "sub": 0}
trials = int(os.environ.get("TRIALS", 30))
for t in range(trials):
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        o_sub = run("sub")
    with torch.cuda.stream(s2):
        o_vid = [run("vid") for _ in range(4)]
    torch.cuda.synchronize()
    if not all(torch.equal(a, b) for a, b in zip(o_sub, ref["sub"])):
        bad["sub"] += 1
    for o in o_vid:
        if not all(torch.equal(a, b) for a, b in zip(o, ref["vid"])):
            bad["vid"] += 1
            d = [float((a.float() - b.float()).abs().max()) for a, b in zip(o, ref["vid"])]
            if bad["vid"] <= 3: print("vid differs: max abs diff of (A, S, Sn) =", d)
print("concurrent runs that differ from solo, of %d trials:" % trials, bad)
