"""Does a per-kernel op return different bits when another stream keeps the GPU busy?  (Hunt for the level-4 branch-stream difference on
the bf16 per-kernel path: tools/experiments/stream_race_probe.py.)  Each op: reference result alone, then TRIALS runs on the main stream
while a side stream runs (a) large torch matmuls, (b) the long-row attention at the video shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import ops
dev = "cuda"
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
N, NA, Li, Lqa, Lw, Lr, D = 4, 5, int(os.environ.get("LI", 100)), 40, 512, 20, 256
def rnd(*s, dt=bf):
    return torch.randn(*s, device=dev, generator=g).to(dt)
C = rnd(N, NA, Lqa, D); Qs = rnd(N, Li, Lw, D); Qv = rnd(N, Li, Lr, D)
cm = torch.ones(N, NA, Lqa, device=dev); cm[:, :, 30:] = 0
lens = torch.randint(5, Lw, (N, Li), device=dev, generator=g)
qms = (torch.arange(Lw, device=dev).view(1, 1, Lw) < lens.unsqueeze(-1)).float()
qmv = torch.ones(N, Li, Lr, device=dev)
W = rnd(D, D, dt=torch.float32) * 0.05; bias = rnd(D, dt=torch.float32)
W3 = rnd(D, 3 * D, dt=torch.float32) * 0.05
gam, bet = torch.ones(3 * D, device=dev), torch.zeros(3 * D, device=dev)
A0 = rnd(N * NA * Li * Lqa // 4, D)
a_rows = rnd(N * NA * Lqa, D); u_rows = rnd(N * NA * Li * Lqa, D)
side = torch.cuda.Stream()
M1, M2 = rnd(4096, 4096), rnd(4096, 4096)

def noise(kind):
    with torch.cuda.stream(side):
        for _ in range(6):
            if kind == "mm":
                torch.mm(M1, M2)
            else:
                ops.structured_attention(C, Qv, cm, qmv, 1.0)

OPS = {
    "attn_long_sub": lambda: ops.structured_attention(C, Qs, cm, qms, 1.0),
    "attn_long_vid": lambda: ops.structured_attention(C, Qv, cm, qmv, 1.0),
    "linear_bf16": lambda: (ops.linear(A0, W, bias, relu=True),),
    "cat3_ln": lambda: (ops.cat3_layernorm(a_rows, u_rows, gam, bet, rep=Li, inner=Lqa, p=0.1, seed=11),),
    "cat3_ln+linear": lambda: (ops.linear(ops.cat3_layernorm(a_rows, u_rows, gam, bet, rep=Li, inner=Lqa, p=0.1, seed=11), W3, bias, relu=True),),
}
with torch.no_grad():
    for name, fn in OPS.items():
        ref = [t.float().clone() for t in fn()]
        torch.cuda.synchronize()
        for kind in ("mm", "attn"):
            bad = 0
            for trial in range(5):
                noise(kind)
                cur = fn()
                torch.cuda.synchronize()
                if any(not torch.equal(c.float(), r) for c, r in zip(cur, ref)):
                    bad += 1
            print(f"{name:16s} noise={kind:5s} differing trials: {bad}/5", flush=True)
