"""BASELINE.json configs[4] (stress): hsz = 256, 512 subtitle words per frame; K1 in bf16 and fp32 storage (long-row kernels,
csrc/str_attn_long.hip) and one whole-model fp32 training step.  Prints one JSON line per measurement.
    python tools/bench_stress.py [N] [Li]"""
import contextlib, json, os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import ops
from tvqaplus_amd.stage import STAGE
from tvqaplus_amd.synth import make_batch, make_opt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Li = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NA, Lqa, Lr, D = 5, 40, 512, 256
dev = "cuda"
g = torch.Generator().manual_seed(2018)
b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=2018)
cm, qm = b.qas_mask.to(dev), b.vid_mask.to(dev)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
U = N * NA * Li * Lqa
only = os.environ.get("STRESS_ONLY")       # "bf16" / "fp32": just that whole-model step (kernel traces)
for dt in (() if only else (torch.bfloat16, torch.float32)):
    C = torch.randn(N, NA, Lqa, D, generator=g).to(dev).to(dt).requires_grad_()
    Q = torch.randn(N, Li, Lr, D, generator=g).to(dev).to(dt).requires_grad_()
    gA = torch.randn(N, NA, Li, Lqa, D, generator=g).to(dev).to(dt)
    fwd = lambda: ops.structured_attention(C, Q, cm, qm, 10.0)
    def fb():
        C.grad = Q.grad = None
        A, S, Sn = ops.structured_attention(C, Q, cm, qm, 10.0)
        A.backward(gA)
    tf, tfb = timed(fwd), timed(fb)
    es = 2 if dt == torch.bfloat16 else 4
    alg = es * (N * NA * Lqa * D + N * Li * Lr * D + U * D) + 4 * (2 * U * Lr)      # inputs once, A, S and S_ written once
    flops = 2 * 2 * U * Lr * D
    print(json.dumps({"kernel": "K1 long rows, %s storage" % str(dt).split(".")[1], "shape": dict(N=N, NA=NA, Li=Li, Lqa=Lqa, Lr=Lr, D=D),
                      "fwd_us": round(tf * 1e6, 1), "fwd_bwd_us": round(tfb * 1e6, 1), "fwd_algorithmic_MB": round(alg / 1e6, 1),
                      "fwd_GBps": round(alg / tf / 1e9, 1), "fwd_TFLOPs": round(flops / tf / 1e12, 2)}))
batch = make_batch(N=N, Li=Li, Lr=20, Lw=512, Lqa=40, seed=3).to(dev)
for storage in (() if os.environ.get("STRESS_K1_ONLY") else ((only,) if only else ("bf16", "fp32"))):
    torch.manual_seed(0)
    opt = make_opt(hsz=256, add_local=True, dropout=0.1, storage_dtype=storage)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model = STAGE(opt).to(dev).train()
    params = [p for p in model.parameters()]
    optim = torch.optim.Adam(params, lr=1e-3)
    def step():
        optim.zero_grad(set_to_none=True)
        (out, targets), _, _, t_loss, _ = model(batch)
        (F.cross_entropy(out, targets, reduction="sum") * (N / len(targets)) + 0.5 * t_loss).backward()
        optim.step()
    torch.cuda.reset_peak_memory_stats()
    t = timed(step, 5)
    print(json.dumps({"workload": "STAGE %s-storage train step, hsz=256, %d x %d frames x 512 subtitle words (+20 regions), 40 QA words" % (storage, N, Li),
                      "ms_per_step": round(t * 1e3, 2), "qa_examples_per_s": round(N / t, 2),
                      "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
    del model, optim, params
