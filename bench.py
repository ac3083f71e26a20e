#!/usr/bin/env python
"""bench.py -- QA-examples/sec (5-candidate fwd+bwd, full training step) of the MI355X-native STAGE at B=16 per GPU,
plus the K1 (StructuredAttention forward) roofline line and a CPU baseline of the same step (oracle port).

    python bench.py [--gpus N --steps K --warmup W]          # N=1 default
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # N>1: one rank per GPU over RCCL (driver does this)

Workload (BASELINE.json configs[1]): full STAGE, hsz=128, 300 frames x 20 regions, 50 subtitle
words/frame, 40 QA words, 5 candidates, B=16 per GPU (weak scaling), --add_local, dropout 0.1, fp32, synthetic ragged
features seeded 2018.  One step = forward + loss (main.py:55-60 w/o att term) + backward + grad all-reduce (N>1) +
clip_grad_norm_(10) + Adam step, i.e. everything main.py:53-66 does per batch.  Inputs are resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bsz", type=int, default=16, help="examples per GPU")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--regions", type=int, default=20)
    ap.add_argument("--sub_words", type=int, default=50)
    ap.add_argument("--qa_words", type=int, default=40)
    ap.add_argument("--hsz", type=int, default=128)
    ap.add_argument("--dense", action="store_true", help="all-ones masks instead of ragged lengths")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--only_roofline", action="store_true")
    ap.add_argument("--cpu_seconds", type=float, default=20.0, help="CPU-baseline time budget")
    ap.add_argument("--h2d", action="store_true", help="developer mode: every step takes its batch from pinned host memory "
                    "through tvqaplus_amd.prefetch.BatchPrefetcher (PCIe-inclusive rate for DESIGN.md; never the headline value)")
    return ap.parse_args()


def train_step(model, batch, bucket, params, optimizer, n_examples):
    bucket.zero()
    (out, targets), att_loss, _, t_loss, _ = model(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") * (1.0 * n_examples / len(targets)) + 0.5 * t_loss
    loss.backward()
    bucket.all_reduce()
    torch.nn.utils.clip_grad_norm_(params, 10.0)
    optimizer.step()
    return loss


def k1_roofline(args, device):
    """Isolated StructuredAttention forward kernel (video-stream shape) timed with events on the launch stream."""
    from tvqaplus_amd import _lib
    from tvqaplus_amd.synth import make_batch
    lib = _lib.load()
    N, NA, Li, Lqa, Lr, D = args.bsz, 5, args.frames, args.qa_words, args.regions, args.hsz
    g = torch.Generator().manual_seed(2018)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=2018, ragged=not args.dense)
    Cn = F.normalize(torch.randn(N, NA, Lqa, D, generator=g), dim=-1).to(device)
    Q = torch.randn(N, Li, Lr, D, generator=g).to(device)
    cm, qm = b.qas_mask.to(device).contiguous(), b.vid_mask.to(device).contiguous()
    A = torch.empty(N, NA, Li, Lqa, D, device=device)
    S = torch.empty(N, NA, Li, Lqa, Lr, device=device)
    Sn = torch.empty_like(S)
    stream = torch.cuda.current_stream()

    def launch():
        _lib.check(lib.stage_str_attn_fwd(Cn.data_ptr(), Q.data_ptr(), cm.data_ptr(), qm.data_ptr(), A.data_ptr(),
                                          S.data_ptr(), Sn.data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0, 0.0, 0,
                                          stream.cuda_stream), "stage_str_attn_fwd")
    for _ in range(5):
        launch()
    reps = 30
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(stream)
        launch()
        e.record(stream)
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    avg_ms = sum(ms) / len(ms)
    U = N * NA * Li * Lqa
    # algorithmic bytes: inputs once + A + S + S_ written once, fp32
    alg = 4 * (N * NA * Lqa * D + N * Li * Lr * D + N * NA * Lqa + N * Li * Lr + U * D + 2 * U * Lr)
    achieved = alg / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed under profiles/ (tools/pmc_k1.sh: WRITE_SIZE + 2 x FETCH_SIZE,
    # the x2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); only valid for the default shape
    default_shape = (N, NA, Li, Lqa, Lr, D) == (16, 5, 300, 40, 20, 128) and not args.dense
    traffic = 646.9e6 + 2 * 36.05e6 if default_shape else None
    return {"bound": "hbm", "kernel": "str_attn_fwd_reg_kernel" if Lr <= 32 else "str_attn_fwd_d128_kernel", "achieved": round(achieved, 1), "peak": 8000.0,
            "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic, "algorithmic_bytes": alg,
            "avg_us": round(avg_ms * 1e3, 1), "min_us": round(ms[0] * 1e3, 1),
            "shape": {"N": N, "NA": NA, "Li": Li, "Lqa": Lqa, "Lr": Lr, "D": D}}


def cpu_baseline(args, opt):
    """The oracle (CPU port of the reference path, plain torch fp32) on the host cores: same step, bounded sample."""
    from oracle import stage_oracle as O
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_batch
    # torch's CPU kernels stop scaling (and thrash) far below a 256-thread host: 32 threads is what is used and reported
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    B = 1
    batch = make_batch(N=B, Li=args.frames, Lr=args.regions, Lw=args.sub_words, Lqa=args.qa_words, seed=2018,
                       ragged=not args.dense)
    torch.manual_seed(2018)
    import contextlib
    with contextlib.redirect_stdout(open(os.devnull, "w")):   # STAGE.__init__ prints which branches are active
        ref_model = STAGE(opt)
    P = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pe"))
         for k, v in ref_model.state_dict().items()}
    params = [v for v in P.values() if v.requires_grad]
    optim = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7)

    def step():
        optim.zero_grad(set_to_none=True)
        out = O.stage_forward(P, opt, batch, training=True)
        O.training_loss(out, n_examples=B).backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        optim.step()

    step()  # warm-up (allocator, thread pool)
    t0 = time.time()
    n = 0
    while n < 1 or (time.time() - t0 < args.cpu_seconds and n < 50):
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(B * n / dt, 4), "unit": "QA-examples/s", "cores": cores, "kind": "port",
            "sample": "%d full training steps of B=%d x %d frames (same per-example shapes, dropout 0.1) in %.1f s"
                      % (n, B, args.frames, dt)}


def main():
    args = parse()
    from tvqaplus_amd import parallel
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_batch, make_opt
    rank, local, world = parallel.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if args.only_roofline:  # developer shortcut: just the K1 kernel line (video and subtitle stream shapes)
        print(json.dumps({"vid": k1_roofline(args, device)}))
        args.regions = args.sub_words
        print(json.dumps({"sub": k1_roofline(args, device)}))
        return
    torch.manual_seed(2018)
    opt = make_opt(hsz=args.hsz, add_local=True, dropout=0.1)
    import contextlib
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model = STAGE(opt)
    model = model.to(device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = parallel.FlatGradBucket(params)
    optimizer = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7)
    batch = make_batch(N=args.bsz, Li=args.frames, Lr=args.regions, Lw=args.sub_words, Lqa=args.qa_words,
                       seed=2018 + rank, ragged=not args.dense).to(device)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.h2d:
        # PCIe-inclusive variant: the same host batch (pinned once) is re-sent every step, overlapped with the previous step
        from tvqaplus_amd.prefetch import BatchPrefetcher
        host = make_batch(N=args.bsz, Li=args.frames, Lr=args.regions, Lw=args.sub_words, Lqa=args.qa_words,
                          seed=2018 + rank, ragged=not args.dense)
        for k, v in host.items():
            if torch.is_tensor(v):
                host[k] = v.pin_memory()
        feed = BatchPrefetcher((host for _ in range(args.warmup + args.steps)), device)
        nxt = lambda: next(feed)
    else:
        nxt = lambda: batch
    for _ in range(args.warmup):
        train_step(model, nxt(), bucket, params, optimizer, args.bsz)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, nxt(), bucket, params, optimizer, args.bsz)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_v = float(loss.detach())
    peak_gb = torch.cuda.max_memory_allocated() / 2 ** 30

    if rank == 0:
        rec = {
            "metric": "QA-examples/sec (5-candidate fwd+bwd) at B=16",
            "value": round(world * args.bsz * args.steps / dt, 3),
            "unit": "QA-examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" + (" (re-sent from pinned host memory every step)" if args.h2d else ""),
            "config": {"workload": "STAGE full training step (fwd + loss + bwd + clip + Adam), hsz=128, add_local, "
                                   "dropout 0.1; per GPU B=%d x 5 candidates x %d frames x %d regions x %d sub words x "
                                   "%d QA words; %s masks" % (args.bsz, args.frames, args.regions, args.sub_words,
                                                              args.qa_words, "all-ones" if args.dense else "ragged"),
                       "global_batch": world * args.bsz, "parallelism": "dp%d (example-sharded, flat 2.2MB grad "
                                                                        "all-reduce over RCCL)" % world,
                       "final_loss": round(loss_v, 4), "peak_hbm_gib": round(peak_gb, 2)},
        }
        if not args.no_roofline:   # rank 0's GPU, after the timed region (the other ranks wait at the final barrier)
            rec["roofline"] = k1_roofline(args, device)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args, opt)
        print(json.dumps(rec))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
