#!/usr/bin/env python
"""bench.py -- QA-examples/sec (5-candidate fwd+bwd, full training step) of the MI355X-native STAGE at B=16 per GPU,
plus the K1 (StructuredAttention forward) roofline line and a CPU baseline of the same step (oracle port).

    python bench.py [--gpus N --steps K --warmup W]          # N=1 default
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # N>1: one rank per GPU over RCCL (driver does this)

Workload (BASELINE.json configs[1]): full STAGE, hsz=128, 300 frames x 20 regions, 50 subtitle
words/frame, 40 QA words, 5 candidates, B=16 per GPU (weak scaling; --scaling strong: global B=16 split over the GPUs),
--add_local --use_sup_att (run_main.sh:45 always passes it), dropout 0.1, fp32, synthetic ragged features seeded 2018.
One step = forward + loss (main.py:55-60: CE_sum * len(qids)/len(targets) + 0.1 * att_loss + 0.5 * temporal_loss, the ratio
taken over the GATHERED batch as the reference's DataParallel does) + backward + grad all-reduce (N>1) +
clip_grad_norm_(10) + Adam step, i.e. everything main.py:53-66 does per batch.  Inputs are resident in HBM.
Developer flags (not the headline line): --dense (all-ones masks), --heads 4, --no_sup_att, --h2d,
--storage bf16 (the bf16 storage mode of BASELINE.json configs[4] at these shapes; `dtype` then says "bf16").
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
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --bsz examples per GPU; strong: --bsz examples in total, sharded over the GPUs (SURVEY 8d)")
    ap.add_argument("--heads", type=int, default=0, help="self-attention heads in both encoders (BASELINE config 3: 4)")
    ap.add_argument("--no_sup_att", action="store_true", help="drop the supervised attention loss term (round-1 workload)")
    ap.add_argument("--att_imgs", type=int, default=4, help="annotated frames per question (synthetic att_labels)")
    ap.add_argument("--att_words", type=int, default=3, help="labelled object words per annotated frame")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--storage", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: the bf16 storage mode (BASELINE configs[4]); the headline line is the fp32 default")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--only_roofline", action="store_true")
    ap.add_argument("--cpu_seconds", type=float, default=20.0, help="CPU-baseline time budget")
    ap.add_argument("--h2d", action="store_true", help="developer mode: every step takes its batch from pinned host memory "
                    "through tvqaplus_amd.prefetch.BatchPrefetcher (PCIe-inclusive rate for DESIGN.md; never the headline value)")
    return ap.parse_args()


def train_step(model, batch, bucket, params, optimizer, n_examples, world=1):
    from tvqaplus_amd import parallel
    bucket.zero()
    (out, targets), att_loss, _, t_loss, _ = model(batch)
    # main.py:59 -- len(qids) / len(targets) of the gathered batch (N_new is data dependent with add_local)
    scale = (1.0 * n_examples / len(targets)) if world == 1 else parallel.global_loss_scale(n_examples, len(targets), out.device)
    loss = F.cross_entropy(out, targets, reduction="sum") * scale + 0.1 * att_loss + 0.5 * t_loss   # att_weight 0.1, ts_weight 0.5 (config.py)
    loss.backward()
    bucket.all_reduce()
    torch.nn.utils.clip_grad_norm_(params, 10.0)
    optimizer.step()
    return loss


def _profile_traffic(name):
    """WRITE_SIZE + 2 * FETCH_SIZE (KiB -> bytes) of the K1 forward from a committed rocprofv3 PMC summary, or None."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        vals = {}
        with open(path) as f:
            for line in f:
                if "str_attn_fwd" in line and ("FETCH_SIZE" in line or "WRITE_SIZE" in line):
                    key = "FETCH_SIZE" if "FETCH_SIZE" in line else "WRITE_SIZE"
                    vals[key] = float(line.rsplit("avg=", 1)[1])
        if len(vals) == 2:
            return round((vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"]) * 1024.0)
    except (OSError, ValueError, IndexError):
        pass
    return None


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
    # HBM bytes per launch from the PMC passes committed under profiles/ (tools/pmc_run.sh: WRITE_SIZE + 2 x FETCH_SIZE in
    # KiB, the x2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); only valid for the default shapes
    traffic = None
    if (N, NA, Li, Lqa, D) == (16, 5, 300, 40, 128) and not args.dense and Lr in (20, 50):
        traffic = _profile_traffic("r02_k1_fwd_pmc_%s.txt" % ("vid" if Lr == 20 else "sub"))
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
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    B = 1
    sup = not args.no_sup_att
    batch = make_batch(N=B, Li=args.frames, Lr=args.regions, Lw=args.sub_words, Lqa=args.qa_words, seed=2018,
                       ragged=not args.dense, att_imgs=args.att_imgs if sup else 0, att_words=args.att_words)
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
        loss = O.training_loss(out, n_examples=B)
        if sup:   # the supervised attention term on the oracle's own attention map (host index building is shared code)
            from tvqaplus_amd import att_host
            loss = loss + 0.1 * att_host.get_att_loss(opt, out["vid_raw_s"].squeeze(2) if out["vid_raw_s"].dim() == 6 else out["vid_raw_s"], batch)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        optim.step()

    step()  # warm-up (allocator, thread pool)
    t0 = time.time()
    n = 0
    while n < 1 or (time.time() - t0 < args.cpu_seconds and n < 50):
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(B * n / dt, 4), "unit": "QA-examples/s", "cores": cores, "host_cores": os.cpu_count(),
            "cpu_model": cpu_model, "kind": "port",
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
    sup = not args.no_sup_att
    opt = make_opt(hsz=args.hsz, add_local=True, dropout=0.1, use_sup_att=sup, input_encoder_n_heads=args.heads,
                   cls_encoder_n_heads=args.heads, storage_dtype=args.storage)
    import contextlib
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model = STAGE(opt)
    model = model.to(device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = parallel.FlatGradBucket(params)
    optimizer = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-7)
    if args.scaling == "strong":      # global batch of --bsz examples, example-major shards
        full = make_batch(N=args.bsz, Li=args.frames, Lr=args.regions, Lw=args.sub_words, Lqa=args.qa_words, seed=2018,
                          ragged=not args.dense, att_imgs=args.att_imgs if sup else 0, att_words=args.att_words)
        batch = parallel.shard_batch(full, rank, world).to(device)
        n_local = len(batch.qid)
        assert n_local > 0, "strong scaling needs at least one example per GPU (use parallel.CandidateLayout beyond that)"
    else:
        batch = make_batch(N=args.bsz, Li=args.frames, Lr=args.regions, Lw=args.sub_words, Lqa=args.qa_words,
                           seed=2018 + rank, ragged=not args.dense, att_imgs=args.att_imgs if sup else 0,
                           att_words=args.att_words).to(device)
        n_local = args.bsz
    n_global = args.bsz if args.scaling == "strong" else world * args.bsz

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.h2d:
        # PCIe-inclusive variant: the same host batch (pinned once) is re-sent every step, overlapped with the previous step
        from tvqaplus_amd.prefetch import BatchPrefetcher
        host = make_batch(N=args.bsz, Li=args.frames, Lr=args.regions, Lw=args.sub_words, Lqa=args.qa_words,
                          seed=2018 + rank, ragged=not args.dense, att_imgs=args.att_imgs if sup else 0,
                          att_words=args.att_words)
        for k, v in host.items():
            if torch.is_tensor(v):
                host[k] = v.pin_memory()
        feed = BatchPrefetcher((host for _ in range(args.warmup + args.steps)), device)
        nxt = lambda: next(feed)
    else:
        nxt = lambda: batch
    for _ in range(args.warmup):
        train_step(model, nxt(), bucket, params, optimizer, n_local, world)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, nxt(), bucket, params, optimizer, n_local, world)
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
            "value": round(n_global * args.steps / dt, 3),
            "unit": "QA-examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.storage == "fp32" else "bf16", "data": "synthetic" + (" (re-sent from pinned host memory every step)" if args.h2d else ""),
            "config": {"workload": "STAGE train step B=%d/GPU x5 cand x%d frames x%d regions x%d sub x%d QA words, hsz=%d, "
                                   "add_local%s%s, dropout 0.1, %s masks; %s"
                                   % (n_local, args.frames, args.regions, args.sub_words, args.qa_words, args.hsz,
                                      " + supervised attention loss" if sup else "",
                                      (" + %d-head self-attention" % args.heads) if args.heads else "",
                                      "all-ones" if args.dense else "ragged",
                                      "bf16 activations / bf16-rounded weights, fp32 statistics, softmax and accumulation"
                                      if args.storage == "bf16" else
                                      "fp32 via fp16-split MFMA GEMMs (error below an fp32 FMA chain)"),
                       "step": "fwd + loss (main.py:55-60) + bwd + grad all-reduce + clip_grad_norm_ + Adam",
                       "global_batch": n_global, "parallelism": "dp%d (example-sharded, flat 2.2MB grad "
                                                                  "all-reduce over RCCL)" % world,
                       "final_loss": round(loss_v, 4), "peak_hbm_gib": round(peak_gb, 2)},
        }
        if not args.no_roofline:   # rank 0's GPU, after the timed region (the other ranks wait at the final barrier)
            rec["roofline"] = k1_roofline(args, device)
            sub_args = argparse.Namespace(**vars(args))
            sub_args.regions = args.sub_words          # the same kernel family on the subtitle stream (50 words per frame)
            rec["roofline_sub"] = k1_roofline(sub_args, device)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args, opt)
        print(json.dumps(rec))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
