"""-m gpu: the multi-rank step with the HIP model on DEVICE tensors, without a multi-GPU node.

Several processes share cuda:0, each runs the HIP ``STAGE`` on its shard, and ``tvqaplus_amd.parallel``'s collectives
(``FlatGradBucket.all_reduce``, ``all_gather_outputs``, ``global_loss_scale``, ``CandidateLayout``) run over a gloo group with
host staging (``parallel._host_staged``) -- the same call sites RCCL serves on an 8-GPU node, on the tensors the HIP ops
produce.  Claim: the sharded HIP steps reproduce the single-process HIP step of the full batch (example-major with add_local
and the reference's loss normalisation, and the 1 x 5 candidate layout)."""
import contextlib
import io
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPT = dict(hsz=32, embedding_size=48, vfeat_size=40, dropout=0.0, add_local=True)
SHAPE = dict(Li=6, Lr=6, Lw=7, Lqa=8, wd_size=48, vfeat_size=40)


def _model(seed):
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_opt
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        model = STAGE(make_opt(**OPT))
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return model.to("cuda:0").train()


def _worker(rank, world, port, q, mode, n_examples):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    import torch.nn.functional as F
    from tvqaplus_amd import parallel
    from tvqaplus_amd.synth import make_batch
    torch.cuda.set_device(0)
    parallel.init_from_env(backend="gloo")
    model = _model(21)                     # same seed on every rank -> identical weights
    full = make_batch(N=n_examples, seed=23, **SHAPE)
    bucket = parallel.FlatGradBucket(model.parameters())
    bucket.zero()
    if mode == "examples":
        local = parallel.shard_batch(full, rank, world).to("cuda:0")
        (out, targets), _, _, t_loss, _ = model(local)
        scale = parallel.global_loss_scale(len(local.qid), len(targets), device="cuda:0")
        loss = F.cross_entropy(out, targets, reduction="sum") * scale + 0.5 * t_loss
        loss.backward()
        bucket.all_reduce()
        counts = [parallel.shard_range(n_examples, r, world)[1] - parallel.shard_range(n_examples, r, world)[0]
                  for r in range(world)]
        model.eval()
        with torch.no_grad():
            ev = model(local)[0]
        gathered = parallel.all_gather_outputs(ev, counts)
    else:
        layout = parallel.CandidateLayout(n_examples)
        local = layout.shard(full).to("cuda:0")
        (out, targets), _, _, t_loss, _ = model(local)
        scale = parallel.global_loss_scale(len(local.qid), len(targets), device="cuda:0",
                                           count_this_rank=layout.part == 0)
        loss = layout.cross_entropy_sum(out, targets) * scale + 0.5 * t_loss
        loss.backward()
        bucket.all_reduce()
        model.eval()
        with torch.no_grad():
            ev = model(layout.shard(full).to("cuda:0"))[0]
        gathered = layout.gather_logits_world(ev)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((bucket.flat.cpu(), gathered.cpu(), float(scale)))
    dist.barrier()
    dist.destroy_process_group()


def _single(n_examples):
    import torch.nn.functional as F
    from tvqaplus_amd.synth import make_batch
    model = _model(21)
    full = make_batch(N=n_examples, seed=23, **SHAPE).to("cuda:0")
    (out, targets), _, _, t_loss, _ = model(full)
    n_new = len(targets)
    (F.cross_entropy(out, targets, reduction="sum") * (float(n_examples) / n_new) + 0.5 * t_loss).backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    model.eval()
    with torch.no_grad():
        ev = model(full)[0]
    return flat.cpu(), ev.cpu(), float(n_examples) / n_new


def _run(world, mode, n_examples, port_base):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + os.getpid() % 1500
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, n_examples)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=500)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _close(a, b, tol):
    return float(((a - b).abs() / (1.0 + b.abs())).max()) < tol


@pytest.mark.timeout(900)
def test_example_sharded_hip_step_equals_single_process(hip_device):
    flat, gathered, scale = _run(2, "examples", 5, 36500)
    ref_flat, ref_ev, ref_scale = _single(5)
    assert abs(scale - ref_scale) < 1e-12                    # the GLOBAL len(qids) / len(targets)
    assert _close(gathered, ref_ev, 1e-4)
    # weight-gradient sums run over different row sets per rank: fp32 reassociation only
    assert _close(flat, ref_flat, 2e-3), float(((flat - ref_flat).abs() / (1 + ref_flat.abs())).max())


@pytest.mark.timeout(900)
def test_candidate_sharded_hip_step_equals_single_process(hip_device):
    flat, gathered, scale = _run(5, "candidates", 1, 38500)   # 1 example on 5 ranks: one candidate per rank (E=1, C=5)
    ref_flat, ref_ev, ref_scale = _single(1)
    assert abs(scale - ref_scale) < 1e-12
    assert _close(gathered, ref_ev, 1e-4)
    assert _close(flat, ref_flat, 2e-3), float(((flat - ref_flat).abs() / (1 + ref_flat.abs())).max())


@pytest.mark.timeout(900)
def test_bench_strong_scaling_single_rank(hip_device):
    """``bench.py --gpus 1 --scaling strong``: the N = 1 degenerate path of the sharded step (FlatGradBucket on device tensors,
    ``shard_batch`` of the global batch) end to end, at a small shape."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--scaling", "strong",
           "--bsz", "3", "--frames", "12", "--regions", "8", "--sub_words", "10", "--qa_words", "9", "--hsz", "64",
           "--no_cpu_baseline", "--no_roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 1 and rec["value"] > 0


@pytest.mark.timeout(900)
def test_bench_two_ranks_sharing_one_gpu(hip_device):
    """``bench.py`` launched as the driver launches it for N = 2 (``python -m torch.distributed.run --nproc-per-node 2``), with both
    ranks on cuda:0 and the collectives over gloo (STAGE_BENCH_SHARED_GPU): barriers, the max over ranks, the all-rank device-time
    pass after the timed region (rank 0 alone would hang in the gradient all-reduce) and rank 0's one JSON line."""
    import json
    port = 29000 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--bsz", "2", "--frames", "12", "--regions", "8", "--sub_words", "10", "--qa_words", "9", "--hsz", "64",
           "--no_cpu_baseline", "--no_roofline"]
    env = dict(os.environ, STAGE_BENCH_SHARED_GPU="1")
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                       # rank 0 prints the one line
    rec = json.loads(lines[0])
    # (default --scaling strong: the --bsz examples are the GLOBAL batch, one example per rank here)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["global_batch"] == 2 and rec["device_ms_per_step"] > 0


def _nccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import torch.nn.functional as F
    from tvqaplus_amd import parallel
    from tvqaplus_amd.synth import make_batch
    try:
        parallel.ALWAYS_COLLECTIVE = True
        rank, local, world = parallel.init_from_env(backend="nccl")
        assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
        model = _model(31)
        batch = make_batch(N=3, seed=33, **SHAPE).to("cuda:0")
        bucket = parallel.FlatGradBucket(model.parameters())
        bucket.zero()
        (out, targets), _, _, t_loss, _ = model(batch)
        scale = parallel.global_loss_scale(len(batch.qid), len(targets), device="cuda:0", as_tensor=True)   # RCCL all-reduce (2 doubles)
        loss = F.cross_entropy(out, targets, reduction="sum") * scale + 0.5 * t_loss
        loss.backward()
        before = [p.grad.detach().clone() if p.grad is not None else None for p in bucket.params]
        bucket.all_reduce()                                                                       # RCCL all-reduce (flat, 1.3 MB)
        gathered = parallel.all_gather_outputs(out.detach())                                      # RCCL all-gather
        torch.cuda.synchronize()
        ok = torch.equal(gathered, out.detach()) and abs(float(scale) - len(batch.qid) / len(targets)) < 1e-6
        for b, p in zip(before, bucket.params):
            ok = ok and ((b is None and p.grad is None) or torch.equal(b, p.grad))
        q.put(("ok" if ok else "mismatch", dist.get_backend()))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put(("error: %r" % (e,), ""))


def test_rccl_collectives_run_on_device_tensors():
    """The RCCL calls of tvqaplus_amd.parallel EXECUTE (backend "nccl" = RCCL on ROCm): a one-rank group on the one GPU, the flat gradient
    all-reduce, the output all-gather and the loss-scale all-reduce issued on device tensors produced by the HIP model.  With one rank
    they are identities -- what this pins is that the library loads, the communicator initialises and the calls complete on this
    stack; the multi-rank arithmetic is covered over gloo above."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    p = ctx.Process(target=_nccl_worker, args=(port, q))
    p.start()
    try:
        status, backend = q.get(timeout=300)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    assert status == "ok" and backend == "nccl", status
