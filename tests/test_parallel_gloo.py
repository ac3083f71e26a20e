"""world_size-2 gloo tests (CPU) of the example-sharded data-parallel path (tvqaplus_amd/parallel.py).

The collectives and the sharding are what is under test; the per-rank compute is the CPU oracle (the HIP product needs
a GPU).  Claim checked: sharding the batch over ranks + ONE flat sum all-reduce of the gradients + an all-gather of
the logits reproduces the single-process result of the full batch (losses are CE(sum), main.py:57-60, 208)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import stage_oracle as O
    from tvqaplus_amd import parallel
    from tvqaplus_amd.synth import make_batch, make_opt
    r, l, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(5)
    opt = make_opt(hsz=16, embedding_size=24, vfeat_size=20, dropout=0.0)
    import contextlib
    import io
    from tvqaplus_amd.stage import STAGE
    with contextlib.redirect_stdout(io.StringIO()):
        model = STAGE(opt)  # parameter container only (same seed on every rank -> identical weights)
    params = [p for p in model.parameters()]
    names = [k for k, _ in model.named_parameters()]
    full = make_batch(N=5, Li=3, Lr=4, Lw=5, Lqa=6, wd_size=24, vfeat_size=20, seed=9)  # 5 examples: uneven shards
    local = parallel.shard_batch(full, rank, world)
    lo, hi = parallel.shard_range(5, rank, world)
    assert len(local.qid) == hi - lo and local.qas_bert.shape[0] == hi - lo

    bucket = parallel.FlatGradBucket(params)
    P = dict(model.state_dict())
    for k, p in zip(names, params):
        P[k] = p
    bucket.zero()
    out = O.stage_forward(P, opt, local, training=True)
    # un-normalised sum losses so that the sum over ranks equals the full-batch loss
    loss = torch.nn.functional.cross_entropy(out["logits"], out["targets"], reduction="sum") + 0.5 * out["temporal_loss"]
    loss.backward()
    bucket.all_reduce()
    counts = [parallel.shard_range(5, rr, world)[1] - parallel.shard_range(5, rr, world)[0] for rr in range(world)]
    logits = parallel.all_gather_outputs(out["logits"].detach(), counts)
    if rank == 0:
        q.put((bucket.flat.clone(), logits))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_step_equals_full_batch():
    sys.path.insert(0, ROOT)
    from oracle import stage_oracle as O
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_batch, make_opt
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, logits = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    torch.manual_seed(5)
    opt = make_opt(hsz=16, embedding_size=24, vfeat_size=20, dropout=0.0)
    model = STAGE(opt)
    P = dict(model.state_dict())
    for k, p in model.named_parameters():
        P[k] = p
    full = make_batch(N=5, Li=3, Lr=4, Lw=5, Lqa=6, wd_size=24, vfeat_size=20, seed=9)
    out = O.stage_forward(P, opt, full, training=True)
    loss = torch.nn.functional.cross_entropy(out["logits"], out["targets"], reduction="sum") + 0.5 * out["temporal_loss"]
    loss.backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    assert torch.allclose(flat, ref, rtol=1e-4, atol=1e-5), float((flat - ref).abs().max())
    assert torch.allclose(logits, out["logits"].detach(), rtol=1e-5, atol=1e-6)


def test_shard_range_partition():
    from tvqaplus_amd.parallel import shard_range
    for n in (1, 5, 16, 17):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
