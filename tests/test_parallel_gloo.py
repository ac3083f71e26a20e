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


# ---------------------------------------------------------------------------------------------------------------
# add_local + the reference's own loss (main.py:57-60): CE_sum * len(qids) / len(targets) + 0.5 * temporal_loss, where
# len(targets) = N_new is data dependent -- the scale must be the GLOBAL N / N_new (parallel.global_loss_scale)
# ---------------------------------------------------------------------------------------------------------------
def _model_and_params(opt):
    import contextlib
    import io
    from tvqaplus_amd.stage import STAGE
    with contextlib.redirect_stdout(io.StringIO()):
        model = STAGE(opt)
    P = dict(model.state_dict())
    for k, p in model.named_parameters():
        P[k] = p
    return model, P


def _worker_add_local(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import torch.nn.functional as F
    from oracle import stage_oracle as O
    from tvqaplus_amd import parallel
    from tvqaplus_amd.synth import make_batch, make_opt
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(7)
    opt = make_opt(hsz=16, embedding_size=24, vfeat_size=20, dropout=0.0, add_local=True, t_iter=1)
    model, P = _model_and_params(opt)
    full = make_batch(N=5, Li=6, Lr=4, Lw=5, Lqa=6, wd_size=24, vfeat_size=20, seed=11)
    local = parallel.shard_batch(full, rank, world)
    bucket = parallel.FlatGradBucket(model.parameters())
    bucket.zero()
    out = O.stage_forward(P, opt, local, training=True)
    n_loc, n_new = len(local.qid), out["targets"].shape[0]
    scale = parallel.global_loss_scale(n_loc, n_new, device="cpu")
    scale_t = parallel.global_loss_scale(n_loc, n_new, device="cpu", as_tensor=True)     # the no-read-back form bench.py uses
    assert scale_t.dim() == 0 and abs(float(scale_t) - scale) < 1e-6 * scale
    loss = F.cross_entropy(out["logits"], out["targets"], reduction="sum") * scale + 0.5 * out["temporal_loss"]
    loss.backward()
    bucket.all_reduce()
    none_after = [k for (k, p) in model.named_parameters() if p.grad is None]
    if rank == 0:
        q.put((bucket.flat.clone(), scale, n_new, none_after))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_add_local_step_with_reference_loss():
    sys.path.insert(0, ROOT)
    import torch.nn.functional as F
    from oracle import stage_oracle as O
    from tvqaplus_amd.synth import make_batch, make_opt
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_add_local, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, scale, n_new0, none_after = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(7)
    opt = make_opt(hsz=16, embedding_size=24, vfeat_size=20, dropout=0.0, add_local=True, t_iter=1)
    model, P = _model_and_params(opt)
    full = make_batch(N=5, Li=6, Lr=4, Lw=5, Lqa=6, wd_size=24, vfeat_size=20, seed=11)
    out = O.stage_forward(P, opt, full, training=True)
    n_new = out["targets"].shape[0]
    assert n_new > 5, "the fixture must produce extra proposals, else the normalisation is not exercised"
    assert abs(scale - 5.0 / n_new) < 1e-12          # the GLOBAL len(qids) / len(targets)
    assert abs(scale - 3.0 / n_new0) > 1e-6          # ... which is not rank 0's local ratio
    loss = F.cross_entropy(out["logits"], out["targets"], reduction="sum") * (5.0 / n_new) + 0.5 * out["temporal_loss"]
    loss.backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    assert torch.allclose(flat, ref, rtol=1e-4, atol=1e-5), float((flat - ref).abs().max())
    # the t_iter > 0 refinement layers never receive a gradient (model/stage.py:516): they stay at grad = None after the
    # all-reduce exactly as in a single process (zero gradients would let Adam's weight decay move them)
    ref_none = sorted(k for k, p in model.named_parameters() if p.grad is None)
    assert ref_none and sorted(none_after) == ref_none


# ---------------------------------------------------------------------------------------------------------------
# candidate x batch layout: 5 ranks, 1 example -> one candidate per rank (E = 1, C = 5); 5 ranks, 2 examples -> E = 2, C = 2
# ---------------------------------------------------------------------------------------------------------------
def _worker_candidates(rank, world, port, q, n_examples):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import stage_oracle as O
    from tvqaplus_amd import parallel
    from tvqaplus_amd.synth import make_batch, make_opt
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(13)
    opt = make_opt(hsz=16, embedding_size=24, vfeat_size=20, dropout=0.0, add_local=True)
    model, P = _model_and_params(opt)
    full = make_batch(N=n_examples, Li=6, Lr=4, Lw=5, Lqa=6, wd_size=24, vfeat_size=20, seed=17)
    layout = parallel.CandidateLayout(n_examples)
    bucket = parallel.FlatGradBucket(model.parameters())
    bucket.zero()
    logits_local = torch.zeros(0, 0)
    if layout.active:
        local = layout.shard(full)
        out = O.stage_forward(P, opt, local, training=True)
        logits_local = out["logits"]
        n_loc, n_new = len(local.qid), out["targets"].shape[0]
    else:
        n_loc, n_new = 0, 0
    # every example is counted once: by the first rank of its group
    scale = parallel.global_loss_scale(n_loc, n_new, device="cpu", count_this_rank=layout.active and layout.part == 0)
    if layout.active:
        loss = layout.cross_entropy_sum(out["logits"], out["targets"]) * scale + 0.5 * out["temporal_loss"]
        loss.backward()
    bucket.all_reduce()
    # eval-mode logits of the full batch, gathered on every rank
    if layout.active:
        ev = O.stage_forward(P, opt, layout.shard(full), training=False)["logits"]
    else:
        ev = torch.zeros(0, 0)
    gathered = layout.gather_logits_world(ev)
    if rank == 0:
        q.put((bucket.flat.clone(), gathered, (layout.E, layout.C)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_examples,expect", [(1, (1, 5)), (2, (2, 2))])
def test_candidate_sharded_step_equals_single_process(n_examples, expect):
    sys.path.insert(0, ROOT)
    import torch.nn.functional as F
    from oracle import stage_oracle as O
    from tvqaplus_amd.synth import make_batch, make_opt
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + 7 * n_examples) % 2000
    procs = [ctx.Process(target=_worker_candidates, args=(r, 5, port, q, n_examples)) for r in range(5)]
    for p in procs:
        p.start()
    flat, gathered, ec = q.get(timeout=400)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ec == expect
    torch.manual_seed(13)
    opt = make_opt(hsz=16, embedding_size=24, vfeat_size=20, dropout=0.0, add_local=True)
    model, P = _model_and_params(opt)
    full = make_batch(N=n_examples, Li=6, Lr=4, Lw=5, Lqa=6, wd_size=24, vfeat_size=20, seed=17)
    out = O.stage_forward(P, opt, full, training=True)
    n_new = out["targets"].shape[0]
    loss = F.cross_entropy(out["logits"], out["targets"], reduction="sum") * (float(n_examples) / n_new) + 0.5 * out["temporal_loss"]
    loss.backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    assert torch.allclose(flat, ref, rtol=2e-4, atol=2e-5), float((flat - ref).abs().max())
    ev = O.stage_forward(P, opt, full, training=False)["logits"]
    assert torch.allclose(gathered, ev.detach(), rtol=1e-5, atol=1e-6)


def test_candidate_layout_partitions():
    from tvqaplus_amd.parallel import CandidateLayout
    for n, w in ((1, 5), (1, 8), (2, 5), (2, 8), (16, 8), (3, 7), (16, 1)):
        seen = {}
        for r in range(w):
            lay = CandidateLayout(n, rank=r, world=w)
            if not lay.active:
                continue
            for e in range(*lay.example_range):
                for k in range(*lay.cand_range):
                    assert (e, k) not in seen
                    seen[(e, k)] = r
        assert len(seen) == 5 * n, (n, w)        # every (example, candidate) pair exactly once


def _absent_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tvqaplus_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = [torch.nn.Parameter(torch.zeros(3)) for _ in range(3)]
    bucket = parallel.FlatGradBucket(ps)
    out = []
    for step in range(3):
        bucket.zero()
        ps[0].grad = torch.ones(3)
        if rank == 0:
            ps[1].grad = torch.ones(3)
        if step >= 1 and rank == 1:
            ps[2].grad = torch.ones(3)          # a branch that only rank 1 switches on after the first step
        try:
            bucket.all_reduce()
            out.append([p.grad is None for p in ps])
        except RuntimeError as e:
            out.append("raised: " + str(e)[:60])
            break
    q.put((rank, out))
    dist.destroy_process_group()


def test_changed_gradient_set_raises_on_every_rank():
    """a rank whose set of gradient-carrying parameters changes after the agreement must not raise alone (its peers would hang
    in their next collective): the change travels with the gradient all-reduce, the step applies the old agreement on every rank, and
    the NEXT step raises everywhere."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_absent_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert res[r][0] == [False, False, True]          # step 0: parameter 2 has no gradient anywhere
        assert res[r][1] == [False, False, True]          # step 1: rank 1 has one now -- the old agreement still holds, on both ranks
        assert isinstance(res[r][2], str) and res[r][2].startswith("raised"), res[r]


def test_flat_clip_equals_torch_clip_grad_norm():
    """FlatGradBucket.clip_grad_norm_ (one norm + one multiply on the packed buffer) against torch.nn.utils.clip_grad_norm_ on the same
    gradients: same norm, same clipped gradients -- with a parameter that received no gradient, above and below the threshold."""
    from tvqaplus_amd import parallel
    torch.manual_seed(3)
    for scale in (0.01, 50.0):
        ps = [torch.nn.Parameter(torch.randn(s)) for s in ((7, 5), (11,), (3, 4, 2), (6,))]
        ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        grads = [scale * torch.randn_like(p) for p in ps]
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i != 3:                                   # the last parameter stays without a gradient
                p.grad, r.grad = grads[i].clone(), grads[i].clone()
        bucket = parallel.FlatGradBucket(ps)
        bucket.all_reduce()
        n_flat = bucket.clip_grad_norm_(10.0)
        n_ref = torch.nn.utils.clip_grad_norm_([r for r in ref if r.grad is not None], 10.0)
        assert abs(float(n_flat) - float(n_ref)) <= 1e-5 * float(n_ref)
        for p, r in zip(ps[:3], ref[:3]):
            assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-7)
        assert ps[3].grad is None
