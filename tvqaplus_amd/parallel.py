"""One process per GPU over RCCL/xGMI.  The reference's only multi-device construct is
``nn.DataParallel`` (main.py:204-206) with ``bsz *= n_gpu`` (config.py:186-188): examples are independent, so the batch
is sharded example-major and exactly two collectives exist:

* ``all_gather_outputs``  -- logits (N_loc, 5) [+ t_scores] of every rank -> all ranks (forward, 320 B .. 192 KB)
* ``all_reduce_grads``    -- ONE flat bucket with every parameter gradient (552 947 fp32 = 2.2 MB at D=128), summed.
                             The losses are sums over examples (CE(sum), main.py:57-60, 208), so the sum over ranks is
                             the single-GPU gradient of the concatenated batch PROVIDED the classification term is
                             scaled by the GLOBAL len(qids) / len(targets) (main.py:59 normalises on the gathered
                             DataParallel outputs; with add_local the number of proposals N_new is data dependent, so
                             a per-rank N_r / N_new_r is a different loss) -- ``global_loss_scale`` below.

When there are more GPUs than examples (inference at small B; BASELINE config 4 names a candidate x batch split) the
five answer candidates of an example are split over the ranks of a ``CandidateLayout`` group as well: every operator between
the QA embedding and ``answer_scores[n, a]`` is independent per (n, a); the context encoders are
replicated inside the group, the classification loss needs the five logits of an example (one all-gather inside the
group, autograd-aware), and add_local training derives its proposal spans from the ground-truth candidate's temporal
scores (model/stage.py:408-418) -- broadcast inside the group by ``CandidateLayout.gt_scores``.

Both are latency-bound at these sizes (2.2 MB over a 153 GB/s xGMI link is ~15 us/hop), so no bucketing / overlap
machinery is warranted; the flat bucket exists to pay the collective latency once instead of ~80 times.
Backend: "nccl" (= RCCL on ROCm) for device tensors, "gloo" for the CPU tests.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run.  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or ALWAYS_COLLECTIVE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


# True: the collectives below are ISSUED even in a one-rank group (where they are identities).  tests/test_parallel_hip.py uses it to
# execute the RCCL calls of this module -- all-reduce of the flat gradient buffer, all-gather of the outputs, the loss-scale
# all-reduce -- on device tensors on a box with a single GPU; production leaves it off (a one-rank job has nothing to exchange).
ALWAYS_COLLECTIVE = False


def _single() -> bool:
    """No exchange needed: no process group, or one rank (unless ALWAYS_COLLECTIVE)."""
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not ALWAYS_COLLECTIVE)


def _host_staged(t: torch.Tensor) -> bool:
    """Device tensors on a ``gloo`` process group (single-GPU tests of the multi-rank path: several ranks share cuda:0 and the
    collectives run over gloo) are staged through host memory; on ``nccl`` (= RCCL) they are reduced in place over xGMI."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce_sum(t: torch.Tensor, group=None) -> None:
    if _host_staged(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_gather(t: torch.Tensor, n: int, group=None) -> List[torch.Tensor]:
    """n equally shaped tensors, one per rank of the group."""
    t = t.contiguous()
    if _host_staged(t):
        h = t.detach().cpu()
        out = [torch.empty_like(h) for _ in range(n)]
        dist.all_gather(out, h, group=group)
        return [o.to(t.device) for o in out]
    out = [torch.empty_like(t) for _ in range(n)]
    dist.all_gather(out, t, group=group)
    return out


def shard_range(n_examples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous example-major shard [lo, hi) of a global batch (sizes differ by at most one)."""
    base, rem = divmod(n_examples, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_TENSOR_KEYS = ("qas_bert", "qas_mask", "sub_bert", "sub_mask", "vid", "vid_mask", "target", "ts_label_mask", "qas")
_LIST_KEYS = ("qid", "vid_name", "anno_st_idx", "q_l", "image_indices", "boxes", "att_labels", "target_list")


def shard_batch(batch, rank: int, world: int):
    """Slice every per-example field of a ``prepare_inputs``-style batch (tvqa_dataset.py:631-688)."""
    n = len(batch["qid"])
    lo, hi = shard_range(n, rank, world)
    out = type(batch)()
    for k, v in batch.items():
        if k in _TENSOR_KEYS and torch.is_tensor(v):
            out[k] = v[lo:hi]
        elif k == "ts_label" and isinstance(v, dict):
            out[k] = {kk: vv[lo:hi] for kk, vv in v.items()}
        elif k in _LIST_KEYS and isinstance(v, list):
            out[k] = v[lo:hi]
        elif k == "mask_host" and isinstance(v, dict):     # host copies of the word / frame validity (tvqaplus_amd/ragged.py)
            out[k] = {kk: vv[lo:hi] for kk, vv in v.items()}
        else:
            out[k] = v
    return out


def all_gather_outputs(local: torch.Tensor, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Concatenate a per-example output over ranks (dim 0).  ``counts``: rows held by each rank when they differ."""
    if _single():
        return local
    world = dist.get_world_size()
    if counts is None or len(set(counts)) == 1:
        return torch.cat(_all_gather(local, world), dim=0)
    mx = max(counts)
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    out = _all_gather(pad, world)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


class FlatGradBucket:
    """All parameter gradients through one contiguous buffer -> ONE all-reduce per step.

    ``zero()`` drops the gradients (``p.grad = None``, what ``optimizer.zero_grad()`` does): autograd then ASSIGNS the new
    gradient of a parameter instead of launching one accumulate-add kernel per parameter into a zeroed view (117 small
    kernels per step at hsz=128).  ``all_reduce()`` packs them into the flat buffer with one multi-tensor copy, reduces,
    and points every ``p.grad`` at its slice of the buffer (with one rank it only packs)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        # one trailing element travels with the gradients: the number of ranks whose set of gradient-carrying parameters changed in
        # this step (_drop_absent)
        self._buf = torch.zeros(total + 1, dtype=ref.dtype, device=ref.device)
        self.flat = self._buf[:total]          # the gradients (what callers and tests see)
        self._changed_pending = None
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off: off + p.numel()].view_as(p))
            off += p.numel()

    def zero(self) -> None:
        for p in self.params:
            p.grad = None

    def reset_absent(self) -> None:
        """Forget which parameters were agreed to be structurally without gradient (call on EVERY rank after enabling or
        disabling a model branch mid-run); the next ``all_reduce`` agrees again."""
        self._absent = None
        self._absent_basis = None

    def pack(self) -> None:
        """Copy the current gradients into the flat buffer and make every ``p.grad`` a view of it.  Parameters without a
        gradient contribute zeros to the buffer; whether such a parameter receives ``p.grad`` afterwards is decided by
        ``_absent_everywhere`` (the reference leaves the never-used t_iter > 0 layers at ``grad = None``, so Adam with
        weight decay does not touch them -- zero gradients would decay them, only when world > 1)."""
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for v, p in zip(self.views, self.params) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._present_local = [p.grad is not None for p in self.params]
        for v, p in zip(self.views, self.params):
            p.grad = v

    def _drop_absent(self) -> None:
        """Parameters that have no gradient on ANY rank keep ``grad = None``.  Which ones is a property of the model
        structure, so it is agreed once (one small all-reduce + host read at the first step) and cached together with the
        LOCAL presence vector it was derived from.  When that vector changes later (a branch enabled after warm-up, a first
        step that skipped a path) the cached decision would silently drop a real gradient: with one rank the decision is
        simply remade (host-only comparison, no device read); with several ranks a one-sided extra collective would
        desynchronise them (and a one-sided exception leaves the peers hanging in their next collective), so the change is REPORTED
        through the gradient all-reduce itself -- one extra element of the flat buffer -- and EVERY rank raises at its next step (the
        step in between applied the old agreement everywhere: the replicas have not diverged)."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        if world == 1 and getattr(self, "_absent", None) is not None and self._present_local != self._absent_basis:
            self._absent = None
        if getattr(self, "_absent", None) is None:
            flags = torch.tensor([1.0 if f else 0.0 for f in self._present_local], device=self.flat.device)
            if world > 1:
                _all_reduce_sum(flags)
            self._absent = [f == 0.0 for f in flags.tolist()]
            self._absent_basis = list(self._present_local)
        for absent, p in zip(self._absent, self.params):
            if absent:
                p.grad = None

    def clip_grad_norm_(self, max_norm: float, eps: float = 1e-6) -> torch.Tensor:
        """``torch.nn.utils.clip_grad_norm_(params, max_norm)`` (main.py:63) on the packed buffer: after ``all_reduce`` every gradient is a
        view of ``flat`` (parameters without a gradient hold zeros there), so the total 2-norm is ONE reduction and the scaling ONE
        multiply -- three small kernels instead of torch's per-tensor norms / stack / norm / clamp / multi-tensor multiply (0.4 ms of
        host time per step: it decides the step when a rank holds two examples).  Same value up to summation order; returns the norm."""
        # precondition: every gradient IS a view of ``flat`` (pack() / all_reduce() ran after this step's backward).  Called earlier --
        # or after a backward that produced fresh p.grad tensors -- the norm would be the previous step's buffer and the real
        # gradients would stay unclipped: pack first (pointer comparisons only, no device work when the precondition holds)
        if any(p.grad is not None and p.grad.data_ptr() != v.data_ptr() for v, p in zip(self.views, self.params)):
            self.pack()
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(float(max_norm) / (total + eps), max=1.0))
        return total

    def flush(self) -> None:
        """Raise NOW if the last ``all_reduce`` reported a changed gradient set (otherwise the report of the LAST step of a run would
        never be read: call after the training loop, on every rank)."""
        self._raise_if_changed()

    def all_reduce(self) -> None:
        """Pack (always: ``flat`` is valid with one rank too), sum over ranks, drop the structurally absent gradients."""
        self._raise_if_changed()
        self.pack()
        self._reduce_packed()

    def _raise_if_changed(self) -> None:
        if self._changed_pending is not None:
            # what the PREVIOUS step's all-reduce said (its read-back has long arrived): some rank's set changed -> every rank sees the
            # same count here, before this step's collective, and raises together
            host, ev = self._changed_pending
            if ev is not None:
                ev.synchronize()
            self._changed_pending = None
            if float(host[0]) > 0.0:
                raise RuntimeError("FlatGradBucket: the set of parameters that receive a gradient changed on %d rank(s) in the previous "
                                   "step (which applied the old agreement on every rank, so the replicas are still identical); call "
                                   "bucket.reset_absent() on every rank when enabling / disabling a model branch" % int(round(float(host[0]))))

    def _reduce_packed(self) -> None:
        if not _single():
            changed = getattr(self, "_absent", None) is not None and self._present_local != self._absent_basis
            self._buf[-1:].fill_(1.0 if changed else 0.0)
            _all_reduce_sum(self._buf)
            if self._buf.is_cuda:
                host = getattr(self, "_changed_host", None)        # ONE pinned word, reused: the previous read was consumed above
                if host is None or host.dtype != self._buf.dtype:
                    host = self._changed_host = torch.empty(1, dtype=self._buf.dtype, pin_memory=True)
                host.copy_(self._buf[-1:], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self._buf.device))
                self._changed_pending = (host, ev)
            else:
                self._changed_pending = (self._buf[-1:].clone(), None)
        self._drop_absent()

# ---------------------------------------------------------------------------------------------------------------
# reference-faithful loss normalisation across ranks (main.py:57-60)
# ---------------------------------------------------------------------------------------------------------------
def global_loss_scale(n_examples_local: int, n_targets_local: int, device=None, count_this_rank: bool = True,
                      as_tensor: bool = False):
    """len(qids) / len(targets) of the GATHERED batch: one 2-element all-reduce.  ``count_this_rank=False`` for the ranks of
    a candidate group that replicate the examples of the group's first rank.  ``as_tensor=True`` returns the ratio as a 0-dim
    tensor on ``device`` WITHOUT reading it back: the training step then has no host synchronisation between forward and backward
    (the collective is stream-ordered on RCCL)."""
    if _single():
        r = float(n_examples_local) / float(n_targets_local)
        return torch.tensor(r, dtype=torch.float32, device=device) if as_tensor else r
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    v = torch.tensor([float(n_examples_local), float(n_targets_local)], dtype=torch.float64, device=device)
    if not count_this_rank:
        v.zero_()
    _all_reduce_sum(v)
    if as_tensor:
        return (v[0] / v[1]).to(torch.float32)
    return float(v[0].item() / v[1].item())


# ---------------------------------------------------------------------------------------------------------------
# candidate x batch layout (more ranks than examples)
# ---------------------------------------------------------------------------------------------------------------
NUM_CANDIDATES = 5   # model/stage.py:79
_GROUP_CACHE = {}    # (E, C, world, id of the default group) -> its candidate groups (communicators are never rebuilt)


def reset_groups() -> None:
    """Forget the cached candidate groups (after ``dist.destroy_process_group()``: they belong to the process group that is gone)."""
    _GROUP_CACHE.clear()


class _GroupCE(torch.autograd.Function):
    """Sum cross-entropy over examples whose five logits are spread over the ranks of a candidate group.
    forward: all-gather the local columns -> full rows -> CE(sum) (the same value on every rank of the group);
    backward: (softmax - onehot) restricted to the local columns.  Each rank back-propagates only through its own
    candidates, so the flat sum all-reduce of the parameter gradients counts every path once."""

    @staticmethod
    def forward(ctx, local_logits, targets, layout):
        full = layout.gather_candidates(local_logits.detach())                 # (M, 5)
        logp = torch.log_softmax(full.double(), dim=1)
        ctx.save_for_backward(logp, targets)
        ctx.cols = layout.cand_range
        return -(logp.gather(1, targets.view(-1, 1)).sum()).to(local_logits.dtype)

    @staticmethod
    def backward(ctx, gout):
        logp, targets = ctx.saved_tensors
        g = torch.exp(logp)
        g.scatter_add_(1, targets.view(-1, 1), -torch.ones_like(g[:, :1]))
        k0, k1 = ctx.cols
        return (g[:, k0:k1] * gout).to(gout.dtype), None, None


class CandidateLayout:
    """world = E x C: E example blocks (example-major, as ``shard_range``), C <= 5 ranks per block splitting the five
    candidates contiguously.  rank r -> block r // C, candidate part r % C; ranks >= E*C idle (``active`` False).

    ``shard(batch)`` returns the local batch: the example slice, ``qas_bert`` / ``qas_mask`` / ``qas`` restricted to the
    local candidates, ``target`` kept GLOBAL, plus ``cand_offset`` (global index of local candidate 0) and ``gt_scores_fn``
    which the model calls in add_local training to obtain the ground-truth candidate's temporal scores
    (tvqaplus_amd.stage.STAGE.get_proposals), and ``dropout_rank`` (the example block: the replicated context encoders of a
    group share one dropout stream, as the five candidates of an example share one context mask in the reference).
    ``global_loss_scale`` costs one blocking host read per step (the data-dependent proposal count of add_local)."""

    def __init__(self, n_examples: int, rank: Optional[int] = None, world: Optional[int] = None):
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        if n_examples <= 0:
            raise ValueError("CandidateLayout needs at least one example (got n_examples=%d)" % n_examples)
        self.rank, self.world, self.n_examples = rank, world, n_examples
        self.E = min(n_examples, world)
        self.C = max(1, min(NUM_CANDIDATES, world // self.E))
        self.active = rank < self.E * self.C
        self.block = rank // self.C if self.active else -1
        self.part = rank % self.C if self.active else -1
        self.example_range = shard_range(n_examples, self.block, self.E) if self.active else (0, 0)
        self.cand_range = shard_range(NUM_CANDIDATES, self.part, self.C) if self.active else (0, 0)
        self.group = None
        self.group_ranks = list(range(self.block * self.C, (self.block + 1) * self.C)) if self.active else []
        if dist.is_initialized() and world > 1 and self.C > 1:
            # new_group is collective over the world: every rank creates every block's group -- ONCE per (E, C) shape of the
            # process group (a layout built per batch, e.g. for a smaller last batch, reuses the communicators)
            key = (self.E, self.C, world, id(dist.group.WORLD))
            groups = _GROUP_CACHE.get(key)
            if groups is None:
                groups = [dist.new_group(ranks=list(range(b * self.C, (b + 1) * self.C))) for b in range(self.E)]
                _GROUP_CACHE[key] = groups
            if self.active:
                self.group = groups[self.block]

    # ---- data --------------------------------------------------------------------------------------------------
    def shard(self, batch):
        lo, hi = self.example_range
        k0, k1 = self.cand_range
        out = type(batch)()
        for k, v in batch.items():
            if k in ("qas_bert", "qas_mask", "qas") and torch.is_tensor(v):
                out[k] = v[lo:hi, k0:k1].contiguous()
            elif k in _TENSOR_KEYS and torch.is_tensor(v):
                out[k] = v[lo:hi]
            elif k == "ts_label" and isinstance(v, dict):
                out[k] = {kk: vv[lo:hi] for kk, vv in v.items()}
            elif k in _LIST_KEYS and isinstance(v, list):
                out[k] = v[lo:hi]
            elif k == "mask_host" and isinstance(v, dict):
                out[k] = {kk: (vv[lo:hi, k0:k1] if kk == "qas" else vv[lo:hi]) for kk, vv in v.items()}   # qas (N, NA, Lqa); *_len (N, Li)
            else:
                out[k] = v
        out["cand_offset"] = k0
        out["gt_scores_fn"] = self.gt_scores
        # dropout stream of the example block: the ranks of a candidate group replicate the same context encoders and must
        # drop the same units there (STAGE._seed); the QA side holds different candidates per rank, so sharing is harmless
        out["dropout_rank"] = self.block
        return out

    # ---- collectives inside the group ----------------------------------------------------------------------------
    def gather_candidates(self, local: torch.Tensor) -> torch.Tensor:
        """local (M, k_local, ...) -> (M, 5, ...) on every rank of the group (candidate parts differ by at most one)."""
        if self.C == 1 or self.group is None:
            return local
        kmax = max(shard_range(NUM_CANDIDATES, p, self.C)[1] - shard_range(NUM_CANDIDATES, p, self.C)[0] for p in range(self.C))
        pad = local.new_zeros((local.shape[0], kmax) + tuple(local.shape[2:]))
        pad[:, : local.shape[1]] = local
        parts = _all_gather(pad, self.C, group=self.group)
        cols = []
        for p, t in enumerate(parts):
            a, b = shard_range(NUM_CANDIDATES, p, self.C)
            cols.append(t[:, : b - a])
        return torch.cat(cols, dim=1)

    def gt_scores(self, t_scores_local: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """(N, k_local, Li, 2) masked temporal scores of the local candidates + global targets (N,) -> (N, Li, 2) of the
        ground-truth candidate (model/stage.py:408-409), identical on every rank of the group."""
        full = self.gather_candidates(t_scores_local.detach())
        return full[torch.arange(full.shape[0], device=full.device), targets]

    def cross_entropy_sum(self, local_logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        if self.C == 1:
            return torch.nn.functional.cross_entropy(local_logits, targets, reduction="sum")
        return _GroupCE.apply(local_logits, targets, self)

    def gather_logits_world(self, local_logits: torch.Tensor) -> torch.Tensor:
        """(N_loc, k_local) -> (N, 5) on every rank of the world: candidates inside the group, then the blocks (taken from
        each block's first rank)."""
        full = self.gather_candidates(local_logits.detach()) if self.active else local_logits.new_zeros((0, NUM_CANDIDATES))
        if not dist.is_initialized() or self.world == 1:
            return full
        counts = [shard_range(self.n_examples, b, self.E)[1] - shard_range(self.n_examples, b, self.E)[0] for b in range(self.E)]
        mx = max(counts)
        pad = full.new_zeros((mx, NUM_CANDIDATES))
        pad[: full.shape[0]] = full
        out = _all_gather(pad, self.world)
        return torch.cat([out[b * self.C][: counts[b]] for b in range(self.E)], dim=0)
