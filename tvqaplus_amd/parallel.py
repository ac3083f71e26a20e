"""One process per GPU over RCCL/xGMI.  The reference's only multi-device construct is
``nn.DataParallel`` (main.py:204-206) with ``bsz *= n_gpu`` (config.py:186-188): examples are independent, so the batch
is sharded example-major and exactly two collectives exist:

* ``all_gather_outputs``  -- logits (N_loc, 5) [+ t_scores] of every rank -> all ranks (forward, 320 B .. 192 KB)
* ``all_reduce_grads``    -- ONE flat bucket with every parameter gradient (552 947 fp32 = 2.2 MB at D=128), summed;
                             the losses are CE(sum), so sum-reduction reproduces the single-GPU gradient of the
                             concatenated batch (main.py:57-60, 208).

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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(n_examples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous example-major shard [lo, hi) of a global batch (sizes differ by at most one)."""
    base, rem = divmod(n_examples, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_TENSOR_KEYS = ("qas_bert", "qas_mask", "sub_bert", "sub_mask", "vid", "vid_mask", "target", "ts_label_mask", "qas")
_LIST_KEYS = ("qid", "vid_name", "anno_st_idx", "q_l", "image_indices", "boxes", "att_labels")


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
        else:
            out[k] = v
    return out


def all_gather_outputs(local: torch.Tensor, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Concatenate a per-example output over ranks (dim 0).  ``counts``: rows held by each rank when they differ."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if counts is None or len(set(counts)) == 1:
        out = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(out, local.contiguous())
        return torch.cat(out, dim=0)
    mx = max(counts)
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


class FlatGradBucket:
    """All parameter gradients through one contiguous buffer -> ONE all-reduce per step.

    ``zero()`` drops the gradients (``p.grad = None``, what ``optimizer.zero_grad()`` does): autograd then ASSIGNS the new
    gradient of a parameter instead of launching one accumulate-add kernel per parameter into a zeroed view (117 small
    kernels per step at hsz=128).  ``all_reduce()`` packs them into the flat buffer with one multi-tensor copy, reduces,
    and points every ``p.grad`` at its slice of the buffer; with one rank it only packs (``flat`` stays inspectable)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off: off + p.numel()].view_as(p))
            off += p.numel()

    def zero(self) -> None:
        for p in self.params:
            p.grad = None

    def pack(self) -> None:
        """Copy the current gradients into the flat buffer (parameters without a gradient contribute zeros) and make
        every ``p.grad`` a view of it."""
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for v, p in zip(self.views, self.params) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(self.views, self.params):
            p.grad = v

    def all_reduce(self) -> None:
        if dist.is_initialized() and dist.get_world_size() > 1:
            self.pack()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
