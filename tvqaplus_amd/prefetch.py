"""Host -> device input pipeline for ``prepare_inputs``-style batches (tvqa_dataset.py:631-688).

One training step consumes 862 MB of fp32 features at the full configuration (``sub_bert`` alone is 737 MB), i.e. >= 13.7 ms
over PCIe Gen5 x16 -- more than half of the 24 ms step.  ``BatchPrefetcher`` moves batch i+1 while step i computes:

* every tensor is staged through a reusable PINNED host buffer (two sets, ping-pong) so the copy is a true async DMA,
* the copies are enqueued on a side stream, an event marks the batch complete, the consumer's stream waits on the event
  (no host synchronisation anywhere),
* ``record_stream`` tells the caching allocator that the device tensors are used on the consumer's stream.

Non-tensor fields (qid lists, att_labels, boxes, ...) pass through untouched.

``feature_dtype=torch.bfloat16`` (for a model built with ``opt.storage_dtype = "bf16"``) rounds the three feature tensors
(``qas_bert``, ``sub_bert``, ``vid``) to bf16 while they are copied into the pinned staging buffers: the model would round them
on entry anyway, and the transfer is 431 MB instead of 862 MB.  Masks, labels and indices keep their types.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional

import torch


class BatchPrefetcher:
    FEATURES = ("qas_bert", "sub_bert", "vid")

    def __init__(self, batches: Iterable, device, depth: int = 2, feature_dtype: Optional[torch.dtype] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchPrefetcher needs a GPU (the HIP path has no CPU fallback)")
        self.src = iter(batches)
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self.depth = max(2, int(depth))
        if feature_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("feature_dtype must be None, torch.float32 or torch.bfloat16")
        self.feature_dtype = feature_dtype
        self._pinned: list = [dict() for _ in range(self.depth)]   # per slot: key -> pinned staging tensor
        self._free_evt: list = [None] * self.depth                  # per slot: event after which the staging set is reusable
        self._slot = 0
        self._queue: list = []
        for _ in range(self.depth - 1):
            self._enqueue()

    # -- staging ------------------------------------------------------------------------------------------------------
    def _stage(self, slot: int, key: str, t: torch.Tensor) -> torch.Tensor:
        dt = t.dtype
        if self.feature_dtype is not None and key in self.FEATURES and t.is_floating_point():
            dt = self.feature_dtype
        if t.is_cuda:
            return t if t.dtype == dt else t.to(dt)
        if t.is_pinned() and t.dtype == dt:
            return t
        buf: Dict[str, torch.Tensor] = self._pinned[slot]
        p = buf.get(key)
        if p is None or p.shape != t.shape or p.dtype != dt:
            p = torch.empty(t.shape, dtype=dt, pin_memory=True)
            buf[key] = p
        p.copy_(t)                                   # converts while it stages when the feature type differs
        return p

    def _enqueue(self) -> bool:
        try:
            host = next(self.src)
        except StopIteration:
            return False
        slot = self._slot
        self._slot = (self._slot + 1) % self.depth
        if self._free_evt[slot] is not None:
            self._free_evt[slot].synchronize()       # the DMA that last read this staging set has finished
        out = type(host)()
        with torch.cuda.stream(self.side):
            for k, v in host.items():
                if torch.is_tensor(v):
                    out[k] = self._stage(slot, k, v).to(self.device, non_blocking=True)
                elif isinstance(v, dict):
                    out[k] = {kk: (self._stage(slot, k + "." + kk, vv).to(self.device, non_blocking=True)
                                   if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                else:
                    out[k] = v
            done = torch.cuda.Event()
            done.record(self.side)
        # host copy of the answer indices for the supervised-attention index building (att_host.build_att_pairs): reading
        # them back from the device tensor inside the step would drain the whole launch queue
        tgt = host.get("target") if hasattr(host, "get") else None
        if torch.is_tensor(tgt) and not tgt.is_cuda and "target_list" not in out:
            out["target_list"] = tgt.tolist()
        self._free_evt[slot] = done
        self._queue.append((out, done))
        return True

    # -- iteration ----------------------------------------------------------------------------------------------------
    def __iter__(self) -> Iterator:
        return self

    def __next__(self):
        if not self._queue:
            if not self._enqueue():
                raise StopIteration
        batch, done = self._queue.pop(0)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(cur)
            elif isinstance(v, dict):
                for vv in v.values():
                    if torch.is_tensor(vv):
                        vv.record_stream(cur)
        self._enqueue()                              # start the next transfer now: it overlaps the caller's step
        return batch
