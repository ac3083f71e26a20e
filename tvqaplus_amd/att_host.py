"""Host-side consumers of the attention map ``vid_raw_s``.

* ``get_att_loss``  -- supervised ranking loss over (positive region, sampled negative region) pairs of the
  ground-truth answer (model/stage.py:557-746).  Index lists are built on the host once per batch; the loss itself is
  one gather + one fused elementwise reduction on the device tensor, so autograd sends a sparse gradient into
  ``raw_s`` (the ``dS_raw_ext`` input of ``stage_str_attn_bwd``).
* ``get_att_prediction`` -- box predictions for inference (model/stage.py:748-806): per (question, answer, annotated
  image, object word) the regions whose cosine score >= 0.2, ascending by score.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch


def _draw(high: int, n: int) -> np.ndarray:
    """One ``torch.randint(0, high, (n,))`` on the default CPU generator: the reference's ``sample_negatives`` draws with
    exactly this call per (image, word) pair, so issuing the same calls in the same order reproduces its negatives."""
    return torch.randint(0, high, (n,)).numpy()


def _sample_negatives_np(row: Optional[np.ndarray], pr: np.ndarray, nr: np.ndarray, num_negatives: int, hard: bool,
                         pool: int, num_hard: int, drop_topk: int) -> Tuple[np.ndarray, np.ndarray]:
    """model/stage.py:557-611 for ONE (image, word) pair.  pr / nr: positive / negative region indices; ``row``: the
    predicted scores of the word's regions (hard mode only).  Returns the region indices of (P*num_negatives) pairs."""
    n_pos = pr.shape[0]
    pos_rep = pr if num_negatives == 1 else np.concatenate((pr,) * num_negatives)      # == pos.repeat(num_negatives, 1)
    if not hard:
        return pos_rep, nr[torch.randint(0, nr.shape[0], (pos_rep.shape[0],)).numpy()]
    # descending by predicted score (torch.sort(descending=True) at :577; scores are distinct floats in practice)
    order = nr[np.argsort(-row[nr], kind="stable")]
    if pool > num_negatives:
        hard_pool = order[drop_topk:drop_topk + pool]
        n_hard = num_negatives
        easy_part = None
        if num_hard < num_negatives:
            easy_pool = order[drop_topk + pool:]
            n_hard = num_hard
            easy_part = easy_pool[_draw(easy_pool.shape[0], (num_negatives - n_hard) * n_pos)]
        hard_part = hard_pool[_draw(hard_pool.shape[0], n_hard * n_pos)]
        return pos_rep, (hard_part if easy_part is None else np.concatenate([hard_part, easy_part]))
    return pos_rep, order[drop_topk:drop_topk + pos_rep.shape[0]]


def _labels_to_host(att_labels) -> List[List[np.ndarray]]:
    """att_labels: per item a list (per annotated image) of (num_words, num_regions) 0/1 tensors.  They are built on the
    host by the data set; if a caller moved them to the device they come back in ONE copy per item."""
    out = []
    for per in att_labels:
        if len(per) and torch.is_tensor(per[0]) and per[0].is_cuda:
            per = list(torch.stack(list(per)).cpu())
        out.append([(l.detach().numpy() if torch.is_tensor(l) else np.asarray(l)) for l in per])
    return out


def targets_on_device_ok(model, batch, n_local_candidates: Optional[int]) -> bool:
    """True when the pair lists can be built WITHOUT knowing the answer indices on the host: the batch carries no host copy of them
    (``target_list``: a batch straight from the reference's prepare_inputs, tvqa_dataset.py:631-688, has only the device tensor), every
    candidate is local (the answer index then only selects a slice of the score tensor, it filters nothing) and the negatives are drawn at
    random (hard negatives read the scores of the answer's rows).  ``AttPairs(..., target_dev=batch.target)`` then adds the answer's
    offset on the device -- instead of ``batch.target.tolist()``, which drains the whole queue (11-16 ms per step, DESIGN finding 18)."""
    if getattr(batch, "target_list", None) is not None or bool(getattr(batch, "use_hard_negatives", False)):
        return False
    if int(getattr(batch, "cand_offset", 0) or 0) != 0:
        return False
    t = getattr(batch, "target", None)
    if not (torch.is_tensor(t) and t.is_cuda):
        return False
    return n_local_candidates is None or int(n_local_candidates) == int(getattr(model, "num_a", n_local_candidates))


def build_att_pairs(model, batch, scores: Optional[torch.Tensor] = None, n_local_candidates: Optional[int] = None,
                    placeholder_targets: bool = False):
    """Host half of ``get_att_loss`` (model/stage.py:612-694): the (positive, sampled negative) index pairs of the batch as
    two (M, 5) int64 arrays of (batch, answer, image, word, region) rows, in the reference's order and with the reference's
    random draws.  Random-negative mode needs no scores (it can run ahead of the device, e.g. in the data loader);
    hard-negative mode reads the predicted scores of the labelled words -- ONE gather + ONE device-to-host copy per
    batch instead of the reference's sort + ``.cpu()`` per word."""
    # host copy of the answer indices when the input pipeline kept one (`.tolist()` of a device tensor waits for the whole
    # queue: measured 11-16 ms inside the step -- a full host/device serialisation per batch, which the reference also pays)
    targets = getattr(batch, "target_list", None)
    if placeholder_targets:      # ``targets_on_device_ok``: candidate column 0 everywhere, the device adds the answer's offset
        targets = [0] * len(batch.att_labels)
    elif targets is None:
        targets = batch.target.tolist()
    targets = list(targets)
    hard = bool(getattr(batch, "use_hard_negatives", False))
    k0 = int(getattr(batch, "cand_offset", 0) or 0)      # candidate-sharded batches: only locally held ground truths contribute
    labels = _labels_to_host(batch.att_labels)
    heads, prs, nrs = [], [], []                           # per (image, word) entry: (b, ca_local, img, word), pos / neg regions
    for b, ca in enumerate(targets):
        ca -= k0
        if n_local_candidates is not None and (ca < 0 or ca >= n_local_candidates):
            continue
        start = int(batch.anno_st_idx[b])
        for local, lab in enumerate(labels[b]):
            nz = lab != 0
            rows = np.flatnonzero(nz.any(axis=1))
            if rows.size == 0:
                continue
            sub = nz[rows]                                  # (#labelled words, Lr): one nonzero each way for the image
            pw, pr_all = np.nonzero(sub)
            nw, nr_all = np.nonzero(~sub)
            pcut = np.searchsorted(pw, np.arange(rows.size + 1)).tolist()
            ncut = np.searchsorted(nw, np.arange(rows.size + 1)).tolist()
            img = start + local
            for j, w in enumerate(rows.tolist()):
                prs.append(pr_all[pcut[j]:pcut[j + 1]])
                nrs.append(nr_all[ncut[j]:ncut[j + 1]])
                heads.append((b, ca, img, w))
    if not heads:
        return None, None
    pred_rows = None
    if hard:
        if scores is None:
            raise ValueError("hard-negative sampling ranks the negatives by their predicted scores: pass `scores`")
        idx = torch.tensor(heads, dtype=torch.long).to(scores.device, non_blocking=True)
        pred_rows = scores.detach()[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]].cpu().numpy()    # (K, Lr): the one sync
    sps, sns = [], []
    for k in range(len(heads)):
        sp, sn = _sample_negatives_np(None if pred_rows is None else pred_rows[k], prs[k], nrs[k], model.num_negatives,
                                      hard, model.negative_pool_size, model.num_hard, model.drop_topk)
        sps.append(sp)
        sns.append(sn)
    counts = np.fromiter((a.shape[0] for a in sps), dtype=np.int64, count=len(sps))
    head_rows = np.repeat(np.asarray(heads, dtype=np.int64), counts, axis=0)
    pos = np.concatenate([head_rows, np.concatenate(sps).astype(np.int64)[:, None]], axis=1)
    neg = np.concatenate([head_rows, np.concatenate(sns).astype(np.int64)[:, None]], axis=1)
    return pos, neg


class PinnedStage:
    """Two pinned host buffers for the per-step index upload, each guarded by the event of the copy that last read it: the host
    may only rewrite a buffer once the asynchronous host-to-device copy issued from it has run (a single unguarded buffer was a
    race whenever nothing else synchronised the step -- add_local off, or a loop with no per-step host read)."""

    def __init__(self):
        self.bufs = [None, None]
        self.events = [None, None]
        self.turn = 0

    def upload(self, t, device) -> torch.Tensor:
        """t: a host tensor or a numpy array (staged with numpy: a torch CPU copy of more than 32 K elements goes through the intra-op
        thread pool, and waking that pool inside the training step cost 50-130 ms every few steps on a 256-thread host)."""
        i = self.turn
        self.turn ^= 1
        if self.events[i] is not None:
            self.events[i].synchronize()           # the copy that read this buffer two steps ago has finished
        is_np = isinstance(t, np.ndarray)
        n = int(t.size) if is_np else t.numel()
        dt = torch.from_numpy(t[:0]).dtype if is_np else t.dtype
        if self.bufs[i] is None or self.bufs[i].numel() < n or self.bufs[i].dtype != dt:
            self.bufs[i] = torch.empty(max(2 * n, 4096), dtype=dt, pin_memory=True)
        if is_np:
            np.copyto(self.bufs[i].numpy()[:n], t.reshape(-1))
        else:
            self.bufs[i][:n].copy_(t)
        with torch.cuda.device(device):
            dev = self.bufs[i][:n].to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
        self.events[i] = ev
        return dev


class AttPairs:
    """Flat indices into a contiguous (N, NA, Li, Lqa, Lr) score tensor for the M positive rows followed by the M negative
    rows, resident on the device.  ``stage`` = a ``PinnedStage`` reused across steps (``pin_memory()`` per call registers
    a fresh page-locked allocation every time: ~0.5 ms of host time on the launch path)."""

    def __init__(self, pos: np.ndarray, neg: np.ndarray, shape, device, stage: Optional[PinnedStage] = None,
                 target_dev: Optional[torch.Tensor] = None):
        """``target_dev`` (N,) int64 on the device: the pairs were built with candidate 0 as a placeholder
        (``build_att_pairs(placeholder_targets=True)``); the answer's slice offset is added here, on the device, without a host read."""
        both = np.concatenate([pos, neg], axis=0)
        _, NA, Li, Lqa, Lr = shape
        if both.size and ((both < 0).any() or (both >= np.asarray([shape[0], NA, Li, Lqa, Lr])).any()):
            raise IndexError("attention-loss pair outside the (N, NA, Li, Lqa, Lr) score tensor")    # (the kernels gather / scatter unchecked)
        flat = (((both[:, 0] * NA + both[:, 1]) * Li + both[:, 2]) * Lqa + both[:, 3]) * Lr + both[:, 4]
        self.m = pos.shape[0]
        self.shape = tuple(shape)
        if target_dev is not None:
            flat = np.concatenate([flat, both[:, 0]])            # the example of every pair rides in the same upload
        t = torch.from_numpy(flat)
        if torch.device(device).type == "cuda":
            if stage is None:
                stage = PinnedStage()
            self.flat = stage.upload(t, device)
        else:
            self.flat = t
        if target_dev is not None:
            n2 = 2 * self.m
            tdev = target_dev.to(self.flat.device)
            ca = tdev.clamp(0, NA - 1).index_select(0, self.flat[n2:])     # (the kernels gather unchecked: the clamp is for memory safety)
            self.flat = self.flat[:n2] + ca * (Li * Lqa * Lr)
            # an answer index outside [0, NA) is an ERROR in the reference (scores[batch_idx, ca_idx] raises, model/stage.py:645) and on
            # the host path above; here nothing can raise without a read-back, so the loss is poisoned instead: get_att_loss multiplies
            # by this factor (1.0, or NaN when any target was out of range) -- the step fails loudly, not silently on the wrong candidate
            self.poison = torch.where(((tdev < 0) | (tdev >= NA)).any(), float("nan"), 1.0).to(torch.float32)
        else:
            self.poison = None
        self.stage = stage


def get_att_loss(model, scores: torch.Tensor, batch, pairs=None):
    """scores (N,5,Li,Lqa,Lr) raw cosine scores; batch.att_labels: per item a list (per annotated image) of
    (num_words, num_regions) 0/1 tensors; batch.anno_st_idx: index of the first annotated image.
    ``pairs``: an ``AttPairs`` prepared ahead (random-negative mode: STAGE.forward_main builds it before the first launch of
    the step) or a ``build_att_pairs`` result; built here otherwise (hard negatives rank by the scores)."""
    if pairs is None:
        pairs = getattr(batch, "att_pairs", None)
    if pairs is None:
        pairs = build_att_pairs(model, batch, scores, n_local_candidates=scores.shape[1])
    if not isinstance(pairs, AttPairs):
        pos, neg = pairs
        if pos is None:
            return scores.sum() * 0.0, None
        pairs = AttPairs(pos, neg, scores.shape, scores.device)
    assert pairs.shape == tuple(scores.shape), (pairs.shape, tuple(scores.shape))
    grouped = getattr(model, "_grouped", None)
    if grouped is not None and grouped() and scores.is_cuda and scores.dtype == torch.float32 and pairs.m > 0:
        # gather + loss + per-pair gradient coefficients in one kernel; the backward zero-fills and scatters (csrc/groups.hip)
        from . import groups
        try:
            loss = groups.att_loss(scores.contiguous(), pairs.flat, pairs.m, model.att_loss_type, model.alpha, model.margin)
            return (loss if pairs.poison is None else loss * pairs.poison), None
        except groups.Unsupported:
            pass      # the plain gather below
    # ONE flat gather: the gradient reaches raw_s as a sparse scatter
    vals = scores.contiguous().reshape(-1).index_select(0, pairs.flat)
    s_pos, s_neg = vals[: pairs.m], vals[pairs.m:]
    if model.att_loss_type == "hinge":
        loss = torch.clamp(model.margin + s_neg - s_pos, min=0).sum()
    elif model.att_loss_type == "lse":
        loss = torch.log1p(torch.exp(model.alpha * (s_neg - s_pos))).sum()
    else:
        raise NotImplementedError("Only support hinge and lse")
    if pairs.poison is not None:
        loss = loss * pairs.poison
    return loss, None  # att_predictions are only produced outside training in the reference (:702)


def get_att_prediction(scores: torch.Tensor, object_vocab, words: torch.Tensor, vid_names, qids, img_indices, boxes,
                       start_indices, score_thd: float = 0.2) -> Optional[list]:
    """model/stage.py:748-806."""
    vocab = set(int(w) for w in object_vocab)
    sc = scores.detach().cpu().numpy()
    wd = words.detach().cpu().numpy()
    out = []
    for b in range(sc.shape[0]):
        start = int(start_indices[b])
        per_q = {}
        for a in range(sc.shape[1]):
            dets = []
            for local, img_boxes in enumerate(boxes[b]):
                g = local + start
                for w_idx, w in enumerate(wd[b, a].tolist()):
                    if w not in vocab:
                        continue
                    row = sc[b, a, g, w_idx]
                    keep = np.nonzero(row >= score_thd)[0]
                    order = np.argsort(row[keep])
                    dets.append({"pred": [float(row[keep[i]]) for i in order],
                                 "bbox": [img_boxes[int(keep[i])] for i in order],
                                 "word": int(w), "qid": int(qids[b]), "vid_name": vid_names[b],
                                 "img_idx": img_indices[b][g]})
            per_q[a] = dets
        out.append(per_q)
    return out
