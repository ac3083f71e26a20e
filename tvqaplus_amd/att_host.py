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
    pos_rep = np.tile(pr, num_negatives)
    if not hard:
        return pos_rep, nr[_draw(nr.shape[0], pos_rep.shape[0])]
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


def build_att_pairs(model, batch, scores: Optional[torch.Tensor] = None, n_local_candidates: Optional[int] = None):
    """Host half of ``get_att_loss`` (model/stage.py:612-694): the (positive, sampled negative) index pairs of the batch as
    two (M, 5) int64 arrays of (batch, answer, image, word, region) rows, in the reference's order and with the reference's
    random draws.  Random-negative mode needs no scores (it can run ahead of the device, e.g. in the data loader);
    hard-negative mode reads the predicted scores of the labelled words -- ONE gather + ONE device-to-host copy per
    batch instead of the reference's sort + ``.cpu()`` per word."""
    targets = batch.target.tolist()
    hard = bool(getattr(batch, "use_hard_negatives", False))
    k0 = int(getattr(batch, "cand_offset", 0) or 0)      # candidate-sharded batches: only locally held ground truths contribute
    labels = _labels_to_host(batch.att_labels)
    entries = []                                           # (b, ca_local, img, word, pr, nr)
    for b, ca in enumerate(targets):
        ca -= k0
        if n_local_candidates is not None and (ca < 0 or ca >= n_local_candidates):
            continue
        start = int(batch.anno_st_idx[b])
        for local, lab in enumerate(labels[b]):
            nz = lab != 0
            rows = np.flatnonzero(nz.any(axis=1))
            for w in rows.tolist():
                entries.append((b, ca, start + local, w, np.flatnonzero(nz[w]), np.flatnonzero(~nz[w])))
    if not entries:
        return None, None
    pred_rows = None
    if hard:
        if scores is None:
            raise ValueError("hard-negative sampling ranks the negatives by their predicted scores: pass `scores`")
        idx = torch.tensor([e[:4] for e in entries], dtype=torch.long)
        idx = idx.to(scores.device, non_blocking=True)
        pred_rows = scores.detach()[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]].cpu().numpy()    # (K, Lr): the one sync
    pos_chunks, neg_chunks = [], []
    for k, (b, ca, img, w, pr, nr) in enumerate(entries):
        sp, sn = _sample_negatives_np(None if pred_rows is None else pred_rows[k], pr, nr, model.num_negatives, hard,
                                      model.negative_pool_size, model.num_hard, model.drop_topk)
        head = np.empty((sp.shape[0], 4), dtype=np.int64)
        head[:] = (b, ca, img, w)
        pos_chunks.append(np.concatenate([head, sp[:, None].astype(np.int64)], axis=1))
        neg_chunks.append(np.concatenate([head, sn[:, None].astype(np.int64)], axis=1))
    return np.concatenate(pos_chunks, axis=0), np.concatenate(neg_chunks, axis=0)


def get_att_loss(model, scores: torch.Tensor, batch):
    """scores (N,5,Li,Lqa,Lr) raw cosine scores; batch.att_labels: per item a list (per annotated image) of
    (num_words, num_regions) 0/1 tensors; batch.anno_st_idx: index of the first annotated image.
    ``batch.att_pairs`` (a ``build_att_pairs`` result prepared ahead, random-negative mode) is used when present."""
    pairs = getattr(batch, "att_pairs", None)
    if pairs is None:
        pairs = build_att_pairs(model, batch, scores, n_local_candidates=scores.shape[1])
    pos, neg = pairs
    if pos is None:
        return scores.sum() * 0.0, None
    # one host->device copy for both index sets, then ONE flat gather: the gradient reaches raw_s as a sparse scatter
    both = torch.from_numpy(np.concatenate([pos, neg], axis=0))
    if scores.is_cuda:
        both = both.pin_memory().to(scores.device, non_blocking=True)
    st = scores.stride()
    flat_idx = both[:, 0] * st[0] + both[:, 1] * st[1] + both[:, 2] * st[2] + both[:, 3] * st[3] + both[:, 4] * st[4]
    vals = scores.reshape(-1).index_select(0, flat_idx) if scores.is_contiguous() else \
        scores[both[:, 0], both[:, 1], both[:, 2], both[:, 3], both[:, 4]]
    m = pos.shape[0]
    s_pos, s_neg = vals[:m], vals[m:]
    if model.att_loss_type == "hinge":
        loss = torch.clamp(model.margin + s_neg - s_pos, min=0).sum()
    elif model.att_loss_type == "lse":
        loss = torch.log1p(torch.exp(model.alpha * (s_neg - s_pos))).sum()
    else:
        raise NotImplementedError("Only support hinge and lse")
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
