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


def _sample_negatives(pred: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, num_negatives: int, hard: bool,
                      pool: int, num_hard: int, drop_topk: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """model/stage.py:557-611.  pos (P,3), neg (Q,3) rows = (img, word, region).  Returns (P*num_negatives, 3) x2."""
    n_pos = pos.shape[0]
    pos_rep = pos.repeat(num_negatives, 1)
    if not hard:
        pick = torch.randint(0, neg.shape[0], (pos_rep.shape[0],))
        return pos_rep, neg[pick]
    order = torch.sort(pred[neg[:, 0], neg[:, 1], neg[:, 2]].detach().cpu(), descending=True)[1]
    if pool > num_negatives:
        hard_pool = neg[order[drop_topk:drop_topk + pool]]
        n_hard = num_negatives
        easy_part = None
        if num_hard < num_negatives:
            easy_pool = neg[order[drop_topk + pool:]]
            n_hard = num_hard
            easy_part = easy_pool[torch.randint(0, easy_pool.shape[0], ((num_negatives - n_hard) * n_pos,))]
        hard_part = hard_pool[torch.randint(0, hard_pool.shape[0], (n_hard * n_pos,))]
        return pos_rep, (hard_part if easy_part is None else torch.cat([hard_part, easy_part], dim=0))
    return pos_rep, neg[order[drop_topk:drop_topk + pos_rep.shape[0]]]


def get_att_loss(model, scores: torch.Tensor, batch):
    """scores (N,5,Li,Lqa,Lr) raw cosine scores; batch.att_labels: per item a list (per annotated image) of
    (num_words, num_regions) 0/1 tensors; batch.anno_st_idx: index of the first annotated image."""
    pos_rows: List[torch.Tensor] = []
    neg_rows: List[torch.Tensor] = []
    targets = batch.target.tolist()
    hard = bool(getattr(batch, "use_hard_negatives", False))
    for b, ca in enumerate(targets):
        labels = batch.att_labels[b]
        start = int(batch.anno_st_idx[b])
        pred = scores[b, ca]
        for local, lab in enumerate(labels):
            lab = lab.detach().cpu()
            if not bool((lab != 0).any()):
                continue
            img = start + local
            for w in torch.nonzero((lab != 0).any(dim=1)).flatten().tolist():
                pr = torch.nonzero(lab[w] != 0).flatten()
                nr = torch.nonzero(lab[w] == 0).flatten()
                pos = torch.stack([torch.full_like(pr, img), torch.full_like(pr, w), pr], dim=1)
                neg = torch.stack([torch.full_like(nr, img), torch.full_like(nr, w), nr], dim=1)
                sp, sn = _sample_negatives(pred, pos, neg, model.num_negatives, hard, model.negative_pool_size,
                                           model.num_hard, model.drop_topk)
                head = torch.tensor([[b, ca]]).expand(sp.shape[0], 2)
                pos_rows.append(torch.cat([head, sp], dim=1))
                neg_rows.append(torch.cat([head, sn], dim=1))
    pi = torch.cat(pos_rows, dim=0).to(scores.device)
    ni = torch.cat(neg_rows, dim=0).to(scores.device)
    s_pos = scores[pi[:, 0], pi[:, 1], pi[:, 2], pi[:, 3], pi[:, 4]]
    s_neg = scores[ni[:, 0], ni[:, 1], ni[:, 2], ni[:, 3], ni[:, 4]]
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
