"""Inference decoding, the prediction file and the TVQA+ metrics, Python 3.

The reference splits this over ``inference.py`` (span decoding ``find_max_pair`` :13-35, the prediction dictionary
:42-72, written with ``save_json`` :98) and ``eval/eval_tvqa_plus.py`` + ``eval/maskrcnn_voc`` (Python 2 only:
``dict.keys()[0]``, implicit relative imports, ``np.nanmean(dict.values())``).  Here:

* ``find_max_pair_batch``   -- the span decoder for a whole batch ON THE DEVICE (no ``.cpu()`` per question); the
                               host version ``find_max_pair`` is kept as the specification it is tested against
* ``PredictionWriter``      -- accumulates ``{"ts_answer": {qid: [[st, ed], answer]}, "raw_bbox": [...]}`` from the
                               model's ``inference_mode`` outputs with the reference's time mapping, and saves it
* ``compute_temporal_metrics``, ``detection_map`` (PASCAL-VOC AP as maskrcnn-benchmark's voc evaluation, numpy),
  ``load_annotation``, ``load_predictions``, ``evaluate_files`` / ``python -m tvqaplus_amd.evaluation`` -- QA Acc.,
  Grd. mAP, Temp. mIoU, ASA (eval/eval_sample.sh:5-9 lists the values of the reference's sample prediction file, which is
  not part of the public tree; tests/golden/eval_*.npz pin these functions against the reference's own code instead).
"""
from __future__ import annotations

import json
from collections import defaultdict
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

NUM_ANSWERS = 5
IMAGE_SIZE = (640, 360)       # eval/eval_tvqa_plus.py:118 (unused by the arithmetic: boxes are absolute pixels)


# ---------------------------------------------------------------------------------------------------------------
# span decoding (inference.py:13-35)
# ---------------------------------------------------------------------------------------------------------------
def find_max_pair(p1: Sequence[float], p2: Sequence[float]) -> Tuple[Tuple[int, int], float]:
    """(k1, k2), k1 <= k2, maximising p1[k1] * p2[k2]: one left-to-right sweep that carries the first arg-max of p1 so
    far; a later end position only wins with a strictly larger product; (0, 1) with value 0 when nothing is positive."""
    best, best_val, lead = (0, 1), 0.0, 0
    for i in range(len(p1)):
        if p1[lead] < p1[i]:
            lead = i
        prod = p1[lead] * p2[i]
        if prod > best_val:
            best, best_val = (lead, i), prod
    return best, float(best_val)


def find_max_pair_batch(p_st: torch.Tensor, p_ed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The same decoder for R rows at once, wherever the tensors live.  p_st, p_ed: (R, L) -> (st (R,), ed (R,), value (R,)).
    Tie rules of the sweep: the start carried to position i is the FIRST arg-max of p_st[:i+1]; the end is the FIRST
    position reaching the best product; rows whose best product is not positive return (0, 1, 0)."""
    R, L = p_st.shape
    run_max, _ = torch.cummax(p_st, dim=1)
    pos = torch.arange(L, device=p_st.device).expand(R, L)
    # position i starts a new strict maximum <=> p_st[i] > max(p_st[:i]); its index is carried forward
    prev = torch.cat([torch.full((R, 1), float("-inf"), device=p_st.device, dtype=p_st.dtype), run_max[:, :-1]], dim=1)
    lead = torch.cummax(torch.where(p_st > prev, pos, torch.zeros_like(pos)), dim=1)[0]
    prod = run_max * p_ed
    val, ed = prod.max(dim=1)                      # first maximal position (torch.max documents first-occurrence)
    ed = torch.argmax((prod == val.unsqueeze(1)).to(torch.int8), dim=1)   # explicit: the FIRST position equal to the maximum
    st = lead.gather(1, ed.unsqueeze(1)).squeeze(1)
    none = val <= 0
    st = torch.where(none, torch.zeros_like(st), st)
    ed = torch.where(none, torch.ones_like(ed), ed)
    val = torch.where(none, torch.zeros_like(val), val)
    return st, ed, val


class PredictionWriter:
    """inference.py:42-72: per batch, the predicted answer, the (st, ed) span of the PREDICTED answer decoded from the
    softmaxed temporal scores and mapped to seconds (frames are sampled at 0.5 fps: ``st * 2 + offset``,
    ``(ed + 1) * 2 + offset`` with ``offset = (image_indices[0] % 6) / 3``), plus the raw box predictions."""

    def __init__(self):
        self.predictions: Dict[str, object] = dict(ts_answer={}, raw_bbox=[])

    def add_batch(self, outputs: Mapping[str, object], qids: Sequence[int], image_indices: Sequence[Sequence[int]]) -> None:
        answer, t_scores = outputs["answer"], outputs["t_scores"]             # (N, 5), (N, 5, Li, 2) softmaxed over Li
        pred = answer.detach().max(1)[1]                                       # (N,)
        N, _, Li, _ = t_scores.shape
        picked = t_scores.detach().gather(1, pred.view(N, 1, 1, 1).expand(N, 1, Li, 2)).squeeze(1)    # (N, Li, 2)
        st, ed, _ = find_max_pair_batch(picked[:, :, 0], picked[:, :, 1])
        rows = torch.stack([st, ed, pred], dim=1).cpu().tolist()               # ONE copy per batch
        if outputs.get("att_predictions"):
            self.predictions["raw_bbox"] += outputs["att_predictions"]
        for qid, (s, e, a), img in zip(qids, rows, image_indices):
            offset = (img[0] % 6) / 3
            self.predictions["ts_answer"][str(qid)] = [[s * 2 + offset, (e + 1) * 2 + offset], int(a)]

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            json.dump(self.predictions, f)


# ---------------------------------------------------------------------------------------------------------------
# temporal metrics (eval/eval_tvqa_plus.py:14-69)
# ---------------------------------------------------------------------------------------------------------------
def temporal_iou(pred: Sequence[float], gt: Sequence[float]) -> float:
    inter = max(0, min(pred[1], gt[1]) - max(pred[0], gt[0]))
    hull = max(pred[1], gt[1]) - min(pred[0], gt[0])          # the reference divides by the hull, not the true union
    return 0 if hull == 0 else 1.0 * inter / hull


def compute_temporal_metrics(pred: Mapping, gt: Mapping) -> Dict[str, float]:
    """pred / gt: {qid: [[st, ed], answer_idx]} (key types may differ: json gives str, the annotation int)."""
    keys = sorted(pred.keys())
    cast = type(next(iter(gt.keys())))
    iou = np.array([temporal_iou(pred[k][0], gt[cast(k)][0]) for k in keys])
    right = np.array([pred[k][1] for k in keys]) == np.array([gt[cast(k)][1] for k in keys])
    res = {}
    for thd in np.arange(0.1, 1, 0.1):
        res["R@{:.2f}".format(thd)] = 1.0 * np.sum(iou >= thd) / len(iou)
    res["miou"] = 1.0 * np.sum(iou) / len(iou)
    res["ans_span_joint_acc@.5"] = 1.0 * np.sum(right * (iou >= 0.5)) / len(right)
    res["qa_acc"] = 1.0 * np.sum(right) / len(right)
    return res


# ---------------------------------------------------------------------------------------------------------------
# grounding mAP (eval/maskrcnn_voc/voc_eval.py + boxlist_ops.py, re-stated on plain arrays)
# ---------------------------------------------------------------------------------------------------------------
def _pairwise_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(N,4) x (M,4) float32 xyxy, inclusive pixel convention (+1 on widths / heights)."""
    area_a = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    lt = np.maximum(a[:, None, :2], b[:, :2])
    rb = np.minimum(a[:, None, 2:], b[:, 2:])
    wh = np.clip(rb - lt + 1, 0, None)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area_a[:, None] + area_b - inter)


def _image_table(boxes_by_image: Mapping[str, list], w2i: Mapping[str, int]) -> Dict[str, Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """{image: [[label, score, xyxy], ...]} -> {image: (labels, scores, boxes float32)} without the <unk> labels; images
    that keep no box are dropped (eval_tvqa_plus.py:94-125)."""
    unk = w2i["<unk>"]
    out = {}
    for name, items in boxes_by_image.items():
        lab = np.array([w2i.get(e[0], unk) for e in items])
        keep = [i for i, l in enumerate(lab) if int(l) != unk]
        if not keep:
            continue
        out[name] = (lab[keep], np.array([items[i][1] for i in keep]),
                     np.array([items[i][2] for i in keep], dtype=np.float32))
    return out


def detection_map(pred_by_image: Mapping[str, list], gt_by_image: Mapping[str, list], w2i: Mapping[str, int],
                  iou_thresh: float = 0.5) -> Dict[str, object]:
    """PASCAL-VOC detection AP per object word and their mean (the reference's ``compute_att_metrics_using_maskrcnn_voc``:
    area under the monotone precision envelope, no 'difficult' boxes, +1 on the far box edges before matching, every
    ground-truth box matched at most once, detections of an image without prediction = one dummy box of class 0)."""
    preds = _image_table(pred_by_image, w2i)
    gts = _image_table(gt_by_image, w2i)
    dummy = (np.array([0]), np.array([0]), np.array([[0, 0, 0, 0]], dtype=np.float32))
    scores, matches = defaultdict(list), defaultdict(list)
    all_gt_labels = []
    for name, (g_lab, _g_sc, g_box) in gts.items():
        p_lab, p_sc, p_box = preds.get(name, dummy)
        all_gt_labels.append(g_lab)
        for l in np.unique(np.concatenate((p_lab, g_lab)).astype(int)):
            sel = p_lab == l
            order = p_sc[sel].argsort()[::-1]
            pb, ps = p_box[sel][order], p_sc[sel][order]
            gb = g_box[g_lab == l]
            scores[l].extend(ps)
            if len(pb) == 0:
                continue
            if len(gb) == 0:
                matches[l].extend((0,) * pb.shape[0])
                continue
            pb, gb = pb.copy(), gb.copy()
            pb[:, 2:] += 1
            gb[:, 2:] += 1
            iou = _pairwise_iou(pb, gb)
            who = iou.argmax(axis=1)
            who[iou.max(axis=1) < iou_thresh] = -1
            taken = np.zeros(gb.shape[0], dtype=bool)
            for j in who:
                if j >= 0:
                    matches[l].append(0 if taken[j] else 1)
                    taken[j] = True
                else:
                    matches[l].append(0)
    gt_all = np.concatenate(all_gt_labels)
    per_class, aps = {}, []
    idx2word = {i: w for w, i in w2i.items()}
    for l in np.unique(gt_all.astype(int)):
        n_pos = int(np.sum(gt_all.astype(int) == l))
        sc, mt = np.array(scores[l]), np.array(matches[l], dtype=np.int8)
        mt = mt[sc.argsort()[::-1]]
        tp, fp = np.cumsum(mt == 1), np.cumsum(mt == 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec = tp / (fp + tp)
        rec = tp / n_pos
        mpre = np.concatenate(([0], np.nan_to_num(prec), [0]))
        mrec = np.concatenate(([0], rec, [1]))
        mpre = np.maximum.accumulate(mpre[::-1])[::-1]
        step = np.where(mrec[1:] != mrec[:-1])[0]
        ap = float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))
        aps.append(ap)
        per_class[idx2word[int(l)]] = {"ap": ap, "class_id": int(l), "label": idx2word[int(l)], "precisions": prec.tolist(),
                                      "recalls": rec.tolist(), "n_tp": int(np.sum(mt == 1)), "n_fp": int(np.sum(mt == 0)),
                                      "n_positives": n_pos}
    return {"metrics_per_class": per_class, "overall_map": float(np.nanmean(aps)) if aps else float("nan")}


# ---------------------------------------------------------------------------------------------------------------
# files (eval/eval_tvqa_plus.py:148-210)
# ---------------------------------------------------------------------------------------------------------------
def clean_label(label: str) -> str:
    return label.replace(u"’", "'").replace(u"‘", "'").lower()


def _image_key(vid_name: str, qid, frame) -> str:
    return "{}_{}_{:05d}".format(vid_name, int(qid), int(frame))


def load_annotation(raw: Sequence[Mapping]) -> Dict[str, Mapping]:
    """tvqa_plus_{val,...}.json entries -> {"ts_answer": {qid: [ts, answer_idx]}, "bbox": {image: [[label, 1, xyxy]]}}."""
    boxes, spans = defaultdict(list), {}
    for e in raw:
        spans[e["qid"]] = [e["ts"], int(e["answer_idx"])]
        for frame, items in e["bbox"].items():
            key = _image_key(e["vid_name"], e["qid"], frame)
            for b in items:
                boxes[key].append([clean_label(b["label"]), 1,
                                   [b["left"], b["top"], b["left"] + b["width"], b["top"] + b["height"]]])
    return dict(ts_answer=spans, bbox=boxes)


def load_predictions(raw_preds: Mapping, gt_data: Sequence[Mapping], w2i: Mapping[str, int]) -> Dict[str, Mapping]:
    """A ``PredictionWriter`` file -> the evaluation layout: only the boxes predicted for the GROUND-TRUTH answer count, and
    of those only words that are annotated in that frame."""
    idx2word = {i: w for w, i in w2i.items()}
    answer = {int(e["qid"]): int(e["answer_idx"]) for e in gt_data}
    gt_boxes = {int(e["qid"]): e["bbox"] for e in gt_data}
    unk = w2i["<unk>"]
    out: Dict[str, list] = {}
    for per_q in raw_preds["raw_bbox"]:
        qid = None
        for a in range(NUM_ANSWERS):
            if len(per_q[str(a)]) > 0:
                qid = per_q[str(a)][0]["qid"]
        assert qid is not None
        for p in per_q[str(answer[int(qid)])]:
            annotated = [w2i.get(clean_label(b["label"]), unk) for b in gt_boxes[int(qid)][str(p["img_idx"])]]
            key = _image_key(p["vid_name"], qid, p["img_idx"])
            rows = out.setdefault(key, [])
            if p["word"] in annotated:
                rows.extend([idx2word[p["word"]], float(p["pred"][i]), b] for i, b in enumerate(p["bbox"]))
    return dict(ts_answer=raw_preds["ts_answer"], bbox=out)


def evaluate(prediction: Mapping, groundtruth: Mapping, w2i: Mapping[str, int]) -> Dict[str, object]:
    res = dict(detection_map(prediction["bbox"], groundtruth["bbox"], w2i))
    res.update(compute_temporal_metrics(prediction["ts_answer"], groundtruth["ts_answer"]))
    return res


def evaluate_files(pred_path: str, gt_path: str, w2i_path: str, preprocessed: bool = False) -> Dict[str, object]:
    with open(gt_path) as f:
        gt_raw = json.load(f)
    with open(w2i_path) as f:
        w2i = json.load(f)
    with open(pred_path) as f:
        raw = json.load(f)
    prediction = raw if preprocessed else load_predictions(raw, gt_raw, w2i)
    return evaluate(prediction, load_annotation(gt_raw), w2i)


def main(argv: Optional[List[str]] = None) -> None:
    import argparse
    ap = argparse.ArgumentParser(description="TVQA+ metrics: QA Acc., Grd. mAP, Temp. mIoU, ASA")
    ap.add_argument("--gt_path", default="data/tvqa_plus_val.json")
    ap.add_argument("--pred_path", required=True)
    ap.add_argument("--word2idx_path", default="data/word2idx.json")
    ap.add_argument("--output_path")
    ap.add_argument("--no_preproc_pred", action="store_true")
    args = ap.parse_args(argv)
    m = evaluate_files(args.pred_path, args.gt_path, args.word2idx_path, args.no_preproc_pred)
    print("QA Acc. {}\nGrd. mAP {}\nTemp. mIoU{}\nASA {}".format(m["qa_acc"], m["overall_map"], m["miou"],
                                                               m["ans_span_joint_acc@.5"]))
    if args.output_path:
        with open(args.output_path, "w") as f:
            f.write(json.dumps(m, indent=4, sort_keys=True))


if __name__ == "__main__":
    main()
