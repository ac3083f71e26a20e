"""CPU: span decoding, prediction file and TVQA+ metrics (tvqaplus_amd/evaluation.py) against fixtures produced by RUNNING
the reference's own code (tests/golden/make_golden_eval.py: eval/eval_tvqa_plus.py, eval/maskrcnn_voc, inference.py's
find_max_pair).  No kernel involved."""
import json
import os

import numpy as np
import pytest
import torch

from tvqaplus_amd import evaluation as E

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_find_max_pair_matches_reference_host_and_batch():
    rows = _load("eval_find_max_pair.json")
    for r in rows:
        (st, ed), val = E.find_max_pair(r["p1"], r["p2"])
        assert (st, ed) == (r["st"], r["ed"]) and val == pytest.approx(r["val"], rel=1e-6, abs=1e-12)
    # the batched (device-capable) decoder: rows of equal length stacked
    by_len = {}
    for r in rows:
        by_len.setdefault(len(r["p1"]), []).append(r)
    for L, rs in by_len.items():
        p1 = torch.tensor([r["p1"] for r in rs], dtype=torch.float64)
        p2 = torch.tensor([r["p2"] for r in rs], dtype=torch.float64)
        st, ed, val = E.find_max_pair_batch(p1, p2)
        assert st.tolist() == [r["st"] for r in rs], L
        assert ed.tolist() == [r["ed"] for r in rs], L
        np.testing.assert_allclose(val.numpy(), [r["val"] for r in rs], rtol=1e-12, atol=0)


@pytest.mark.parametrize("case", range(3))
def test_metrics_match_reference(case):
    c = _load("eval_metrics.json")[case]
    gt_raw, raw, w2i, exp = c["gt"], c["pred"], c["w2i"], c["expected"]
    pred = E.load_predictions(raw, gt_raw, w2i)
    assert sorted(pred["bbox"].keys()) == exp["pred_bbox_keys"]
    assert {k: len(v) for k, v in pred["bbox"].items()} == exp["n_pred_boxes"]
    ann = E.load_annotation(gt_raw)
    m = E.evaluate(pred, ann, w2i)
    assert m["overall_map"] == pytest.approx(exp["overall_map"], rel=1e-12, abs=1e-15)
    assert sorted(m["metrics_per_class"].keys()) == sorted(exp["per_class"].keys())
    for k, e in exp["per_class"].items():
        g = m["metrics_per_class"][k]
        assert g["ap"] == pytest.approx(e["ap"], rel=1e-12, abs=1e-15)
        assert (g["n_tp"], g["n_fp"], g["n_positives"]) == (e["n_tp"], e["n_fp"], e["n_positives"])
    for k, v in exp["temporal"].items():
        assert m[k] == pytest.approx(v, rel=1e-12, abs=1e-15), k


def test_perfect_prediction_scores_one(tmp_path):
    """Known answer: predicting the annotation itself gives QA Acc. = mIoU = ASA = mAP = 1 (through the file interface)."""
    c = _load("eval_metrics.json")[0]
    gt_raw, w2i = c["gt"], c["w2i"]
    ts_answer, raw_bbox = {}, []
    for e in gt_raw:
        ts_answer[str(e["qid"])] = [e["ts"], int(e["answer_idx"])]
        per_q = {str(a): [] for a in range(5)}
        for frame, items in e["bbox"].items():
            for b in items:
                w = w2i.get(E.clean_label(b["label"]), w2i["<unk>"])
                per_q[e["answer_idx"]].append({"pred": [0.9], "word": w, "qid": e["qid"], "vid_name": e["vid_name"],
                                               "img_idx": int(frame),
                                               "bbox": [[b["left"], b["top"], b["left"] + b["width"], b["top"] + b["height"]]]})
        if not per_q[e["answer_idx"]]:
            per_q[e["answer_idx"]].append({"pred": [], "bbox": [], "word": 1, "qid": e["qid"], "vid_name": e["vid_name"],
                                           "img_idx": int(next(iter(e["bbox"])))})
        raw_bbox.append(per_q)
    paths = {k: str(tmp_path / (k + ".json")) for k in ("gt", "pred", "w2i")}
    for k, v in (("gt", gt_raw), ("pred", dict(ts_answer=ts_answer, raw_bbox=raw_bbox)), ("w2i", w2i)):
        with open(paths[k], "w") as f:
            json.dump(v, f)
    m = E.evaluate_files(paths["pred"], paths["gt"], paths["w2i"])
    assert m["qa_acc"] == 1.0 and m["miou"] == 1.0 and m["ans_span_joint_acc@.5"] == 1.0
    assert m["overall_map"] == pytest.approx(1.0)


def test_prediction_writer_schema_and_time_mapping(tmp_path):
    """inference.py:56-72: answer = arg-max, span of the predicted answer, seconds = index * 2 + (first image index % 6) / 3."""
    N, Li = 3, 7
    g = torch.Generator().manual_seed(3)
    answer = torch.randn(N, 5, generator=g)
    t = torch.softmax(torch.randn(N, 5, Li, 2, generator=g) * 3, dim=2)
    img = [[4 + 3 * n + i for i in range(Li)] for n in range(N)]
    w = E.PredictionWriter()
    w.add_batch({"answer": answer, "t_scores": t, "att_predictions": [{"0": []}]}, [11, 12, 13], img)
    for n, qid in enumerate((11, 12, 13)):
        a = int(answer[n].argmax())
        (st, ed), _ = E.find_max_pair(t[n, a, :, 0].tolist(), t[n, a, :, 1].tolist())
        off = (img[n][0] % 6) / 3
        assert w.predictions["ts_answer"][str(qid)] == [[st * 2 + off, (ed + 1) * 2 + off], a]
    assert w.predictions["raw_bbox"] == [{"0": []}]
    p = str(tmp_path / "pred.json")
    w.save(p)
    assert json.load(open(p))["ts_answer"].keys() == {"11", "12", "13"}
