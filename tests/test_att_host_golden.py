"""CPU: the host-side consumers of the attention map (tvqaplus_amd/att_host.py) against
fixtures generated from the reference's get_att_loss / get_att_prediction (tests/golden/make_golden.py att).
Pure index building + torch ops on top of the score tensor: no HIP kernel involved, so this runs without a GPU."""
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import Fixture
from tvqaplus_amd import att_host

ATT_CASES = ["att_lse_random", "att_hinge_hard", "att_lse_pool_mix", "att_hinge_pool"]


def _load(name):
    fx = Fixture(name)
    cfg = json.loads(str(fx["cfg"]))
    scores = torch.from_numpy(fx["scores"]).requires_grad_()
    labels = [[torch.from_numpy(l) for l in per] for per in fx["labels"]]
    N = scores.shape[0]
    batch = SimpleNamespace(target=torch.from_numpy(fx["target"]), att_labels=labels, anno_st_idx=[cfg["start"]] * N,
                            use_hard_negatives=cfg["hard"])
    model = SimpleNamespace(num_negatives=cfg["num_negatives"], negative_pool_size=cfg["pool"], num_hard=cfg["num_hard"],
                            drop_topk=cfg["drop_topk"], att_loss_type=cfg["loss_type"], margin=0.1, alpha=20.0)
    return fx, cfg, scores, batch, model


@pytest.mark.parametrize("name", ATT_CASES)
def test_att_loss_matches_reference(name):
    fx, cfg, scores, batch, model = _load(name)
    torch.manual_seed(cfg["seed"])                      # same generator state => same negative draws as the reference
    loss, preds = att_host.get_att_loss(model, scores, batch)
    assert preds is None
    np.testing.assert_allclose(float(loss.detach()), float(fx["loss"]), rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(scores.grad.numpy(), fx["grad"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ATT_CASES[:2])
def test_att_prediction_matches_reference(name):
    fx, cfg, scores, batch, model = _load(name)
    N, Li = scores.shape[0], cfg["Li"]
    words = torch.from_numpy(fx["words"])
    boxes = fx["boxes"].tolist()
    got = att_host.get_att_prediction(scores.detach(), fx["vocab"].tolist(), words, ["v%d" % b for b in range(N)],
                                      list(range(N)), [list(range(100, 100 + Li))] * N, boxes, [cfg["start"]] * N)
    exp = json.loads(str(fx["preds"]))
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert sorted(g.keys()) == sorted(int(k) for k in e.keys())
        for a in g:
            ge, ee = g[a], e[str(a)]
            assert len(ge) == len(ee)
            for dg, de in zip(ge, ee):
                assert dg["word"] == de["word"] and dg["qid"] == de["qid"] and dg["vid_name"] == de["vid_name"]
                assert dg["img_idx"] == de["img_idx"] and dg["bbox"] == de["bbox"]
                np.testing.assert_allclose(dg["pred"], de["pred"], rtol=0, atol=0)


def test_unknown_loss_type_raises():
    fx, cfg, scores, batch, model = _load(ATT_CASES[0])
    model.att_loss_type = "bce"
    with pytest.raises(NotImplementedError):
        att_host.get_att_loss(model, scores, batch)


def test_placeholder_targets_plus_device_offset_equal_the_host_build():
    """``build_att_pairs(placeholder_targets=True)`` + ``AttPairs(target_dev=...)`` (the answer's slice offset added next to the score
    tensor instead of ``batch.target.tolist()``) index exactly the pairs the host-target build indexes, with the same random draws."""
    fx, cfg, scores, batch, model = _load("att_lse_random")
    shape = tuple(scores.shape)
    torch.manual_seed(cfg["seed"])
    pos, neg = att_host.build_att_pairs(model, batch, None)
    torch.manual_seed(cfg["seed"])
    pos0, neg0 = att_host.build_att_pairs(model, batch, None, placeholder_targets=True)
    assert (pos0[:, 1] == 0).all() and (neg0[:, 1] == 0).all()
    a = att_host.AttPairs(pos, neg, shape, "cpu")
    b = att_host.AttPairs(pos0, neg0, shape, "cpu", target_dev=batch.target)
    assert a.m == b.m and torch.equal(a.flat, b.flat)
    # on the host nothing qualifies for the device path (the decision needs a device-resident target and no host copy)
    assert not att_host.targets_on_device_ok(model, batch, None)
