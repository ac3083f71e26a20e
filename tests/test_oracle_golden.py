"""Pins the CPU oracle (oracle/stage_oracle.py) against golden vectors produced by the imported reference.

Tolerance: the oracle re-expresses the same fp32 torch arithmetic, so it must agree to rounding
(1e-5 relative-to-(1+|x|)); the north-star tolerance for the HIP path (1e-3) is applied in the -m gpu tests."""
import json

import pytest
import torch

from conftest import ENC_CASES, K1_CASES, MODEL_CASES, Fixture, rel_err
from oracle import stage_oracle as O

TOL = 1e-5


@pytest.mark.parametrize("name", MODEL_CASES)
def test_whole_model(name):
    fx = Fixture(name)
    P = fx.group("param")
    exp = fx.group("out")
    train = fx.mode == "train"
    if train:
        P = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith(".pe") else v)
             for k, v in P.items()}
    batch = fx.batch()
    opt = fx.opt
    opt.mha_dropout = 0.0  # fixtures were generated with the MHA dropout module's p set to 0
    with torch.set_grad_enabled(train):
        out = O.stage_forward(P, opt, batch, training=train)
    assert rel_err(out["logits"], exp["logits"]) < TOL
    if "targets" in exp:
        assert torch.equal(out["targets"], exp["targets"])
    if "t_scores" in exp:
        assert rel_err(out["t_scores"], exp["t_scores"]) < TOL
    if "t_prob" in exp:
        assert rel_err(torch.softmax(out["t_scores"], dim=2), exp["t_prob"]) < TOL
    if "temporal_loss" in exp:
        assert rel_err(out["temporal_loss"], exp["temporal_loss"]) < TOL
    for k in ("sub_raw_s", "sub_normalized_s", "vid_raw_s", "vid_normalized_s"):
        if k in exp:
            assert rel_err(out[k], exp[k]) < TOL, k
    if train:
        loss = O.training_loss(out, n_examples=batch.target.shape[0])
        if opt.use_sup_att:   # host logic on top of the attention map (pinned on its own in test_att_host_golden.py)
            from types import SimpleNamespace
            from tvqaplus_amd import att_host
            torch.manual_seed(int(fx["att_seed"]))
            m = SimpleNamespace(num_negatives=opt.num_negatives, negative_pool_size=opt.negative_pool_size,
                                num_hard=opt.num_hard, drop_topk=opt.drop_topk, att_loss_type=opt.att_loss_type,
                                margin=opt.margin, alpha=opt.alpha)
            att_loss, _ = att_host.get_att_loss(m, out["vid_raw_s"], batch)
            assert rel_err(att_loss, exp["att_loss"]) < TOL
            loss = loss + 0.1 * att_loss
        assert rel_err(loss, exp["loss"]) < TOL
        loss.backward()
        G = fx.group("grad")
        for k, g in G.items():
            got = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
            assert rel_err(got, g) < 2e-5, k


@pytest.mark.parametrize("name", K1_CASES)
def test_structured_attention(name):
    fx = Fixture(name)
    C = torch.from_numpy(fx["C"]).requires_grad_()
    Q = torch.from_numpy(fx["Q"]).requires_grad_()
    cm, qm = torch.from_numpy(fx["c_mask"]), torch.from_numpy(fx["q_mask"])
    A, S, S_mask, S_ = O.structured_attention(C, Q, cm, qm, float(fx["scale"]))
    assert rel_err(A, torch.from_numpy(fx["A"])) < TOL
    assert rel_err(S, torch.from_numpy(fx["S"])) < TOL
    assert torch.equal(S_mask, torch.from_numpy(fx["S_mask"]))
    assert rel_err(S_, torch.from_numpy(fx["S_norm"])) < TOL
    ((A * torch.from_numpy(fx["gA"])).sum() + (S * torch.from_numpy(fx["gS"])).sum()
     + (S_ * torch.from_numpy(fx["gSn"])).sum()).backward()
    assert rel_err(C.grad, torch.from_numpy(fx["dC"])) < 2e-5
    assert rel_err(Q.grad, torch.from_numpy(fx["dQ"])) < 2e-5


@pytest.mark.parametrize("name", ENC_CASES)
def test_encoder(name):
    fx = Fixture(name)
    cfg = json.loads(str(fx["cfg"]))
    P = {"enc." + k: v.requires_grad_(not k.endswith(".pe")) for k, v in fx.group("param").items()}
    x = torch.from_numpy(fx["x"]).requires_grad_()
    y = O.stacked_encoder(x, torch.from_numpy(fx["mask"]), P, "enc", 1, cfg["n_conv"], cfg["nh"], 0.1, False)
    assert rel_err(y, torch.from_numpy(fx["y"])) < TOL
    (y * torch.from_numpy(fx["gy"])).sum().backward()
    assert rel_err(x.grad, torch.from_numpy(fx["dx"])) < 2e-5
    for k, g in fx.group("grad").items():
        assert rel_err(P["enc." + k].grad, g) < 2e-5, k


def test_position_table_beyond_500():
    """The reference crashes for L > 500 (max_len); the oracle continues the same closed form."""
    tab = O.position_table({}, "none", 600, 32)
    ref = O.position_table({}, "none", 500, 32)
    assert torch.equal(tab[:500], ref)


def fp64_gradients(fx):
    """Parameter gradients of a train fixture's loss from the oracle evaluated in fp64 (the well-conditioned yardstick)."""
    opt = fx.opt
    opt.mha_dropout = 0.0
    P64 = {k: (v.double().requires_grad_(not k.endswith(".pe")) if v.is_floating_point() else v)
           for k, v in fx.group("param").items()}
    b = fx.batch()
    b64 = type(b)({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()})
    ref = O.stage_forward(P64, opt, b64, training=True)
    O.training_loss(ref, n_examples=b.target.shape[0]).backward()
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P64.items() if v.requires_grad}


def test_reference_fp32_gradients_are_off_fp64_by_more_than_the_output_tolerance():
    """Why parameter gradients are held to 6e-3 and not to the outputs' 1e-3 (tests/test_hip_stage.py: GTOL): the
    REFERENCE's own fp32 gradients of the mid fixture deviate from an fp64 evaluation of the same graph by several 1e-3 of
    (1 + |g|) -- LayerNorms over padded / near-constant rows amplify rounding by rstd ~ 316.  A tolerance below that spread
    would test the summation order of torch's CPU kernels, not the arithmetic.  The HIP path is held to 4e-3 against the SAME
    fp64 yardstick in tests/test_hip_stage.py::test_gradients_against_fp64."""
    fx = Fixture("mid_train")
    G, G64 = fx.group("grad"), fp64_gradients(fx)
    worst = max(rel_err(G[k].double(), G64[k]) for k in G64)
    assert 1.5e-3 < worst < 6e-3, worst


def test_committed_fixtures_have_the_generators_key_sets():
    """tests/golden/MANIFEST.json is written by make_golden.py next to the fixtures: every committed .npz must carry exactly the keys the
    current generator writes (a fixture that predates a generator change shows up here)."""
    import glob
    import json
    import os
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(here, "MANIFEST.json")))
    files = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(here, "*.npz")))
    assert files == sorted(man), set(files) ^ set(man)
    for name in files:
        assert sorted(np.load(os.path.join(here, name + ".npz"), allow_pickle=False).files) == man[name], name


def test_fixtures_regenerate_bit_exactly_from_the_reference(tmp_path):
    """Where the reference is importable (the build container: /root/reference), two fixtures are regenerated by the committed generator
    and compared array by array.  The reference never travels: elsewhere this is skipped."""
    import os
    import subprocess
    import sys
    import numpy as np
    ref = os.environ.get("TVQA_REFERENCE", "/root/reference")
    if not os.path.exists(os.path.join(ref, "model", "stage.py")):
        pytest.skip("reference not present")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    env = dict(os.environ, TVQA_GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(here, "make_golden.py"), "only", "tiny_train", "k1_small"], env=env,
                          stdout=subprocess.DEVNULL)
    for name in ("tiny_train", "k1_small"):
        a, b = np.load(os.path.join(str(tmp_path), name + ".npz")), np.load(os.path.join(here, name + ".npz"))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (name, k)
