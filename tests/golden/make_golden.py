#!/usr/bin/env python
"""Generate golden vectors by IMPORTING the reference (read-only) in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference Python never travels to the GPU box; these fixtures (inputs + expected outputs, data only)
do.  Each .npz holds: ``opt`` (JSON), ``param/<state_dict key>``, ``in/<batch key>``, ``out/<name>`` and, for
train cases, ``grad/<state_dict key>`` of  CE_sum*N/N_new + 0.5*temporal_loss  (main.py:55-60 without the
att-loss term).  All parameters are perturbed away from their default init (LN affine != identity) so every
term is exercised.  Dropout: opt.dropout=0 for train cases and the MultiHeadedAttention p=0.1 dropout module
is set to p=0 in memory (the reference files are not edited).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("TVQA_GOLDEN_OUT", HERE)      # where the fixtures are written (tests/test_oracle_golden.py regenerates into a scratch directory)
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TVQA_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "model"))  # model/stage.py uses py2 implicit-relative imports
sys.path.insert(0, REF)

from model.stage import STAGE as RefSTAGE  # noqa: E402
from model.context_query_attention import StructuredAttention as RefSA  # noqa: E402
from model.encoder import StackedEncoder as RefEnc  # noqa: E402

from tvqaplus_amd.synth import make_batch, make_opt  # noqa: E402


def perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.add_(0.15 * torch.randn(p.shape, generator=g) * (p.abs().mean() + 0.1))


def np32(t):
    return t.detach().cpu().numpy()


def run_case(name, opt_kw, batch_kw, mode, seed):
    torch.manual_seed(seed)
    opt = make_opt(**opt_kw)
    model = RefSTAGE(opt)
    perturb(model, seed + 1)
    for m in model.modules():  # fixed p=0.1 attention dropout -> 0 so train-mode outputs are deterministic
        if isinstance(m, torch.nn.Dropout) and mode == "train":
            m.p = 0.0
    batch = make_batch(seed=seed + 2, wd_size=opt.embedding_size, vfeat_size=opt.vfeat_size, **batch_kw)
    rec = {"opt": np.array(json.dumps(vars(opt))), "mode": np.array(mode),
           "batch_kw": np.array(json.dumps(batch_kw))}
    for k, v in model.state_dict().items():
        rec["param/" + k] = np32(v)
    for k in ("qas_bert", "qas_mask", "sub_bert", "sub_mask", "vid", "vid_mask", "target", "ts_label_mask"):
        rec["in/" + k] = np32(batch[k])
    rec["in/ts_label_st"], rec["in/ts_label_ed"] = np32(batch.ts_label["st"]), np32(batch.ts_label["ed"])

    if batch.att_labels is not None:
        rec["in/att_labels"] = np.stack([np.stack([np32(l) for l in per]) for per in batch.att_labels])
    if mode == "train":
        model.train()
        torch.manual_seed(seed + 7)   # the negative sampling of get_att_loss draws from the global generator
        rec["att_seed"] = np.array(seed + 7)
        (out, targets), att_loss, _, t_loss, t_scores, other = model.forward_main(batch)
        loss = torch.nn.functional.cross_entropy(out, targets, reduction="sum") * (len(batch.qid) / len(targets)) \
            + 0.5 * t_loss
        if opt.use_sup_att:           # main.py:57-60 with att_weight = 0.1
            loss = loss + 0.1 * att_loss
            rec["out/att_loss"] = np32(att_loss)
        loss.backward()
        rec["out/logits"], rec["out/targets"] = np32(out), np32(targets)
        rec["out/temporal_loss"], rec["out/t_scores"], rec["out/loss"] = np32(t_loss), np32(t_scores), np32(loss)
        for k, p in model.named_parameters():
            rec["grad/" + k] = np32(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    elif mode == "eval":
        model.eval()
        with torch.no_grad():
            out, att_loss, _, t_loss, t_prob, other = model.forward_main(batch)
        rec["out/logits"], rec["out/temporal_loss"], rec["out/t_prob"] = np32(out), np32(t_loss), np32(t_prob)
        rec["out/t_scores"] = np32(other["temporal_scores"])
    elif mode == "inference":
        model.eval()
        model.inference_mode = True
        with torch.no_grad():
            res = model(batch)
        rec["out/logits"], rec["out/t_prob"] = np32(res["answer"]), np32(res["t_scores"])
        other = {}
    for k in ("sub_raw_s", "sub_normalized_s", "vid_raw_s", "vid_normalized_s"):
        if k in other:
            rec["out/" + k] = np32(other[k])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def k1_case(name, N, Li, Lr, Lqa, D, seed, scale=10.0):
    """StructuredAttention alone (model/context_query_attention.py:35-101), eval mode, with gradients."""
    g = torch.Generator().manual_seed(seed)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g).requires_grad_()
    Q = torch.randn(N, 1, Li, Lr, D, generator=g).requires_grad_()
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=seed, empty_frames=True)
    c_mask, q_mask = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    with torch.no_grad():  # zero-norm rows exercise the 1e-12 clamp of F.normalize
        C[0, 1, 0, 0].zero_()
        Q[0, 0, 0, 1].zero_()
    sa = RefSA(dropout=0.1, scale=scale).eval()
    A, S, S_mask, S_ = sa(C, Q, c_mask, q_mask)
    gA = torch.randn(A.shape, generator=g)
    gS = torch.randn(S.shape, generator=g) * 0.1
    gSn = torch.randn(S_.shape, generator=g) * 0.1
    ((A * gA).sum() + (S * gS).sum() + (S_ * gSn).sum()).backward()
    rec = dict(C=np32(C), Q=np32(Q), c_mask=np32(c_mask), q_mask=np32(q_mask), scale=np.float32(scale),
               A=np32(A), S=np32(S), S_mask=np32(S_mask), S_norm=np32(S_), gA=np32(gA), gS=np32(gS), gSn=np32(gSn),
               dC=np32(C.grad), dQ=np32(Q.grad))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def encoder_case(name, M, L, D, k, n_conv, nh, seed):
    """StackedEncoder alone (model/encoder.py:66-74) incl. the MHA query-row mask quirk, eval, with grads."""
    torch.manual_seed(seed)
    enc = RefEnc(n_blocks=1, n_conv=n_conv, kernel_size=k, hidden_size=D, dropout=0.1, num_heads=nh).eval()
    perturb(enc, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(M, L, D, generator=g).requires_grad_()
    lens = torch.randint(0, L + 1, (M,), generator=g)
    lens[0] = L
    mask = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
    y = enc(x, mask)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    rec = dict(x=np32(x), mask=np32(mask), y=np32(y), gy=np32(gy), dx=np32(x.grad),
               cfg=np.array(json.dumps(dict(k=k, n_conv=n_conv, nh=nh))))
    for kk, v in enc.state_dict().items():
        rec["param/" + kk] = np32(v)
    for kk, p in enc.named_parameters():
        rec["grad/" + kk] = np32(p.grad)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


TINY = dict(N=2, Li=2, Lr=4, Lw=5, Lqa=6)          # BASELINE.json config 1 (D=16)
SMALL = dict(N=3, Li=7, Lr=5, Lw=9, Lqa=8, empty_frames=True)
# Train (gradient) cases avoid frames whose regions are all masked while the frame itself is valid: such a frame turns
# into an all -1e10 row entering a LayerNorm, where the reference's own fp32 backward is rounding noise (torch's CPU
# kernel returns dx = 0 and a 1e5-scale dgamma for a constant row) -- there is nothing well defined to match.
SMALL_T = dict(N=3, Li=7, Lr=5, Lw=9, Lqa=8)
MID = dict(N=2, Li=6, Lr=20, Lw=50, Lqa=40)        # full per-frame shapes, D=128


def att_case(name, seed, loss_type, hard, pool, num_hard, drop_topk, num_negatives=2, start=0):
    """get_att_loss (model/stage.py:612-746, training mode) and get_att_prediction (:748-806) on a synthetic score
    tensor: the host index building + negative sampling are what is pinned here (same torch seed => same draws)."""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(seed)
    N, Li, Lqa, Lr, n_img = 3, 6, 7, 16, 3
    scores = (torch.rand(N, 5, Li, Lqa, Lr, generator=g) * 2 - 1).requires_grad_()
    target = torch.randint(0, 5, (N,), generator=g)
    labels = []
    for b in range(N):
        per_img = []
        for i in range(n_img):
            lab = (torch.rand(Lqa, Lr, generator=g) < 0.1).float()
            if b == 1 and i == 0:
                lab.zero_()                                  # an annotated image without any positive
            lab[:, 0] = 0                                    # every word keeps at least one negative region
            per_img.append(lab)
        labels.append(per_img)
    words = torch.randint(0, 12, (N, 5, Lqa), generator=g)
    boxes = [[[[int(x) for x in torch.randint(0, 100, (4,), generator=g)] for _ in range(Lr)] for _ in range(n_img)]
             for _ in range(N)]
    self = SimpleNamespace(negative_pool_size=pool, num_hard=num_hard, att_loss_type=loss_type, margin=0.1, alpha=20.0,
                           training=True, vfeat_flag=True, sample_negatives=RefSTAGE.sample_negatives)
    torch.manual_seed(seed + 1)
    loss, _ = RefSTAGE.get_att_loss(self, scores, labels, target, words, ["v%d" % b for b in range(N)], list(range(N)),
                                    [3] * N, [list(range(Li))] * N, boxes, [start] * N, num_negatives=num_negatives,
                                    use_hard_negatives=hard, drop_topk=drop_topk)
    loss.backward()
    vocab = [1, 4, 7, 9]
    preds = RefSTAGE.get_att_prediction(self, scores.detach(), vocab, words, ["v%d" % b for b in range(N)],
                                        list(range(N)), [list(range(100, 100 + Li))] * N, boxes, [start] * N)
    rec = {"scores": np32(scores), "target": np32(target), "labels": np.stack([np.stack([np32(l) for l in per]) for per in labels]),
           "words": np32(words), "boxes": np.array(boxes), "vocab": np.array(vocab),
           "cfg": np.array(json.dumps(dict(seed=seed + 1, loss_type=loss_type, hard=hard, pool=pool, num_hard=num_hard,
                                           drop_topk=drop_topk, num_negatives=num_negatives, start=start, Li=Li))),
           "loss": np32(loss), "grad": np32(scores.grad), "preds": np.array(json.dumps(preds))}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print("%-24s loss %.6f  preds %d" % (name, float(loss), sum(len(v) for q in preds for v in q.values())))


def main():
    run_case("tiny_eval", dict(hsz=16, embedding_size=64), TINY, "eval", 11)
    run_case("tiny_inference", dict(hsz=16, embedding_size=64), TINY, "inference", 11)
    run_case("tiny_train", dict(hsz=16, dropout=0.0), TINY, "train", 12)
    run_case("tiny_train_local", dict(hsz=16, embedding_size=64, dropout=0.0, add_local=True), TINY, "train", 13)
    run_case("small_local_eval", dict(hsz=32, embedding_size=48, vfeat_size=40, add_local=True), SMALL, "eval", 21)
    run_case("small_local_train", dict(hsz=32, embedding_size=48, vfeat_size=40, add_local=True, dropout=0.0),
             SMALL_T, "train", 22)
    run_case("small_heads_train", dict(hsz=32, embedding_size=48, vfeat_size=40, add_local=True, dropout=0.0,
                                       input_encoder_n_heads=2, cls_encoder_n_heads=4, t_iter=1), SMALL_T, "train", 23)
    run_case("small_heads_eval", dict(hsz=32, embedding_size=48, vfeat_size=40, input_encoder_n_heads=2,
                                      cls_encoder_n_heads=4, t_iter=1), SMALL, "eval", 24)
    run_case("small_subonly_train", dict(hsz=32, embedding_size=48, vfeat_flag=False, dropout=0.0), SMALL_T, "train", 25)
    run_case("small_vidonly_train", dict(hsz=32, embedding_size=48, vfeat_size=40, sub_flag=False, dropout=0.0,
                                         add_local=True), SMALL_T, "train", 26)
    run_case("mid_train", dict(hsz=128, embedding_size=96, vfeat_size=64, dropout=0.0, add_local=True), MID, "train", 31)
    run_case("mid_eval", dict(hsz=128, embedding_size=96, vfeat_size=64, add_local=True), MID, "eval", 32)
    k1_case("k1_small", N=2, Li=3, Lr=5, Lqa=7, D=16, seed=41)
    k1_case("k1_mid", N=2, Li=5, Lr=20, Lqa=40, D=128, seed=42)
    k1_case("k1_sub", N=1, Li=3, Lr=50, Lqa=40, D=128, seed=43)
    encoder_case("enc_k7", M=5, L=20, D=32, k=7, n_conv=2, nh=0, seed=51)
    encoder_case("enc_k5_heads", M=6, L=11, D=32, k=5, n_conv=2, nh=4, seed=52)


def emptyframe_main():
    """A VALID frame whose regions are all masked, in TRAINING mode (the other train fixtures avoid it).  The forward of
    that frame is well defined (eval fixtures hold it); its backward sends a constant -1e10 row through three LayerNorms,
    where the reference's own CPU kernel returns rounding noise for the affine weight gradients (xhat = 0 exactly;
    tests/test_oracle_golden.py lists the three parameters that are therefore not compared)."""
    run_case("small_emptyframe_train", dict(hsz=32, embedding_size=48, vfeat_size=40, add_local=True, dropout=0.0),
             dict(N=3, Li=7, Lr=5, Lw=9, Lqa=8, empty_frames=True), "train", 27)


def att_model_main():
    run_case("small_supatt_train", dict(hsz=32, dropout=0.0, use_sup_att=True, att_loss_type="lse", embedding_size=64, vfeat_size=48),
             dict(N=3, Li=5, Lr=12, Lw=6, Lqa=8, att_imgs=2, ragged=False), "train", 41)   # dense masks: a masked (-1e10) positive makes lse inf


def att_main():
    att_case("att_lse_random", 31, "lse", False, 0, 2, 0)
    att_case("att_hinge_hard", 32, "hinge", True, 0, 2, 1)
    att_case("att_lse_pool_mix", 33, "lse", True, 3, 1, 0)
    att_case("att_hinge_pool", 34, "hinge", True, 4, 2, 1, num_negatives=3)


def write_manifest():
    """MANIFEST.json: the key set of every fixture as THIS generator writes it (tests/test_oracle_golden.py holds the committed files to it)."""
    import glob
    man = {os.path.basename(f)[:-4]: sorted(np.load(f, allow_pickle=False).files) for f in sorted(glob.glob(os.path.join(OUT, "*.npz")))}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "att":   # only the attention-loss / box-prediction fixtures
        att_main()
        att_model_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "emptyframe":
        emptyframe_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "only":   # selected whole-model cases (tests: bit-exact regeneration check)
        cases = set(sys.argv[2:])
        _run = run_case
        run_case = lambda name, *a, **k: _run(name, *a, **k) if name in cases else None   # noqa: E731
        main_names = [n for n in cases]
        k1c, encc = k1_case, encoder_case
        k1_case = lambda name, **k: k1c(name, **k) if name in cases else None              # noqa: E731
        encoder_case = lambda name, **k: encc(name, **k) if name in cases else None        # noqa: E731
        main()
    else:
        main()
        att_main()
        att_model_main()
        emptyframe_main()
        write_manifest()
