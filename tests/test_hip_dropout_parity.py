"""-m gpu: a TRAINING step with dropout 0.1 -- the configuration bench.py times and run_main.sh trains -- against the reference
arithmetic, exactly (missing #4).

The reference's dropout draws from torch's generator; the HIP path's keep masks are a pure function of (seed, element index)
(csrc/common.h: mix64 / drop4 -- one SplitMix64 hash per 4 consecutive elements, 16 bits per element, keep <=> field >= round(p * 65536)).
So the masks of a HIP step can be REBUILT on the host from the seeds the model drew (the numpy restatement below, itself pinned
against the library's exported ``stage_dropout_keepmask``) and handed to the oracle (``oracle.stage_oracle.drop_masks``), which then
computes the reference's forward / backward with the same units dropped at every one of its dropout sites
(model/stage.py:85-138, model/encoder.py:41-44, model/context_query_attention.py:95-96, model/stage.py:469-482, :536).

Site order: ``stage_forward`` visits the sites in the order the model draws its seeds (statement branch, subtitle branch + attention,
video branch + attention, concat_fc, classifier encoder, temporal head, classifier).  On the ragged path a site's element index runs
over the COMPACT rows (tvqaplus_amd/ragged.py); the masks are scattered to the reference's dense rows through the layout's own row
maps, and the rows the ragged path never computes get RANDOM masks: they must not reach any output or gradient.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import stage_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3          # outputs, losses, attention maps (north star)
GTOL = 6e-3         # parameter gradients (tests/test_hip_stage.py: the reference's own fp32 gradients are off fp64 by 4.4e-3)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def thresh16(p: float) -> int:
    t = np.float32(p) * np.float32(65536.0) + np.float32(0.5)          # drop_thresh16 (csrc/common.h)
    return 0 if t <= 0 else (65535 if t >= 65535 else int(t))


def keep_bits(seed: int, p: float, n: int) -> np.ndarray:
    """keep flag of elements 0..n-1 of dropout stream ``seed`` (csrc/common.h: mix64 + drop4)."""
    with np.errstate(over="ignore"):
        idx = np.arange((n + 3) // 4, dtype=np.uint64)
        z = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        h = z ^ (z >> np.uint64(31))
    f = np.stack([(h >> np.uint64(16 * i)) & np.uint64(0xFFFF) for i in range(4)], axis=1).reshape(-1)[:n]
    return f >= np.uint64(thresh16(p))


def keep_mult(seed: int, p: float, n: int) -> torch.Tensor:
    inv_keep = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(keep_bits(seed, p, n).astype(np.float32) * inv_keep)


def test_host_restatement_of_the_dropout_stream_equals_the_library(hip_device):
    """the numpy hash above against ``stage_dropout_keepmask`` (include/stage_hip.h): bit for bit, several seeds / widths / rates."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    for seed, p, rows, K in ((12345, 0.1, 37, 128), (0x7FFFFFFFFFFFFFFF, 0.1, 5, 300), (987654321987, 0.35, 64, 768), (3, 0.5, 1, 4)):
        words = (K + 31) // 32
        mask = torch.zeros(words * rows, dtype=torch.int32, device=hip_device)
        _lib.check(lib.stage_dropout_keepmask(ctypes.c_float(p), ctypes.c_ulonglong(seed), mask.data_ptr(), rows, K,
                                              torch.cuda.current_stream().cuda_stream), "stage_dropout_keepmask")
        got = mask.cpu().numpy().view(np.uint32).reshape(words, rows)
        exp = keep_bits(seed, p, rows * K).reshape(rows, K)
        for r in range(rows):
            for w in range(words):
                bits = exp[r, 32 * w: 32 * w + 32]
                word = int(sum(int(b) << i for i, b in enumerate(bits)))
                assert int(got[w, r]) == word, (seed, p, r, w)


def _site_masks(model, seeds, p, batch, rng):
    """One mask BUILDER per dropout site, in the oracle's visiting order.  Each builder gets the oracle's tensor x at that site and
    returns the multipliers in x's (dense) layout."""
    lay, clays = model.last_ragged, model.last_ragged_ctx
    inv_keep = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    it = iter(seeds)

    def dense():
        s = next(it)
        return lambda x: keep_mult(s, p, x.numel())

    def scattered(rows_dense_of_compact):
        """site computed on compact rows: mask row c of the compact tensor belongs to dense row rows_dense_of_compact[c]; every other
        dense row is never computed by the product -- it gets an arbitrary mask."""
        s = next(it)
        idx = torch.from_numpy(np.asarray(rows_dense_of_compact, dtype=np.int64))

        def build(x):
            K = x.shape[-1]
            rows = x.numel() // K
            m = torch.from_numpy((rng.random((rows, K)) >= p).astype(np.float32) * inv_keep)
            m[idx] = keep_mult(s, p, idx.numel() * K).view(-1, K)
            return m
        return build

    def ctx(name):
        cl = clays.get(name)
        if cl is None:
            return dense()
        return scattered(cl.tab.src_rows_host())

    def stmt():
        if lay is None:
            return dense()
        ci = lay.tab.compact_index()                     # (U, 4): n, a, i, w
        t = lay.tab
        return scattered(((ci[:, 0] * t.NA + ci[:, 1]) * t.Li + ci[:, 2]) * t.Lqa + ci[:, 3])

    n_in = (model.input_encoder.stacked_encoderBlocks[0].n_conv + 1) // 2
    n_cls = (model.cls_encoder.stacked_encoderBlocks[0].n_conv + 1) // 2
    sites = [dense(), dense()] + [dense() for _ in range(n_in)]                      # statements: bridge LN, LN(300), encoder
    for name in ("sub", "vid"):
        sites += [ctx(name), ctx(name)] + [ctx(name) for _ in range(n_in)]            # context branch
        sites += [dense(), ctx(name), stmt()]                                         # attention: Cn, Qn, LN([a, b, a*b])
    sites += [stmt()]                                                                 # concat_fc
    sites += [stmt() for _ in range(n_cls)]                                           # classifier encoder
    sites += [dense(), dense(), dense()]                                              # temporal head: projection, start, end
    sites += [dense()]                                                                # classifier
    assert next(it, None) is None, "the model drew more seeds than the oracle has dropout sites"
    return sites


@pytest.mark.parametrize("path", ["ragged", "dense_rows", "per_op"])
def test_train_step_with_dropout_vs_oracle_with_exported_masks(hip_device, path):
    from tvqaplus_amd import att_host
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_batch, make_opt
    p = 0.1
    torch.manual_seed(21)
    opt = make_opt(hsz=128, embedding_size=64, vfeat_size=48, dropout=p, add_local=True, use_sup_att=True)
    model = STAGE(opt)
    with torch.no_grad():
        for q in model.parameters():
            q.add_(0.05 * torch.randn_like(q))
    batch = make_batch(N=3, Li=8, Lr=10, Lw=12, Lqa=14, wd_size=64, vfeat_size=48, seed=33, att_imgs=2, att_words=2)
    P = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in model.state_dict().items()}
    model = model.to(hip_device).train()
    model.use_ragged = path == "ragged"
    model.use_groups = path != "per_op"
    # record the dropout seeds in drawing order (a declined group rewinds the stream and the per-kernel path draws the same
    # seeds again: first occurrences, in order, are the sites)
    drawn, inner = [], model._seed

    def recording_seed():
        s = inner()
        if s not in drawn:
            drawn.append(s)
        return s
    model._seed = recording_seed
    torch.manual_seed(77)                 # the attention loss draws its negatives from torch's default generator
    (logits, targets), att_loss, _, t_loss, t_scores, other = model.forward_main(batch.to(hip_device))
    loss = F.cross_entropy(logits, targets, reduction="sum") * (3 / len(targets)) + 0.5 * t_loss + 0.1 * att_loss
    loss.backward()
    torch.cuda.synchronize()
    assert (model.last_ragged is not None) == (path == "ragged")
    if path == "ragged":
        assert set(model.last_ragged_ctx) == {"sub", "vid"}

    sites = _site_masks(model, drawn, p, batch, np.random.default_rng(5))
    torch.manual_seed(77)
    with O.drop_masks(sites):
        ref = O.stage_forward(P, opt, batch, training=True)
    ref_att = att_host.get_att_loss(opt, ref["vid_raw_s"], batch)[0]
    ref_loss = O.training_loss(ref, n_examples=3) + 0.1 * ref_att
    ref_loss.backward()

    assert torch.equal(targets.cpu(), ref["targets"]), "proposal set differs"
    assert rel_err(logits, ref["logits"]) < TOL
    assert rel_err(t_scores, ref["t_scores"]) < TOL
    assert rel_err(t_loss, ref["temporal_loss"]) < TOL
    assert rel_err(att_loss, ref_att) < TOL
    assert rel_err(loss, ref_loss) < TOL
    for k in ("sub_raw_s", "sub_normalized_s", "vid_raw_s", "vid_normalized_s"):
        got, exp = other[k], ref[k]
        assert rel_err(got.reshape(exp.shape), exp) < TOL, k
    errs = {k: rel_err(q.grad if q.grad is not None else torch.zeros_like(q), P[k].grad if P[k].grad is not None else torch.zeros_like(P[k]))
            for k, q in model.named_parameters()}
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] < GTOL, (worst, sorted((k, "%.1e" % e) for k, e in errs.items() if e >= GTOL))


def test_attention_pairs_with_the_answer_taken_on_the_device(hip_device):
    """att_host.targets_on_device_ok: a batch without ``target_list`` (what the reference's prepare_inputs delivers) must give the
    same attention loss and the same gradient into the score map as one with the host copy -- without ``batch.target.tolist()``."""
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_batch, make_opt
    torch.manual_seed(4)
    opt = make_opt(hsz=128, embedding_size=64, vfeat_size=48, dropout=0.0, add_local=True, use_sup_att=True)
    model = STAGE(opt).to(hip_device).train()
    res = []
    for strip in (False, True):
        batch = make_batch(N=3, Li=8, Lr=10, Lw=12, Lqa=14, wd_size=64, vfeat_size=48, seed=33, att_imgs=3, att_words=2).to(hip_device)
        if strip:
            batch.pop("target_list", None)
            batch.pop("mask_host", None)
            real = torch.Tensor.tolist

            def no_tolist(self):
                assert not self.is_cuda, "a device tensor was read on the host inside the step"
                return real(self)
            torch.Tensor.tolist = no_tolist
        try:
            model.zero_grad(set_to_none=True)
            torch.manual_seed(9)
            (logits, targets), att_loss, _, t_loss, _ = model(batch)
            (F.cross_entropy(logits, targets, reduction="sum") + 0.5 * t_loss + 0.1 * att_loss).backward()
        finally:
            if strip:
                torch.Tensor.tolist = real
        res.append((att_loss.detach().clone(), {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}))
    assert float(res[0][0]) != 0.0
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert rel_err(res[1][1][k], res[0][1][k]) < 1e-5, k
