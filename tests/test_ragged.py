"""CPU: the ragged-token-row tables (tvqaplus_amd/ragged.py) -- layout invariants, and the CLAIM they rest on, checked with the oracle's own
classifier encoder in fp64: running model/stage.py:502-503 (cls_encoder + mask_logits + max over the words) on the live rows only, with
zero padding behind Lc = min(Lqa, last valid word + 1 + halo) and -1e10 for dead frames, gives the same pooled outputs, the same
gradient on every live row, the same parameter gradients -- and the dense computation's gradient on every other row is exactly zero."""
import numpy as np
import pytest
import torch

from oracle import stage_oracle as O
from tvqaplus_amd import ragged


def _masks(rng, N, NA, Li, Lqa, holes=False):
    qa = np.zeros((N, NA, Lqa), dtype=bool)
    for n in range(N):
        for a in range(NA):
            qa[n, a, : rng.integers(0 if (n + a) % 7 == 3 else 1, Lqa + 1)] = True
    fl = np.zeros((N, Li), dtype=bool)
    for n in range(N):
        fl[n, : rng.integers(1, Li + 1)] = True
    if holes:
        qa[0, 0, 1] = False                      # a hole inside the valid words stays live
        fl[0, 0] = False                         # dead frames need not be trailing
        if N > 1:
            fl[1] = False                        # an example without a live frame
    return qa, fl


@pytest.mark.parametrize("holes", [False, True])
def test_tables_partition_the_live_rows(holes):
    rng = np.random.default_rng(5)
    N, NA, Li, Lqa, halo = 3, 5, 7, 12, 4
    qa, fl = _masks(rng, N, NA, Li, Lqa, holes)
    t = ragged.RaggedTables(qa, fl, halo)
    idx = t.compact_index()
    # every live (n, a, i, w) exactly once, in (n, a, i, w) order
    want = [(n, a, i, w) for n in range(N) for a in range(NA) for i in range(Li) if fl[n, i]
            for w in range(int(t.Lc[n * NA + a]))]
    assert [tuple(r) for r in idx.tolist()] == want
    assert t.U == len(want) and t.S == t.seq.shape[0]
    # Lc: last valid word + 1 + halo, clipped; 0 for a candidate without a valid word
    for g in range(N * NA):
        v = np.nonzero(qa[g // NA, g % NA])[0]
        assert int(t.Lc[g]) == (0 if v.size == 0 else min(Lqa, int(v[-1]) + 1 + halo))
    # frame-compact rows: distinct, inside the tensor, never in a dump slot
    ri = t.rowinfo_host()
    slots, first = t.fmap[N * Li: N * Li + N], t.fmap[N * Li + N:]
    dump = {int(first[n] + a * slots[n] + slots[n] - 1) for n in range(N) for a in range(NA)}
    fc_seq = ri[:, 1] // Lqa
    assert len(set(ri[:, 1].tolist())) == t.U and ri[:, 1].max(initial=-1) < t.Fc
    assert not (set(fc_seq.tolist()) & dump) and len(dump) == N * NA
    assert t.Fc == NA * int(slots.sum()) * Lqa
    # rowinfo agrees with the index: QA row, dense output row, word
    n_, a_, i_, w_ = idx.T
    assert np.array_equal(ri[:, 0], (n_ * NA + a_) * Lqa + w_)
    assert np.array_equal(ri[:, 2], (n_ * NA + a_) * Li + i_)
    assert np.array_equal(ri[:, 3], w_)
    # the frame map: slot of a live frame = its rank among the live frames of its example; dead frames negative
    fm = t.fmap[: N * Li].reshape(N, Li)
    for n in range(N):
        assert fm[n][fl[n]].tolist() == list(range(int(fl[n].sum()))) and (fm[n][~fl[n]] < 0).all()
    # frame-compact row of (n, a, live frame slot s, w)
    assert np.array_equal(ri[:, 1], (first[n_] + a_ * slots[n_] + fm[n_, i_]) * Lqa + w_)


def test_layout_uploads_and_aligns():
    rng = np.random.default_rng(7)
    qa, fl = _masks(rng, 2, 5, 4, 9)
    t = ragged.RaggedTables(qa, fl, 4)
    lay = ragged.RaggedLayout(t, "cpu")
    assert lay.Ucap >= lay.U and lay.Ucap % ragged.CAP_STEP == 0
    assert np.array_equal(lay.fmap.numpy(), t.fmap) and np.array_equal(lay.gdesc.numpy().reshape(-1, 4), t.gdesc)
    assert np.array_equal(lay.seq.numpy().reshape(-1, 4), t.seq) and np.array_equal(lay.seqfc.numpy(), t.seqfc)
    for v in (lay.gdesc, lay.seq, lay.seqfc):           # int4 loads on the device
        assert (v.data_ptr() - lay.tables.data_ptr()) % 16 == 0


def test_host_masks_from_batch_and_device():
    from tvqaplus_amd.synth import make_batch
    b = make_batch(N=2, Li=5, Lr=4, Lw=6, Lqa=7, wd_size=8, vfeat_size=8, seed=3, empty_frames=True)
    qa, fl = ragged.host_masks(b, "vid")
    qa2, fl2 = ragged.masks_from_device(b.qas_mask, b.vid_mask)
    assert np.array_equal(qa, qa2) and np.array_equal(fl, fl2)
    assert not np.array_equal(fl, ragged.host_masks(b, "sub")[1])      # the blanked video frame is live in the subtitle stream


@pytest.mark.parametrize("k,n_conv", [(5, 2), (3, 3), (7, 1)])
def test_live_rows_are_all_the_reference_needs(k, n_conv):
    """fp64, the oracle's encoder: dense (reference semantics) against the live-row computation."""
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    N, NA, Li, Lqa, D = 2, 3, 4, 14, 8
    qa, fl = _masks(rng, N, NA, Li, Lqa, holes=True)
    halo = ragged.conv_halo(1, n_conv, k)
    t = ragged.RaggedTables(qa, fl, halo)
    key = "cls_encoder.stacked_encoderBlocks.0"
    # parameters of one encoder block (names as in the reference's state_dict)
    from tvqaplus_amd.stage import _PositionTable
    P = {key + ".position_encoding.pe": _PositionTable.table(500, D).double()}
    for i in range(n_conv):
        P[f"{key}.layer_norm.{i}.weight"] = (1 + 0.1 * torch.randn(D)).double()
        P[f"{key}.layer_norm.{i}.bias"] = (0.1 * torch.randn(D)).double()
        P[f"{key}.conv.{i}.depthwise_conv.weight"] = (0.5 * torch.randn(D, 1, k)).double()
        P[f"{key}.conv.{i}.depthwise_conv.bias"] = (0.1 * torch.randn(D)).double()
        P[f"{key}.conv.{i}.pointwise_conv.weight"] = (0.4 * torch.randn(D, D, 1)).double()
        P[f"{key}.conv.{i}.pointwise_conv.bias"] = (0.1 * torch.randn(D)).double()
    P[key + ".final_layer_norm.weight"] = (1 + 0.1 * torch.randn(D)).double()
    P[key + ".final_layer_norm.bias"] = (0.1 * torch.randn(D)).double()
    names = [n for n in P if not n.endswith(".pe")]

    stmt = torch.randn(N, NA, Li, Lqa, D, dtype=torch.float64)
    mask = torch.from_numpy(qa[:, :, None, :] & fl[:, None, :, None]).double()           # model/stage.py:386
    w_out = torch.randn(N, NA, Li, D, dtype=torch.float64)                                # a fixed functional of the pooled output

    def run_dense():
        x = stmt.clone().requires_grad_(True)
        Pd = {n: (v.clone().requires_grad_(True) if n in names else v) for n, v in P.items()}
        y = O.encoder_block(x.view(N * NA * Li, Lqa, D), mask.view(-1, Lqa), Pd, key, n_conv, 0, 0.0, False)
        mx = O.mask_logits(y, mask.view(-1, Lqa, 1)).max(dim=1)[0].view(N, NA, Li, D)
        (mx * w_out).sum().backward()
        return mx.detach(), x.grad, {n: Pd[n].grad for n in names}

    def run_live():
        x = stmt.clone().requires_grad_(True)
        Pd = {n: (v.clone().requires_grad_(True) if n in names else v) for n, v in P.items()}
        mx = torch.full((N, NA, Li, D), -1e10, dtype=torch.float64)
        pieces = []
        for s in range(t.S):
            start, ln, g, dense = (int(v) for v in t.seq[s])
            n, a, i = g // NA, g % NA, dense - g * Li
            y = O.encoder_block(x[n, a, i, :ln].unsqueeze(0), mask[n, a, i, :ln].unsqueeze(0), Pd, key, n_conv, 0, 0.0, False)
            pieces.append((n, a, i, O.mask_logits(y, mask[n, a, i, :ln].view(1, ln, 1)).max(dim=1)[0][0]))
        tot = sum((p * w_out[n, a, i]).sum() for n, a, i, p in pieces)
        tot.backward()
        for n, a, i, p in pieces:
            mx[n, a, i] = p.detach()
        return mx, x.grad, {n: Pd[n].grad for n in names}

    mx_d, gx_d, gp_d = run_dense()
    mx_l, gx_l, gp_l = run_live()
    assert torch.equal(mx_d == -1e10, mx_l == -1e10)
    assert torch.allclose(mx_d, mx_l, rtol=1e-12, atol=1e-12)
    live = torch.zeros(N, NA, Li, Lqa, dtype=torch.bool)
    for n, a, i, w in t.compact_index().tolist():
        live[n, a, i, w] = True
    assert float(gx_d[~live].abs().max()) == 0.0            # the reference's own gradient is exactly zero off the live rows
    assert torch.allclose(gx_d, gx_l, rtol=1e-10, atol=1e-12)
    for n in names:
        assert torch.allclose(gp_d[n], gp_l[n], rtol=1e-9, atol=1e-11), n
    # and the halo is tight: one word less is not enough whenever a padded word exists behind the valid ones
    if (t.Lc < Lqa).any() and halo > 0:
        t2 = ragged.RaggedTables(qa, fl, halo - 1)
        assert t2.U < t.U


def test_context_tables():
    rng = np.random.default_rng(3)
    N, Li, L, halo = 3, 6, 11, 4
    lens = rng.integers(0, L + 1, size=(N, Li))
    lens[0, 0], lens[1, 2] = L, 0
    t = ragged.CtxTables(lens, L, halo)
    qlen = np.where(lens > 0, np.minimum(L, lens + halo), 0).reshape(-1)
    assert np.array_equal(t.cq[:, 1], qlen) and t.U == int(qlen.sum()) and t.S == int((qlen > 0).sum())
    src = t.src_rows_host()
    want = [f * L + w for f in range(N * Li) for w in range(int(qlen[f]))]
    assert src.tolist() == want
    # frames in order, back to back; a dead frame points at a valid row (0) with no rows of its own
    live = qlen > 0
    assert np.array_equal(t.cq[live, 0], np.concatenate([[0], np.cumsum(qlen[live])[:-1]]))
    assert (t.cq[~live] == 0).all()
    assert np.array_equal(t.seq[:, 0], t.cq[live, 0]) and np.array_equal(t.seq[:, 1], qlen[live])
    lay = ragged.CtxLayout(t, "cpu")
    assert np.array_equal(lay.cq.numpy().reshape(-1, 2), t.cq) and (lay.seq.data_ptr() - lay.tables.data_ptr()) % 16 == 0
    assert np.array_equal(ragged.mask_lens(np.arange(L)[None, :] < lens.reshape(-1, 1)), lens.reshape(-1))


@pytest.mark.parametrize("k,n_conv", [(7, 2), (3, 1)])
def test_context_rows_are_all_the_attention_needs(k, n_conv):
    """fp64, the oracle's encoder over a context stream: the valid positions of every frame -- the only ones the attention reads
    (model/context_query_attention.py:58-61: everything else is masked) -- and all gradients are the same whether the encoder sees the
    padded frame or its first len + halo rows; the padded computation's own input gradient is exactly zero behind them."""
    torch.manual_seed(4)
    rng = np.random.default_rng(4)
    F_, L, D = 7, 16, 8
    lens = rng.integers(0, L + 1, size=(1, F_))
    lens[0, 0], lens[0, 1] = L, 0
    halo = ragged.conv_halo(1, n_conv, k)
    t = ragged.CtxTables(lens, L, halo)
    key = "input_encoder.stacked_encoderBlocks.0"
    from tvqaplus_amd.stage import _PositionTable
    P = {key + ".position_encoding.pe": _PositionTable.table(500, D).double()}
    for i in range(n_conv):
        P[f"{key}.layer_norm.{i}.weight"] = (1 + 0.1 * torch.randn(D)).double()
        P[f"{key}.layer_norm.{i}.bias"] = (0.1 * torch.randn(D)).double()
        P[f"{key}.conv.{i}.depthwise_conv.weight"] = (0.5 * torch.randn(D, 1, k)).double()
        P[f"{key}.conv.{i}.depthwise_conv.bias"] = (0.1 * torch.randn(D)).double()
        P[f"{key}.conv.{i}.pointwise_conv.weight"] = (0.4 * torch.randn(D, D, 1)).double()
        P[f"{key}.conv.{i}.pointwise_conv.bias"] = (0.1 * torch.randn(D)).double()
    P[key + ".final_layer_norm.weight"] = (1 + 0.1 * torch.randn(D)).double()
    P[key + ".final_layer_norm.bias"] = (0.1 * torch.randn(D)).double()
    names = [n for n in P if not n.endswith(".pe")]
    valid = torch.from_numpy(np.arange(L)[None, :] < lens.reshape(-1, 1)).double()       # (F, L)
    x0 = torch.randn(F_, L, D, dtype=torch.float64) * valid.unsqueeze(-1)                  # padded features are zero
    w_out = torch.randn(F_, L, D, dtype=torch.float64) * valid.unsqueeze(-1)               # the attention only reads valid positions

    def run(ragged_rows):
        x = x0.clone().requires_grad_(True)
        Pd = {n: (v.clone().requires_grad_(True) if n in names else v) for n, v in P.items()}
        if not ragged_rows:
            y = O.encoder_block(x, valid, Pd, key, n_conv, 0, 0.0, False)
        else:
            y = torch.zeros(F_, L, D, dtype=torch.float64)
            for f in range(F_):
                n = int(t.cq[f, 1])
                if n:
                    y[f, :n] = O.encoder_block(x[f, :n].unsqueeze(0), valid[f, :n].unsqueeze(0), Pd, key, n_conv, 0, 0.0, False)[0]
        (y * w_out).sum().backward()
        return y.detach() * valid.unsqueeze(-1), x.grad, {n: Pd[n].grad for n in names}

    yd, gxd, gpd = run(False)
    yr, gxr, gpr = run(True)
    assert torch.allclose(yd, yr, rtol=1e-12, atol=1e-12)
    keep = torch.from_numpy(np.arange(L)[None, :] < t.cq[:, 1:2]).bool()
    assert float(gxd[~keep].abs().max()) == 0.0
    assert torch.allclose(gxd, gxr, rtol=1e-10, atol=1e-12)
    for n in names:
        assert torch.allclose(gpd[n], gpr[n], rtol=1e-9, atol=1e-11), n


def test_bucket_plan_covers_every_live_frame_once():
    from tvqaplus_amd.ragged import bucket_plan
    rng = np.random.default_rng(5)
    L, halo, step = 512, 6, 64
    lens = rng.integers(0, L + 1, size=300)
    lens[:7] = [0, 1, 58, 59, 506, 507, 512]
    plan = bucket_plan(lens, L, halo, step)
    seen = np.concatenate([ix for ix, _ in plan])
    assert sorted(seen.tolist()) == np.nonzero(lens > 0)[0].tolist()
    for ix, lb in plan:
        assert lb % step == 0 or lb == L
        assert (np.minimum(L, lens[ix] + halo) <= lb).all() and (np.minimum(L, lens[ix] + halo) > lb - step).all()
    assert [lb for _, lb in plan] == sorted(lb for _, lb in plan)
    assert bucket_plan(np.zeros(4, int), L, halo, step) == []


def test_work_table_covers_every_atom_once():
    """RaggedTables.work_table (the balanced launch of the fused [a, b, a*b] backward): every live frame of every group in exactly one
    segment, quads of frames intact for groups of more than 32 words, segments ordered by workgroup AND by group, per-workgroup tile
    counts within one atom (<= 5 tiles) of each other."""
    rng = np.random.default_rng(0)
    for trial in range(120):
        N, NA, Li, Lqa = int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(1, 60)), int(rng.integers(4, 41))
        qa = np.zeros((N, NA, Lqa), bool)
        for n in range(N):
            for a in range(NA):
                qa[n, a, :rng.integers(0, Lqa + 1)] = True
        fl = rng.random((N, Li)) < rng.random()
        tab = ragged.RaggedTables(qa, fl, 4)
        n_wg = int(rng.choice([1, 3, 16, 512]))
        wt = tab.work_table(n_wg)
        G = N * NA
        o1 = ((n_wg + 1) + 3) & ~3
        o2 = o1 + ((2 * G + 3) & ~3)
        wf, gseg, seg = wt[:n_wg + 1], wt[o1:o1 + 2 * G].reshape(G, 2), wt[o2:].reshape(-1, 4)
        assert wf[0] == 0 and wf[-1] == len(seg) and np.all(np.diff(wf) >= 0)
        frames = np.where(tab.Lc > 0, np.repeat(tab.nlive, NA), 0)
        cover = [np.zeros(f, int) for f in frames]
        work = np.zeros(n_wg)
        for si, (g, f0, f1, w) in enumerate(seg):
            assert wf[w] <= si < wf[w + 1] and gseg[g, 0] <= si < gseg[g, 0] + gseg[g, 1]
            assert 0 <= f0 < f1 <= frames[g]
            if tab.Lc[g] > 32:
                assert f0 % 4 == 0
            cover[g][f0:f1] += 1
            work[w] += (f1 - f0) if tab.Lc[g] <= 32 else 5 * ((f1 - f0 + 3) // 4)
        for g in range(G):
            assert np.all(cover[g] == 1)
            assert gseg[g, 1] == int((seg[:, 0] == g).sum())
        if len(seg) and n_wg > 1:
            busy = work[:max(1, int(np.ceil(work.sum() / max(work.max(), 1))))]
            assert work.max() - work.sum() / n_wg <= 5 + 1e-9
