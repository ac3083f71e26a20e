"""-m gpu: ragged token rows (tvqaplus_amd/ragged.py, csrc/ragged.hip and the *_fc / *_rag kernels) against the dense HIP path -- which is
what the reference computes, every padded row included (tests/test_hip_stage.py holds THAT against the reference's golden vectors and the
oracle; the full-size tests there run ragged by default).  Dropout 0: the counter-based dropout stream is indexed by row, so the two
layouts draw different masks by construction; with dropout on the ragged path is checked for sanity and determinism."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import stage_oracle as O

pytestmark = pytest.mark.gpu


def _layout(batch, halo, stream, device):
    from tvqaplus_amd import ragged
    qa, fl = ragged.host_masks(batch, stream)
    return ragged.RaggedLayout(ragged.RaggedTables(qa, fl, halo), device)


@pytest.mark.parametrize("Lr,train", [(20, False), (20, True), (50, True), (12, False), (34, True),
                                      # every tail shape of the two forward kernels (last region tile of 1..4 quads, full tiles)
                                      (8, False), (14, False), (14, True), (16, False), (24, True), (28, False), (32, True), (40, False),
                                      (48, True), (64, False)])
def test_k1_frame_compact_equals_dense(hip_device, Lr, train):
    """stage_str_attn_fwd_fc / stage_str_attn_bwd_fused_fc against the dense entry points: same score maps, the rows of A of every live
    frame bit-identical, dead frames untouched, and the three gradients equal when dA is the dense gradient restricted to live frames."""
    from tvqaplus_amd import _lib
    from tvqaplus_amd.ops import _stream
    from tvqaplus_amd.synth import make_batch
    lib = _lib.load()
    torch.manual_seed(3)
    N, NA, Li, Lqa, D = 3, 5, 11, 40, 128
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=Lr, Lqa=Lqa, wd_size=8, vfeat_size=8, seed=13, empty_frames=True)
    lay = _layout(b, 4, "vid", hip_device)
    dev = hip_device
    qmask, cmask = b.vid_mask.to(dev), b.qas_mask.to(dev)
    Cn = F.normalize(torch.randn(N, NA, Lqa, D), dim=-1).to(dev)
    Q = (torch.randn(N, Li, Lr, D) * qmask.cpu().unsqueeze(-1)).to(dev)
    p, seed = (0.1, 1234567) if train else (0.0, 0)
    A = torch.empty(N, NA, Li, Lqa, D, device=dev)
    S, Sn = torch.empty(N, NA, Li, Lqa, Lr, device=dev), torch.empty(N, NA, Li, Lqa, Lr, device=dev)
    assert lib.stage_str_attn_fwd(Cn.data_ptr(), Q.data_ptr(), cmask.data_ptr(), qmask.data_ptr(), A.data_ptr(), S.data_ptr(), Sn.data_ptr(),
                                  N, NA, Li, Lqa, Lr, D, 10.0, p, seed, _stream()) == 0
    Afc = torch.full((lay.Fc, D), 7.0, device=dev)
    S2, Sn2 = torch.empty_like(S), torch.empty_like(Sn)
    assert lib.stage_str_attn_fwd_fc(Cn.data_ptr(), Q.data_ptr(), cmask.data_ptr(), qmask.data_ptr(), Afc.data_ptr(), S2.data_ptr(),
                                     Sn2.data_ptr(), lay.fmap.data_ptr(), None, N, NA, Li, Lqa, Lr, D, 10.0, p, seed, _stream()) == 0
    assert torch.equal(S, S2) and torch.equal(Sn, Sn2)
    # the same with COMPACT region rows (valid regions + a halo of 3): identical outputs without dropout; with dropout the counter is
    # indexed by compact row, so only the maps' support and the masked entries are compared
    from tvqaplus_amd import ragged
    ct = ragged.CtxTables(b.mask_host["vid_len"], Lr, 3)
    cl = ragged.CtxLayout(ct, dev)
    Qc = Q.view(-1, D)[cl.src_rows[:cl.U].long()].contiguous()
    Afc3 = torch.full((lay.Fc, D), 7.0, device=dev)
    S3, Sn3 = torch.empty_like(S), torch.empty_like(Sn)
    assert lib.stage_str_attn_fwd_fc(Cn.data_ptr(), Qc.data_ptr(), cmask.data_ptr(), qmask.data_ptr(), Afc3.data_ptr(), S3.data_ptr(),
                                     Sn3.data_ptr(), lay.fmap.data_ptr(), cl.cq.data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0, p, seed, _stream()) == 0
    if not train:
        assert torch.equal(S, S3) and torch.equal(Sn, Sn3) and torch.equal(Afc, Afc3)
    else:
        assert torch.equal(S <= -1e9, S3 <= -1e9) and torch.equal(Sn == 0, Sn3 == 0) and bool(torch.isfinite(Afc3).all())
    t = lay.tab
    slots, first = t.fmap[N * Li: N * Li + N], t.fmap[N * Li + N:]
    fm = t.fmap[: N * Li].reshape(N, Li)
    Afc_h, A_h = Afc.cpu().view(-1, Lqa, D), A.cpu()
    touched = np.zeros(Afc_h.shape[0], dtype=bool)
    for n in range(N):
        for a in range(NA):
            for i in range(Li):
                if fm[n, i] >= 0:
                    s = int(first[n] + a * slots[n] + fm[n, i])
                    touched[s] = True
                    assert torch.equal(Afc_h[s], A_h[n, a, i]), (n, a, i)
    dumps = [int(first[n] + a * slots[n] + slots[n] - 1) for n in range(N) for a in range(NA)]
    rest = np.ones_like(touched)
    rest[touched] = False
    rest[dumps] = False
    assert not rest.any()                                            # every sequence is a live frame or a dump slot
    # ---- backward ----
    if Lr % 2:
        return
    dA = torch.randn(N, NA, Li, Lqa, D)
    live_f = torch.from_numpy(fm >= 0)
    dA = dA * live_f.view(N, 1, Li, 1, 1)                            # the statement mask blocks the gradient of dead frames
    dA_fc = torch.full((lay.Fc, D), float("nan")).view(-1, Lqa, D)
    for n in range(N):
        for a in range(NA):
            for i in range(Li):
                if fm[n, i] >= 0:
                    dA_fc[int(first[n] + a * slots[n] + fm[n, i])] = dA[n, a, i]
    dA, dA_fc = dA.to(dev), dA_fc.to(dev)
    assert lib.stage_rag_zero_dump(dA_fc.data_ptr(), lay.fmap.data_ptr(), N, NA, Li, Lqa, D, _stream()) == 0
    assert not torch.isnan(dA_fc).any()
    Qn = torch.empty_like(Q)
    assert lib.stage_l2norm_fwd(Q.data_ptr(), Qn.data_ptr(), None, N * Li * Lr, D, 1e-12, p, seed, _stream()) == 0
    ext = (torch.randn(N, NA, Li, Lqa, Lr) * (torch.rand(N, NA, Li, Lqa, Lr) < 0.01)).to(dev) * qmask.view(N, 1, Li, 1, Lr)
    wsb = lib.stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    for e in (None, ext):
        outs = []
        for fc in (False, True):
            dQ, dQn, dCn = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Cn)
            if fc:
                rc = lib.stage_str_attn_bwd_fused_fc(dA_fc.data_ptr(), None if e is None else e.data_ptr(), Cn.data_ptr(), Q.data_ptr(),
                                                     Qn.data_ptr(), Sn.data_ptr(), qmask.data_ptr(), dQ.data_ptr(), dQn.data_ptr(),
                                                     dCn.data_ptr(), lay.fmap.data_ptr(), None, N, NA, Li, Lqa, Lr, D, 10.0, ws.data_ptr(),
                                                     wsb, _stream())
            else:
                rc = lib.stage_str_attn_bwd_fused(dA.data_ptr(), None if e is None else e.data_ptr(), Cn.data_ptr(), Q.data_ptr(),
                                                  Qn.data_ptr(), Sn.data_ptr(), qmask.data_ptr(), dQ.data_ptr(), dQn.data_ptr(),
                                                  dCn.data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0, ws.data_ptr(), wsb, _stream())
            assert rc == 0
            outs.append((dQ.clone(), dQn.clone(), dCn.clone()))
        for x, y in zip(*outs):
            assert torch.equal(x, y)
        if not train:      # compact region rows: the same gradients on the rows that exist (the others are exact zeros in the dense result)
            dQ, dQn, dCn = torch.full((cl.U, D), 5.0, device=dev), torch.full((cl.U, D), 5.0, device=dev), torch.empty_like(Cn)
            Qnc = torch.empty_like(Qc)
            assert lib.stage_l2norm_fwd(Qc.data_ptr(), Qnc.data_ptr(), None, cl.U, D, 1e-12, 0.0, 0, _stream()) == 0
            assert lib.stage_str_attn_bwd_fused_fc(dA_fc.data_ptr(), None if e is None else e.data_ptr(), Cn.data_ptr(), Qc.data_ptr(),
                                                   Qnc.data_ptr(), Sn.data_ptr(), qmask.data_ptr(), dQ.data_ptr(), dQn.data_ptr(),
                                                   dCn.data_ptr(), lay.fmap.data_ptr(), cl.cq.data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0,
                                                   ws.data_ptr(), wsb, _stream()) == 0
            src = cl.src_rows[:cl.U].long()
            gone = torch.ones(N * Li * Lr, dtype=torch.bool, device=dev)
            gone[src] = False
            assert torch.equal(dQ, outs[0][0].view(-1, D)[src]) and torch.equal(dQn, outs[0][1].view(-1, D)[src])
            assert torch.equal(dCn, outs[0][2])
            assert float(outs[0][0].view(-1, D)[gone].abs().max()) == 0.0 and float(outs[0][1].view(-1, D)[gone].abs().max()) == 0.0


def _pair(opt_kw, batch_kw, device, seed=21):
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_batch, make_opt
    torch.manual_seed(seed)
    opt = make_opt(hsz=128, embedding_size=96, vfeat_size=64, **opt_kw)
    model = STAGE(opt)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    model = model.to(device)
    model.mha_dropout_override = 0.0           # (the reference's self-attention dropout is a fixed 0.1 whatever opt.dropout says)
    batch = make_batch(wd_size=96, vfeat_size=64, **batch_kw)
    return model, batch


def _train_step(model, batch, n_ex, att=False):
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(99)                      # negative sampling of the attention loss draws from the default generator
    (out, targets), att_loss, _, t_loss, t_scores, other = model.forward_main(batch)
    loss = F.cross_entropy(out, targets, reduction="sum") * (n_ex / len(targets)) + 0.5 * t_loss
    if att:
        loss = loss + 0.1 * att_loss
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
    return out.detach(), targets, t_scores.detach(), float(loss), {k: v.detach() for k, v in other.items()}, grads


@pytest.mark.parametrize("cfg", [
    dict(opt=dict(add_local=True, dropout=0.0), batch=dict(N=3, Li=12, Lr=20, Lw=16, Lqa=40, seed=31, empty_frames=True)),
    dict(opt=dict(add_local=True, dropout=0.0, use_sup_att=True), batch=dict(N=2, Li=14, Lr=20, Lw=50, Lqa=40, seed=32, att_imgs=3, att_words=2)),
    dict(opt=dict(add_local=False, dropout=0.0, vfeat_flag=False), batch=dict(N=2, Li=10, Lr=20, Lw=24, Lqa=33, seed=33)),
    dict(opt=dict(add_local=True, dropout=0.0, sub_flag=False), batch=dict(N=3, Li=9, Lr=12, Lw=8, Lqa=40, seed=34, empty_frames=True)),
    dict(opt=dict(add_local=True, dropout=0.0, cls_encoder_kernel_size=3, cls_encoder_n_conv=3), batch=dict(N=2, Li=8, Lr=20, Lw=20, Lqa=29, seed=35)),
    # self-attention in both encoders (BASELINE configs[2]): nothing inside a live frame can be dropped, dead frames still are
    dict(opt=dict(add_local=True, dropout=0.0, input_encoder_n_heads=4, cls_encoder_n_heads=4), batch=dict(N=3, Li=10, Lr=20, Lw=18, Lqa=40, seed=36, empty_frames=True)),
    dict(opt=dict(add_local=True, dropout=0.0, cls_encoder_n_blocks=2), batch=dict(N=2, Li=9, Lr=12, Lw=14, Lqa=24, seed=37)),
])
def test_ragged_model_equals_dense_model(hip_device, cfg):
    """Whole training step, ragged against dense (the reference's semantics): identical proposals, outputs / maps / loss to fp32
    rounding, every parameter gradient to summation order."""
    model, batch = _pair(cfg["opt"], cfg["batch"], hip_device)
    batch = batch.to(hip_device)
    n_ex = len(batch.qid)
    att = bool(cfg["opt"].get("use_sup_att"))
    model.train()
    model.use_ragged = True
    r = _train_step(model, batch, n_ex, att)
    lay = model.last_ragged
    assert lay is not None, "the ragged path did not take this configuration"
    assert lay.U < lay.N * lay.NA * lay.Li * lay.Lqa
    streams = (["sub"] if model.sub_flag else []) + (["vid"] if model.vfeat_flag else [])
    assert sorted(model.last_ragged_ctx) == sorted(streams), "the context streams did not run on ragged rows"
    model.use_ragged_ctx = False                 # ragged statement rows over dense context streams: the same numbers again
    m = _train_step(model, batch, n_ex, att)
    assert model.last_ragged is not None and not model.last_ragged_ctx
    assert torch.equal(r[1], m[1]) and rel_err(r[0], m[0]) < 2e-5 and rel_err(r[2], m[2]) < 2e-5
    assert max(rel_err(r[5][k], m[5][k]) for k in m[5]) < 3e-4
    model.use_ragged_ctx = True
    model.use_ragged = False
    d = _train_step(model, batch, n_ex, att)
    assert model.last_ragged is None
    assert torch.equal(r[1], d[1])
    assert rel_err(r[0], d[0]) < 2e-5 and rel_err(r[2], d[2]) < 2e-5
    assert abs(r[3] - d[3]) < 2e-5 * (1 + abs(d[3]))
    for k in d[4]:
        assert rel_err(r[4][k], d[4][k]) < 2e-5, k
    worst = max((rel_err(r[5][k], d[5][k]), k) for k in d[5])
    assert worst[0] < 3e-4, worst
    # eval / inference conventions through the same layout
    model.eval()
    outs = []
    for flag in (True, False):
        model.use_ragged = flag
        with torch.no_grad():
            o, _, _, tl, tp, other = model.forward_main(batch)
        outs.append((o, tp, tl))
    assert rel_err(outs[0][0], outs[1][0]) < 2e-5 and rel_err(outs[0][1], outs[1][1]) < 2e-5


def test_ragged_model_vs_oracle_fp64(hip_device):
    """The ragged path directly against the CPU oracle (fp64) on a batch with ragged words and dead frames.  (No blanked valid frame
    here: x * m + (1 - m) * -1e10 absorbs x only in fp32, so fp64 is no yardstick for that case -- the dense-vs-ragged test above and
    the small_emptyframe_train fixture cover it.)"""
    model, batch = _pair(dict(add_local=True, dropout=0.0), dict(N=2, Li=13, Lr=20, Lw=18, Lqa=40, seed=41), "cpu")
    opt = model.opt
    P = {k: (v.double().requires_grad_(not k.endswith(".pe")) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    b64 = type(batch)({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()})
    ref = O.stage_forward(P, opt, b64, training=True)
    ref_loss = O.training_loss(ref, n_examples=2)
    ref_loss.backward()
    model = model.to(hip_device).train()
    out, targets, t_scores, loss, other, grads = _train_step(model, batch.to(hip_device), 2)
    assert model.last_ragged is not None
    assert torch.equal(targets.cpu(), ref["targets"])
    assert rel_err(out, ref["logits"]) < 1e-3 and rel_err(t_scores, ref["t_scores"]) < 1e-3
    assert abs(loss - float(ref_loss)) < 1e-3 * (1 + abs(float(ref_loss)))
    for k in ("sub_raw_s", "sub_normalized_s", "vid_raw_s", "vid_normalized_s"):
        assert rel_err(other[k], ref[k]) < 1e-3, k
    for k, g in grads.items():
        e = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        assert rel_err(g, e) < 4e-3, (k, rel_err(g, e))


def test_ragged_layout_from_device_masks_and_dropout(hip_device):
    """Without the loader's host copies the layout comes from ONE read-back of the masks -- the same tables; with dropout on the step
    is finite, deterministic for a given seed state, and consumes the same number of seeds as the dense path."""
    model, batch = _pair(dict(add_local=True), dict(N=2, Li=10, Lr=20, Lw=20, Lqa=40, seed=51), hip_device)
    batch = batch.to(hip_device)
    model.train()
    model._seed_state = 12345
    a = _train_step(model, batch, 2)
    lay_a, seeds_a = model.last_ragged, model._seed_state
    nohost = type(batch)({k: v for k, v in batch.items() if k != "mask_host"})
    model._seed_state = 12345
    b = _train_step(model, nohost, 2)
    assert np.array_equal(lay_a.tab.seq, model.last_ragged.tab.seq) and np.array_equal(lay_a.tab.fmap, model.last_ragged.tab.fmap)
    assert torch.equal(a[0], b[0]) and a[3] == b[3] and model._seed_state == seeds_a
    assert all(torch.isfinite(g).all() for g in a[5].values())
    model.use_ragged = False
    model._seed_state = 12345
    _train_step(model, batch, 2)
    assert model._seed_state == seeds_a


def test_ragged_attention_group_refuses_a_second_backward(hip_device):
    model, batch = _pair(dict(add_local=False, dropout=0.0), dict(N=2, Li=8, Lr=20, Lw=20, Lqa=40, seed=61), hip_device)
    model.train()
    (out, targets), _, _, t_loss, _, _ = model.forward_main(batch.to(hip_device))
    assert model.last_ragged is not None
    loss = out.sum() + t_loss
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="once per forward"):
        loss.backward()


@pytest.mark.parametrize("storage,Lw", [("fp32", 96), ("fp32", 200), ("bf16", 96)])
def test_context_length_buckets_equal_one_dense_batch(hip_device, storage, Lw):
    """Rows longer than 64 (the long-row attention kernels; bf16 storage): the context stream runs as length buckets
    (ragged.bucket_plan, STAGE._base_encoder_buckets) -- same outputs and gradients as the one dense (frames, L, .) batch."""
    model, batch = _pair(dict(add_local=True, dropout=0.0, storage_dtype=storage), dict(N=2, Li=9, Lr=20, Lw=Lw, Lqa=24, seed=41,
                                                                                       empty_frames=True), hip_device)
    batch = batch.to(hip_device)
    n_ex = len(batch.qid)
    model.train()
    r = _train_step(model, batch, n_ex)
    plan = model.last_buckets.get("sub")
    assert plan and len(plan) >= 2 and "vid" not in model.last_buckets, model.last_buckets
    assert sum(n * lb for n, lb in plan) < 2 * 9 * Lw
    model.use_ctx_buckets = False
    d = _train_step(model, batch, n_ex)
    assert not model.last_buckets
    tol_o, tol_g = (2e-5, 3e-4) if storage == "fp32" else (2e-2, 5e-2)
    assert torch.equal(r[1], d[1])
    assert rel_err(r[0], d[0]) < tol_o and rel_err(r[2], d[2]) < tol_o
    for k in d[4]:
        assert rel_err(r[4][k].float(), d[4][k].float()) < tol_o, k
    worst = max((rel_err(r[5][k], d[5][k]), k) for k in d[5])
    assert worst[0] < tol_g, worst


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fused_cat3_backward_balanced_launch_equals_the_chunk_grid(hip_device, p):
    """stage_cat3_dx_ln_bwd_rag with the balanced work table (persistent workgroups walking segments of equal tile counts,
    ragged.RaggedTables.work_table) against the (group, chunk) grid (wtab = NULL): db bit for bit (each row is computed by the same
    code), da / d gamma / d beta to summation order."""
    from tvqaplus_amd import _lib, ragged
    from tvqaplus_amd.ops import _stream
    lib = _lib.load()
    rng = np.random.default_rng(3)
    N, NA, Li, Lqa, D = 5, 5, 37, 40, 128
    qa = np.zeros((N, NA, Lqa), bool)
    for n in range(N):
        for a in range(NA):
            qa[n, a, :rng.integers(0, Lqa + 1)] = True
    qa[0, 0, :] = True
    fl = rng.random((N, Li)) < 0.8
    tab = ragged.RaggedTables(qa, fl, 4)
    lay = ragged.RaggedLayout(tab, hip_device)
    assert lay.wtab is not None and lay.n_wg == lib.stage_cat3_rag_work_groups()
    U, Fc, G = lay.U, lay.Fc, N * NA
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(U, D, generator=g).cuda()
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (D // 32, U), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    W = (0.08 * torch.randn(D, 3 * D, generator=g)).cuda()
    a = torch.randn(G * Lqa, D, generator=g).cuda()
    b_fc = torch.randn(Fc, D, generator=g).cuda()
    mean = (0.1 * torch.randn(U, generator=g)).cuda(); rstd = (1 + 0.1 * torch.rand(U, generator=g)).cuda()
    gamma = (1 + 0.1 * torch.randn(3 * D, generator=g)).cuda()
    wsb = lib.stage_cat3_dx_ln_bwd_rag_ws_bytes(G, Li, Lqa)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    outs = []
    for wt in (None, lay.wtab.data_ptr()):
        da = torch.full((G * Lqa, D), float("nan"), device="cuda"); db = torch.zeros(Fc, D, device="cuda")
        dg = torch.empty(3 * D, device="cuda"); dbt = torch.empty(3 * D, device="cuda")
        _lib.check(lib.stage_cat3_dx_ln_bwd_rag(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b_fc.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), gamma.data_ptr(), da.data_ptr(), db.data_ptr(), dg.data_ptr(), dbt.data_ptr(),
                                                lay.gdesc.data_ptr(), wt, U, Fc, D, G, Li, Lqa, p, 4321, ws.data_ptr(), wsb, _stream()), "cf rag")
        torch.cuda.synchronize()
        outs.append((da, db, dg, dbt))
    assert torch.equal(outs[0][1], outs[1][1])
    for x, y, nm in zip(outs[0], outs[1], ("da", "db", "dgamma", "dbeta")):
        assert torch.isfinite(y).all(), nm
        assert float((x - y).abs().max()) <= 2e-5 * float(x.abs().max()) + 1e-6, nm
    assert float(outs[1][0].abs().max()) > 0


def test_small_batches_run_the_padded_rows(hip_device):
    """stage.py: ragged_min_rows -- below 200 000 padded statement rows (a rank of a strong-scaled job) the step is bound by the host and
    the ragged layout is not built; the threshold is a property of the model, 0 forces the layout (what this suite runs with)."""
    from tvqaplus_amd.stage import STAGE
    from tvqaplus_amd.synth import make_batch, make_opt
    torch.manual_seed(1)
    model = STAGE(make_opt(hsz=128, embedding_size=64, vfeat_size=48, dropout=0.0, add_local=True)).to(hip_device).train()
    batch = make_batch(N=2, Li=8, Lr=10, Lw=12, Lqa=14, wd_size=64, vfeat_size=48, seed=3).to(hip_device)
    assert model.ragged_min_rows == 0                       # tests/conftest.py: STAGE_RAGGED_MIN_ROWS=0
    model(batch)
    assert model.last_ragged is not None
    model.ragged_min_rows = 200000                          # the product default: 2 * 5 * 8 * 14 rows are far below it
    (out_d, tgt_d), _, _, tl_d, ts_d = model(batch)
    assert model.last_ragged is None and not model.last_ragged_ctx
    model.ragged_min_rows = 0
    (out_r, tgt_r), _, _, tl_r, ts_r = model(batch)
    assert model.last_ragged is not None
    assert torch.equal(tgt_d, tgt_r) and rel_err(out_d, out_r) < 2e-5 and rel_err(ts_d, ts_r) < 2e-5 and rel_err(tl_d, tl_r) < 2e-5
