"""csrc/cat3_bwd_dw.hip: the backward of  y = ReLU(drop(LN_3D([a, b, a*b])) W^T + c)  (model/stage.py:381-385, :276-279) with the
Linear's own gradients formed inside -- no saved z.  Held against (i) the kernel it replaces for da / db / d gamma / d beta
(csrc/cat3_fused.hip: stage_cat3_dx_ln_bwd*, itself pinned against the reference fixtures through the whole-model tests) and (ii) an
fp64 contraction over the z the forward kernels write for dW / dc.  Through the C ABI, on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = 128


@pytest.fixture(autouse=True)
def _dw_on(monkeypatch):
    """the kernel is the default since round 6 (STAGE_CAT3_DW=0 switches it off; profiles/r06_cat3_dw_ab.txt)"""
    monkeypatch.setenv("STAGE_CAT3_DW", "1")


def _bits(mask, U):
    """[D/32][rows] int32 words -> (rows, D) bool"""
    return ((mask.unsqueeze(-1) >> torch.arange(32, device=mask.device, dtype=torch.int32)) & 1).permute(1, 0, 2).reshape(U, D).bool()


def _check(new, old, dW_ref, dc_ref, tol=2e-5, tol_w=4e-6):
    names = ("da", "db", "dgamma", "dbeta")
    for nm, x, y in zip(names, old, new[:4]):
        assert torch.isfinite(y).all(), nm
        assert float((x - y).abs().max()) <= tol * float(x.abs().max()) + 1e-6, (nm, float((x - y).abs().max()), float(x.abs().max()))
    dW, dc = new[4], new[5]
    assert torch.isfinite(dW).all() and torch.isfinite(dc).all()
    sw = float(dW_ref.abs().max())
    assert float((dW.double() - dW_ref).abs().max()) <= tol_w * sw + 1e-9, ("dW", float((dW.double() - dW_ref).abs().max()), sw)
    assert float((dc.double() - dc_ref).abs().max()) <= tol_w * float(dc_ref.abs().max()) + 1e-9, "dc"
    assert sw > 0


@pytest.mark.parametrize("rep,inner,G,p,spread", [
    (1, 1, 4096 + 77, 0.1, 0),      # flat rows, last tile partly filled
    (1, 1, 9000, 0.0, 0),
    (1, 1, 20000, 0.1, 24),         # row magnitudes over 2^+-24: the running scale of the row contraction moves many times
    (12, 40, 10, 0.1, 0),           # broadcast a, inner = 40: main + rest tiles
    (31, 29, 5, 0.0, 0),            # inner <= 32: one padded tile per frame
    (29, 29, 5, 0.2, 12),
    (7, 40, 15, 0.1, 0),            # 7 frames: the last group of four is ragged
    (300, 40, 3, 0.1, 0),           # many items per workgroup
])
def test_cat3_bwd_dw_dense(hip_device, rep, inner, G, p, spread):
    from tvqaplus_amd import _lib
    lib = _lib.load()
    U = G * rep * inner if rep > 1 else G
    g = torch.Generator().manual_seed(11)
    a = torch.randn((U // rep) if rep > 1 else U, D, generator=g).cuda()
    b = torch.randn(U, D, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(3 * D, generator=g)).cuda()
    beta = (0.1 * torch.randn(3 * D, generator=g)).cuda()
    W = (0.08 * torch.randn(D, 3 * D, generator=g)).cuda()
    dy = torch.randn(U, D, generator=g)
    if spread:
        dy = dy * torch.exp2(torch.randint(-spread, spread + 1, (U, 1), generator=g).float())
    dy = dy.cuda()
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (D // 32, U), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    st = torch.cuda.current_stream().cuda_stream
    if not lib.stage_cat3_bwd_dw_supported(U, D, rep, inner):
        pytest.skip("dW-inside backward switched off")
    z = torch.empty(U, 3 * D, device="cuda"); mean = torch.empty(U, device="cuda"); rstd = torch.empty(U, device="cuda")
    _lib.check(lib.stage_cat3_layernorm_fwd(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), U, D, rep, inner, 1e-5, p, 4242, st), "ln fwd")
    dyg = (dy * _bits(mask, U)).double()
    dW_ref, dc_ref = dyg.t() @ z.double(), dyg.sum(0)
    a_rows = a.shape[0]

    def outputs():
        return (torch.full((a_rows, D), float("nan"), device="cuda"), torch.full((U, D), float("nan"), device="cuda"),
                torch.empty(3 * D, device="cuda"), torch.empty(3 * D, device="cuda"))
    da0, db0, dg0, dbt0 = outputs()
    wsb = lib.stage_cat3_dx_ln_bwd_ws_bytes(U, D, rep, inner)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.stage_cat3_dx_ln_bwd(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        gamma.data_ptr(), da0.data_ptr(), db0.data_ptr(), dg0.data_ptr(), dbt0.data_ptr(), U, D, rep, inner, p,
                                        4242, ws.data_ptr(), wsb, st), "old fused bwd")
    da1, db1, dg1, dbt1 = outputs()
    dW = torch.full((D, 3 * D), float("nan"), device="cuda"); dc = torch.full((D,), float("nan"), device="cuda")
    wsb = lib.stage_cat3_bwd_dw_ws_bytes(U, D, rep, inner)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    for _ in range(2):                                    # twice: the workspace is reused, the result must not depend on what it held
        _lib.check(lib.stage_cat3_bwd_dw(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), da1.data_ptr(), db1.data_ptr(), dg1.data_ptr(),
                                         dbt1.data_ptr(), dW.data_ptr(), dc.data_ptr(), U, D, rep, inner, p, 4242, ws.data_ptr(), wsb, st),
                   "dW-inside bwd")
    torch.cuda.synchronize()
    _check((da1, db1, dg1, dbt1, dW, dc), (da0, db0, dg0, dbt0), dW_ref, dc_ref)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_cat3_bwd_dw_ragged(hip_device, p):
    """ragged token rows (balanced work table): forward with and without the z store agree bit for bit in everything else; the
    backward against the old fused kernel + the weight-gradient contraction over the z the forward wrote."""
    from tvqaplus_amd import _lib, ragged
    from tvqaplus_amd.ops import _stream
    lib = _lib.load()
    rng = np.random.default_rng(3)
    N, NA, Li, Lqa = 5, 5, 37, 40
    qa = np.zeros((N, NA, Lqa), bool)
    for n in range(N):
        for ai in range(NA):
            qa[n, ai, :rng.integers(0, Lqa + 1)] = True
    qa[0, 0, :] = True
    fl = rng.random((N, Li)) < 0.8
    tab = ragged.RaggedTables(qa, fl, 4)
    lay = ragged.RaggedLayout(tab, hip_device)
    U, Fc, G = lay.U, lay.Fc, N * NA
    if not lib.stage_cat3_bwd_dw_rag_supported(U, Fc, D, G, Li, Lqa):
        pytest.skip("dW-inside backward switched off")
    assert lay.wtab is not None and lay.n_wg == lib.stage_cat3_rag_work_groups() == lib.stage_cat3_bwd_dw_rag_work_groups()
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(U, D, generator=g).cuda()
    W = (0.08 * torch.randn(D, 3 * D, generator=g)).cuda()
    bias = (0.1 * torch.randn(D, generator=g)).cuda()
    a = torch.randn(G * Lqa, D, generator=g).cuda()
    b_fc = torch.randn(Fc, D, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(3 * D, generator=g)).cuda()
    beta = (0.1 * torch.randn(3 * D, generator=g)).cuda()
    st = _stream()
    fwd = []
    for keep_z in (True, False):
        z = torch.full((U, 3 * D), float("nan"), device="cuda"); mean = torch.empty(U, device="cuda"); rstd = torch.empty(U, device="cuda")
        y = torch.empty(U, D, device="cuda"); mask = torch.zeros(D // 32, U, dtype=torch.int32, device="cuda")
        wsb = lib.stage_cat3_ln_gemm_fwd_ws_bytes()
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        _lib.check(lib.stage_cat3_ln_gemm_fwd_rag(a.data_ptr(), b_fc.data_ptr(), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                                  z.data_ptr() if keep_z else None, mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), mask.data_ptr(),
                                                  lay.rowinfo.data_ptr(), U, G * Lqa, Fc, D, 1e-5, p, 4321, ws.data_ptr(), wsb, st), "fwd rag")
        torch.cuda.synchronize()
        fwd.append((z, mean, rstd, y, mask))
    (z, mean, rstd, y, mask), (z_no, mean2, rstd2, y2, mask2) = fwd
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2) and torch.equal(y, y2) and torch.equal(mask, mask2)
    assert torch.isnan(z_no).all() and torch.isfinite(z).all()          # nothing was stored
    dyg = (dy * _bits(mask, U)).double()
    dW_ref, dc_ref = dyg.t() @ z.double(), dyg.sum(0)

    def outputs():
        return (torch.full((G * Lqa, D), float("nan"), device="cuda"), torch.zeros(Fc, D, device="cuda"),
                torch.empty(3 * D, device="cuda"), torch.empty(3 * D, device="cuda"))
    da0, db0, dg0, dbt0 = outputs()
    wsb = lib.stage_cat3_dx_ln_bwd_rag_ws_bytes(G, Li, Lqa)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.stage_cat3_dx_ln_bwd_rag(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b_fc.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), gamma.data_ptr(), da0.data_ptr(), db0.data_ptr(), dg0.data_ptr(), dbt0.data_ptr(),
                                            lay.gdesc.data_ptr(), lay.wtab.data_ptr(), U, Fc, D, G, Li, Lqa, p, 4321, ws.data_ptr(), wsb, st),
               "old rag bwd")
    da1, db1, dg1, dbt1 = outputs()
    dW = torch.full((D, 3 * D), float("nan"), device="cuda"); dc = torch.full((D,), float("nan"), device="cuda")
    wsb = lib.stage_cat3_bwd_dw_rag_ws_bytes(G, Lqa)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.stage_cat3_bwd_dw_rag(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b_fc.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), da1.data_ptr(), db1.data_ptr(), dg1.data_ptr(),
                                         dbt1.data_ptr(), dW.data_ptr(), dc.data_ptr(), lay.gdesc.data_ptr(), lay.wtab.data_ptr(), U, Fc, D, G, Li,
                                         Lqa, p, 4321, ws.data_ptr(), wsb, st), "dW-inside rag bwd")
    torch.cuda.synchronize()
    _check((da1, db1, dg1, dbt1, dW, dc), (da0, db0, dg0, dbt0), dW_ref, dc_ref)
    assert float(da1.abs().max()) > 0


@pytest.mark.parametrize("which,dims,p", [
    ("qa_ctx", dict(N=2, NA=5, Li=12, Lqa=40, Lr=20), 0.1),
    ("qa_ctx", dict(N=1, NA=5, Li=31, Lqa=29, Lr=10), 0.0),
    ("concat_fc", dict(U=4096 + 77), 0.1),
])
def test_group_path_without_z_equals_the_path_with_z(hip_device, which, dims, p, monkeypatch):
    """K-groups (csrc/groups.hip): the default path (STAGE_CAT3_DW unset / 1: forward writes no z, flags[0] = 2, one backward kernel) against STAGE_CAT3_DW=0
    (z written, weight-gradient GEMM + fused dX / LayerNorm backward): identical forward, gradients to summation order."""
    from tvqaplus_amd import groups
    dev = hip_device
    g = torch.Generator().manual_seed(7)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    ln_w, ln_b = (1.0 + 0.1 * torch.randn(3 * D, generator=g)).to(dev).requires_grad_(), rnd(3 * D, scale=0.1).requires_grad_()
    W, c = rnd(D, 3 * D, scale=0.08).requires_grad_(), rnd(D, scale=0.1).requires_grad_()
    seeds = [11, 12, 13]
    results = []
    for no_dw in (True, False):
        monkeypatch.setenv("STAGE_CAT3_DW", "0" if no_dw else "1")
        gg = torch.Generator().manual_seed(3)
        if which == "qa_ctx":
            N, NA, Li, Lqa, Lr = (dims[k] for k in ("N", "NA", "Li", "Lqa", "Lr"))
            qa = (torch.randn(N, NA, Lqa, D, generator=gg)).to(dev).requires_grad_()
            cx = (torch.randn(N, Li, Lr, D, generator=gg)).to(dev).requires_grad_()
            qm = (torch.rand(N, NA, Lqa, generator=gg) > 0.2).float().to(dev)
            cm = (torch.rand(N, Li, Lr, generator=gg) > 0.2).float().to(dev)
            out, S, Sn = groups.qa_ctx(qa, cx, qm, cm, 10.0, p, seeds, [ln_w, ln_b, W, c])
            ins = [qa, cx]
        else:
            U = dims["U"]
            s = torch.randn(U, D, generator=gg).to(dev).requires_grad_()
            v = torch.randn(U, D, generator=gg).to(dev).requires_grad_()
            l2w, l2b = torch.ones(D, device=dev, requires_grad=True), torch.zeros(D, device=dev, requires_grad=True)
            out = groups.concat_fc(s, v, p, seeds[:1], [ln_w, ln_b, W, c, l2w, l2b])
            ins = [s, v]
        go = torch.randn(out.shape, generator=gg).to(dev)
        (out * go).sum().backward()
        torch.cuda.synchronize()
        results.append([out.detach().clone()] + [t.grad.clone() for t in ins] + [ln_w.grad.clone(), ln_b.grad.clone(), W.grad.clone(), c.grad.clone()])
        for t in (ln_w, ln_b, W, c):
            t.grad = None
    assert torch.equal(results[0][0], results[1][0])
    for nm, x, y in zip(["d_a", "d_b", "d_gamma", "d_beta", "dW", "dc"], results[0][1:], results[1][1:]):
        scale = float(x.abs().max()) + 1e-12
        assert float((x - y).abs().max()) <= 2e-5 * scale, (nm, float((x - y).abs().max()), scale)
    assert not torch.equal(results[0][5], results[1][5])  # (different kernels formed dW: bit-identical would mean the switch did nothing)


@pytest.mark.parametrize("rep,inner,G", [(1, 1, 200000), (60, 40, 40)])
def test_cat3_bwd_dw_repeats_bit_for_bit(hip_device, rep, inner, G):
    """the two wave roles of a workgroup hand tiles to each other through LDS (two barriers per tile, LDS-DMA two tiles ahead): a race
    would show as launches of the same inputs that differ.  Every output of six launches, at a size where every workgroup walks many
    tiles, must be the same bits (the accumulation order is fixed: per workgroup, then a fixed-order reduction of the partials)."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    U = G * rep * inner if rep > 1 else G
    if not lib.stage_cat3_bwd_dw_supported(U, D, rep, inner):
        pytest.skip("dW-inside backward switched off")
    g = torch.Generator().manual_seed(23)
    a = torch.randn((U // rep) if rep > 1 else U, D, generator=g).cuda()
    b = torch.randn(U, D, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(3 * D, generator=g)).cuda()
    beta = (0.1 * torch.randn(3 * D, generator=g)).cuda()
    W = (0.08 * torch.randn(D, 3 * D, generator=g)).cuda()
    dy = (torch.randn(U, D, generator=g) * torch.exp2(torch.randint(-6, 7, (U, 1), generator=g).float())).cuda()
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (D // 32, U), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    st = torch.cuda.current_stream().cuda_stream
    z = torch.empty(U, 3 * D, device="cuda"); mean = torch.empty(U, device="cuda"); rstd = torch.empty(U, device="cuda")
    _lib.check(lib.stage_cat3_layernorm_fwd(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), U, D, rep, inner, 1e-5, 0.1, 77, st), "ln fwd")
    del z
    wsb = lib.stage_cat3_bwd_dw_ws_bytes(U, D, rep, inner)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    first = None
    for trial in range(6):
        outs = (torch.full((a.shape[0], D), float("nan"), device="cuda"), torch.full((U, D), float("nan"), device="cuda"),
                torch.empty(3 * D, device="cuda"), torch.empty(3 * D, device="cuda"), torch.empty(D, 3 * D, device="cuda"),
                torch.empty(D, device="cuda"))
        _lib.check(lib.stage_cat3_bwd_dw(dy.data_ptr(), mask.data_ptr(), W.data_ptr(), a.data_ptr(), b.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), *[o.data_ptr() for o in outs], U, D, rep, inner,
                                         0.1, 77, ws.data_ptr(), wsb, st), "dW-inside bwd")
        torch.cuda.synchronize()
        assert all(torch.isfinite(o).all() for o in outs)
        if first is None:
            first = outs
        else:
            for nm, x, y in zip(("da", "db", "dgamma", "dbeta", "dW", "dc"), first, outs):
                assert torch.equal(x, y), (nm, trial)
