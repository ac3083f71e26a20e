"""-m gpu: every C-ABI kernel group against the CPU oracle / a plain fp32 torch reference of the same op.

Tolerance: the north star asks for 1e-3 fp32 on outputs; the kernels use exact-f32 MFMA and fp32 reductions, so the
tests hold them to 2e-4 relative-to-(1+|x|) (summation-order noise only), gradients included."""
import json
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import ENC_CASES, K1_CASES, Fixture, rel_err
from oracle import stage_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-4


def dev(t, grad=False):
    return t.detach().clone().cuda().requires_grad_(grad)


def cpu(t, grad=False):
    return t.detach().clone().cpu().requires_grad_(grad)


def check(name, got, exp, tol=TOL):
    e = rel_err(got, exp)
    assert e < tol, "%s: rel err %.3e >= %.1e" % (name, e, tol)


@pytest.fixture(scope="module")
def ops(hip_device):
    from tvqaplus_amd import ops as _ops
    from tvqaplus_amd import _lib
    assert _lib.load().stage_hip_abi_version() == _lib.ABI_VERSION
    return _ops


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,K", [(7, 16), (33, 128), (10, 300), (5, 768), (1000, 128), (3, 48), (129, 32),
                                     (4099, 128), (5003, 64), (2500, 256)])
def test_layernorm_plain(ops, rows, K):
    g = torch.Generator().manual_seed(rows * 1000 + K)
    x = torch.randn(rows, K, generator=g) * 2 + 0.5
    w, b = torch.randn(K, generator=g), torch.randn(K, generator=g)
    gy = torch.randn(rows, K, generator=g)
    xc, wc, bc = cpu(x, True), cpu(w, True), cpu(b, True)
    F.layer_norm(xc, (K,), wc, bc, 1e-5).backward(gy)
    xd, wd, bd = dev(x, True), dev(w, True), dev(b, True)
    y, s = ops.layernorm(xd, wd, bd)
    assert s is None
    check("y", y, F.layer_norm(x, (K,), w, b, 1e-5))
    y.backward(gy.cuda())
    check("dx", xd.grad, xc.grad)
    check("dgamma", wd.grad, wc.grad)
    check("dbeta", bd.grad, bc.grad)


@pytest.mark.parametrize("M,L,K,period", [(5, 7, 32, 0), (4, 9, 128, 9), (3, 20, 16, 20)])
def test_layernorm_fused_add(ops, M, L, K, period):
    g = torch.Generator().manual_seed(M * 100 + L)
    x = torch.randn(M, L, K, generator=g)
    res = torch.randn((L + 3, K) if period else (M, L, K), generator=g)
    w, b = torch.randn(K, generator=g), torch.randn(K, generator=g)
    gy, gs = torch.randn(M, L, K, generator=g), torch.randn(M, L, K, generator=g)
    xc, rc, wc, bc = cpu(x, True), cpu(res, not period), cpu(w, True), cpu(b, True)
    sc = xc + (rc[:L] if period else rc)
    yc = F.layer_norm(sc, (K,), wc, bc, 1e-5)
    ((yc * gy).sum() + (sc * gs).sum()).backward()
    xd, rd, wd, bd = dev(x, True), dev(res, not period), dev(w, True), dev(b, True)
    y, s = ops.layernorm(xd, wd, bd, res=rd, res_period=period)
    check("y", y, yc)
    check("sum", s, sc)
    ((y * gy.cuda()).sum() + (s * gs.cuda()).sum()).backward()
    check("dx", xd.grad, xc.grad)
    if not period:
        check("dres", rd.grad, rc.grad)
    check("dgamma", wd.grad, wc.grad)
    check("dbeta", bd.grad, bc.grad)


@pytest.mark.parametrize("G,rep,inner,D", [(3, 1, 5, 16), (4, 6, 7, 32), (2, 3, 40, 128), (10, 2, 6, 64),
                                           (5, 7, 123, 128), (3, 1, 1001, 128)])
def test_cat3_layernorm(ops, G, rep, inner, D):
    g = torch.Generator().manual_seed(G * 100 + rep)
    a = torch.randn(G * inner, D, generator=g)
    b = torch.randn(G * rep * inner, D, generator=g)
    w, bb = torch.randn(3 * D, generator=g), torch.randn(3 * D, generator=g)
    gy = torch.randn(G * rep * inner, 3 * D, generator=g)
    ac, bc, wc, bbc = cpu(a, True), cpu(b, True), cpu(w, True), cpu(bb, True)
    ae = ac.view(G, 1, inner, D).expand(G, rep, inner, D).reshape(-1, D)
    yc = F.layer_norm(torch.cat([ae, bc, ae * bc], -1), (3 * D,), wc, bbc, 1e-5)
    yc.backward(gy)
    ad, bd, wd, bbd = dev(a, True), dev(b, True), dev(w, True), dev(bb, True)
    y = ops.cat3_layernorm(ad, bd, wd, bbd, rep=rep, inner=inner)
    check("y", y, yc)
    y.backward(gy.cuda())
    check("da", ad.grad, ac.grad)
    check("db", bd.grad, bc.grad)
    check("dgamma", wd.grad, wc.grad)
    check("dbeta", bbd.grad, bbc.grad)


@pytest.mark.parametrize("M,N,K,relu", [(5, 16, 16, True), (300, 128, 384, True), (257, 300, 768, True),
                                         (1000, 128, 300, False), (77, 1, 128, False), (130, 48, 20, True),
                                         (4100, 128, 128, True), (64, 5, 7, False),
                                         # M >= 4096: the streaming kernels (ragged last row tile, partial column
                                         # tiles, several K chunks, ragged last K chunk, several column tiles)
                                         (5000, 300, 768, True), (4133, 128, 300, False), (8200, 384, 128, True),
                                         (4097, 128, 384, True), (6000, 44, 132, False), (4096, 128, 64, True),
                                         # row counts that are not multiples of 4 (ragged token rows): the ReLU bit mask's word rows are
                                         # then only dword aligned in the weight-gradient kernels
                                         (4099, 128, 128, True), (8191, 128, 384, True), (4102, 300, 768, True), (5001, 128, 300, True)])
def test_linear(ops, M, N, K, relu):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    gy = torch.randn(M, N, generator=g)
    xd, wd, bd = dev(x, True), dev(w, True), dev(b, True)
    y = ops.linear(xd, wd, bd, relu=relu)
    xc, wc, bc = cpu(x, True), cpu(w, True), cpu(b, True)
    yc = F.linear(xc, wc, bc)
    if relu:
        # ReLU is discontinuous: a pre-activation within rounding of 0 may land on the other side.  The backward is
        # checked with the mask the kernel actually produced (its forward is checked against relu separately).
        check("y", y, torch.relu(yc))
        yc = yc * (y.detach().cpu() > 0).float()
    else:
        check("y", y, yc)
    yc.backward(gy)
    y.backward(gy.cuda())
    check("dx", xd.grad, xc.grad)
    check("dw", wd.grad, wc.grad)
    check("db", bd.grad, bc.grad)


@pytest.mark.parametrize("M,N,K", [(4200, 128, 128), (4111, 96, 260)])
def test_gemm_nt_gate_residual_c_abi(ops, M, N, K):
    """stage_gemm_nt with the gate and the residual operand (not reachable through ops.linear) straight through the C-ABI."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    x, gate = torch.randn(M, K, generator=g), torch.randn(M, K, generator=g)
    w, b, res = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    xd, gd, wd, bd, rd = (t.cuda().contiguous() for t in (x, gate, w, b, res))
    y = torch.empty(M, N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.stage_gemm_nt(xd.data_ptr(), gd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y.data_ptr(),
                                 M, N, K, 1, st), "stage_gemm_nt")
    ref = torch.relu(F.linear((x * (gate > 0)).double(), w.double(), b.double())) + res.double()
    check("y", y, ref.float())


@pytest.mark.parametrize("scale_x,scale_w", [(1e-30, 1.0), (1e-20, 1e-18), (1e18, 1e-18), (3e4, 3e4), (1.0, 1e-35)])
def test_linear_dynamic_range(ops, scale_x, scale_w):
    """The streaming GEMMs compute fp32 products as a two-way fp16 split (three products kept) on operands scaled by
    powers of two into fp16's range -- one exponent per X row and one per weight tile (`csrc/gemm_stream.hip`); the tiled
    fallback uses the 3-way bf16 split.  Held to: tiny-but-normal and huge operands reproduce an fp64 product to the usual
    tolerance relative to the result's scale; operands whose products underflow fp32 entirely give (near-)zeros, never
    NaN / Inf."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 4200, 128, 384
    x = torch.randn(M, K, generator=g) * scale_x
    w = torch.randn(N, K, generator=g) * scale_w
    y = ops.linear(x.cuda(), w.cuda())
    ref = (x.double() @ w.double().t())
    assert bool(torch.isfinite(y).all())
    mag = float(ref.abs().max())
    if mag < 1e-37:                                   # the products underflow fp32: nothing but (denormal) zeros is expected
        assert float(y.abs().max()) <= 1e-36
    else:
        err = float((y.double().cpu() - ref).abs().max()) / mag
        # residual terms below 2^-126 flush: with operands at 1e-30 only the hi x hi term survives (bf16-level, 2^-8)
        tol = 2e-5 if min(scale_x, scale_w) > 1e-25 and scale_x * scale_w > 1e-30 else 1e-2
        assert err < tol, (err, tol)


def test_gemm_nt_scale_growth_inside_a_row():
    """fp16-split NT GEMM: the row exponent is set by the first 32-k line and RAISED when a later line would leave fp16's
    range (the wave's accumulators are rescaled).  Rows whose magnitude jumps by 1e3 .. 1e12 in the middle of the row, drops,
    or starts at exactly zero must still match an fp64 product to fp32-class accuracy relative to the row's result."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(21)
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in ((8192, 128, 384), (4500, 200, 256), (4100, 128, 128)):
        x = torch.randn(M, K, generator=g)
        prof = torch.ones(M, K)
        jump = 10.0 ** torch.randint(3, 13, (M,), generator=g).float()
        at = torch.randint(1, K // 32, (M,), generator=g) * 32
        cols = torch.arange(K)[None, :]
        kind = torch.arange(M) % 4
        prof = torch.where((kind == 0)[:, None] & (cols >= at[:, None]), jump[:, None], prof)          # jumps up at a line start
        prof = torch.where((kind == 1)[:, None] & (cols >= at[:, None]), 1.0 / jump[:, None], prof)    # drops
        prof = torch.where((kind == 2)[:, None] & (cols < at[:, None]), torch.zeros(()), prof)         # zero, then data
        prof = torch.where((kind == 3)[:, None] & (cols == (at[:, None] + 5)), jump[:, None], prof)    # one huge element
        x = x * prof * 1e-3
        w = torch.randn(N, K, generator=g) * 0.05
        xd, wd = x.cuda(), w.cuda()
        y = torch.empty(M, N, device="cuda")
        assert lib.stage_gemm_nt(xd.data_ptr(), None, wd.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st) == 0
        ref = x.double() @ w.double().t()
        scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
        err = ((y.double().cpu() - ref).abs() / scale).max()
        assert bool(torch.isfinite(y).all()) and float(err) < 3e-6, (M, N, K, float(err))


@pytest.mark.parametrize("M,N,K", [(40000, 128, 128), (36000, 128, 384), (20000, 300, 768), (9000, 72, 200)])
def test_gemm_tn_scale_growth_along_the_rows(M, N, K):
    """fp16-split weight-gradient GEMM: the scale belongs to an operand COLUMN and follows the column's running maximum down
    the slab; the tile's producer announces every change and the consumers rescale their accumulators.  Columns whose
    magnitude grows by orders of magnitude along the rows (per column at a different row), in both operands, against fp64."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    st = torch.cuda.current_stream().cuda_stream
    rows = torch.arange(M)[:, None].float()

    def ramp(C):
        start = torch.randint(0, M, (1, C), generator=g).float()
        decades = torch.randint(0, 9, (1, C), generator=g).float()
        grow = 10.0 ** (decades * ((rows - start) / (0.1 * M)).clamp(0, 1))      # ramps up over 10 % of the rows
        sign = torch.where(torch.arange(C)[None, :] % 3 == 0, -1.0, 1.0)          # every third column ramps DOWN instead
        return torch.where(sign > 0, grow, 10.0 ** decades / grow) * 10.0 ** torch.randint(-6, 2, (1, C), generator=g).float()
    dy = torch.randn(M, N, generator=g) * ramp(N)
    x = torch.randn(M, K, generator=g) * ramp(K)
    dyd, xd = dy.cuda(), x.cuda()
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    assert lib.stage_gemm_tn(dyd.data_ptr(), None, xd.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st) == 0
    ref = dy.double().t() @ x.double()
    # error scale of an entry: the largest |dy| of its column times the largest |x| of its column, times sqrt(M)
    scale = dy.abs().amax(dim=0).double()[:, None] * x.abs().amax(dim=0).double()[None, :] * (M ** 0.5)
    err = ((dw.double().cpu() - ref).abs() / scale).max()
    assert bool(torch.isfinite(dw).all()) and float(err) < 2e-6, float(err)
    refb = dy.double().sum(dim=0)
    assert float(((db.double().cpu() - refb).abs() / (dy.abs().amax(dim=0).double() * M ** 0.5)).max()) < 1e-5


def test_gemm_accuracy_is_fp32_class():
    """The fp16-split GEMMs against an fp64 product, next to torch's own fp32 matmul on the same operands: forward / dX
    (stage_gemm_nt) and weight gradient (stage_gemm_tn) must not be less accurate than the fp32 library GEMM (factor 1.5
    for the luck of one draw), and far inside the 1e-3 parity bar."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in ((60000, 128, 384), (8192, 384, 128), (5000, 300, 768), (40000, 128, 128)):
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) * 0.1).cuda()
        dy = torch.randn(M, N, generator=g).cuda()
        ref = x.double() @ w.double().t()
        y = torch.empty(M, N, device="cuda")
        assert lib.stage_gemm_nt(x.data_ptr(), None, w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, st) == 0
        e_ours = float((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        e_torch = float(((x @ w.t()).double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        assert e_ours < 1.5 * e_torch and e_ours < 1e-6, ("nt", M, N, K, e_ours, e_torch)
        refw = dy.double().t() @ x.double()
        dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
        wsb = lib.stage_gemm_tn_ws_bytes(M, N, K); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        assert lib.stage_gemm_tn(dy.data_ptr(), None, x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), wsb, st) == 0
        e_ours = float((dw.double() - refw).pow(2).mean().sqrt() / refw.pow(2).mean().sqrt())
        e_torch = float(((dy.t() @ x).double() - refw).pow(2).mean().sqrt() / refw.pow(2).mean().sqrt())
        assert e_ours < 1.5 * e_torch and e_ours < 1e-6, ("tn", M, N, K, e_ours, e_torch)


def test_linear_is_transpose_safe(ops):
    """A = I with an asymmetric weight: catches swapped MFMA output rows/cols."""
    K = 64
    x = torch.eye(K)
    w = torch.arange(K * K, dtype=torch.float32).view(K, K) / 100.0
    y = ops.linear(x.cuda(), w.cuda())
    check("y", y, w.t())


@pytest.mark.parametrize("M,L,D,k", [(3, 5, 16, 7), (6, 20, 128, 7), (5, 40, 128, 5), (2, 9, 32, 3), (4, 2, 16, 5),
                                     (3, 100, 64, 9), (2, 33, 128, 1), (7, 64, 128, 3), (40, 300, 128, 7)])
def test_dwconv(ops, M, L, D, k):
    g = torch.Generator().manual_seed(M * L + k)
    x = torch.randn(M, L, D, generator=g)
    w, b = torch.randn(D, 1, k, generator=g), torch.randn(D, generator=g)
    gy = torch.randn(M, L, D, generator=g)
    xc, wc, bc = cpu(x, True), cpu(w, True), cpu(b, True)
    yc = F.conv1d(xc.transpose(1, 2), wc, bc, padding=k // 2, groups=D).transpose(1, 2)
    yc.backward(gy)
    xd, wd, bd = dev(x, True), dev(w, True), dev(b, True)
    y = ops.dwconv(xd, wd, bd)
    check("y", y, yc)
    y.backward(gy.cuda())
    check("dx", xd.grad, xc.grad)
    check("dw", wd.grad, wc.grad)
    check("db", bd.grad, bc.grad)


@pytest.mark.parametrize("M,L,D,k,period,p", [(3, 5, 16, 7, 0, 0.0), (6, 20, 128, 7, 20, 0.0), (5, 40, 128, 5, 0, 0.0),
                                               (2, 33, 32, 3, -1, 0.0), (7, 70, 128, 5, 0, 0.25), (4, 9, 64, 9, 9, 0.1),
                                               (300, 40, 128, 5, 0, 0.1)])
def test_ln_dwconv_fused_matches_unfused(ops, M, L, D, k, period, p):
    """The fused LayerNorm -> depthwise-conv op against the composition of the two separate ops (same dropout counter):
    outputs, exported sum and every gradient.  period: -1 no residual, 0 full residual, L position-table residual."""
    g = torch.Generator().manual_seed(M * L + k + D)
    x = torch.randn(M, L, D, generator=g)
    res = None if period < 0 else (torch.randn(M, L, D, generator=g) if period == 0 else torch.randn(L, D, generator=g))
    gam, bet = torch.randn(D, generator=g), torch.randn(D, generator=g)
    w, b = torch.randn(D, 1, k, generator=g), torch.randn(D, generator=g)
    gh, gs = torch.randn(M, L, D, generator=g), torch.randn(M, L, D, generator=g)
    outs = []
    for fused in (False, True):
        xd, gd, bd, wd, cd = (dev(t, True) for t in (x, gam, bet, w, b))
        rd = None if res is None else dev(res, period == 0)
        kw = dict(p=p, seed=1234, res=rd, res_period=max(period, 0))
        if fused:
            h, s_ = ops.ln_dwconv(xd, gd, bd, wd, cd, **kw)
        else:
            y, s_ = ops.layernorm(xd, gd, bd, **kw)
            h = ops.dwconv(y, wd, cd)
        loss = (h * gh.cuda()).sum()
        if s_ is not None:
            loss = loss + (s_ * gs.cuda()).sum()
        loss.backward()
        outs.append((h.detach(), None if s_ is None else s_.detach(), xd.grad, None if rd is None else rd.grad, gd.grad,
                     bd.grad, wd.grad, cd.grad))
    names = ("h", "sum", "dx", "dres", "dgamma", "dbeta", "dw", "db")
    for n, a, bb in zip(names, outs[1], outs[0]):
        if a is None or bb is None:
            assert a is None and bb is None, n
            continue
        check(n, a, bb.cpu())


def test_l2norm(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 300, generator=g)
    x[5].zero_()
    check("y", ops.l2norm(x.cuda()), F.normalize(x, p=2, dim=-1))


@pytest.mark.parametrize("R,L,D,win", [(7, 6, 16, False), (10, 40, 128, False), (9, 30, 32, True)])
def test_masked_max(ops, R, L, D, win):
    g = torch.Generator().manual_seed(R + L)
    x = torch.randn(R, L, D, generator=g)
    lens = torch.randint(0, L + 1, (R,), generator=g)
    m = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
    gy = torch.randn(R, D, generator=g)
    xc = cpu(x, True)
    masked = O.mask_logits(xc, m.unsqueeze(2))
    window = None
    if win:
        st = torch.randint(0, L - 1, (R,), generator=g)
        ed = st + 1 + torch.randint(0, 5, (R,), generator=g)
        window = torch.stack([st, ed], 1).int()
        yc = torch.stack([masked[r, st[r]:ed[r]].max(0)[0] for r in range(R)])
    else:
        yc = masked.max(1)[0]
    yc.backward(gy)
    xd = dev(x, True)
    y = ops.masked_max(xd, m.cuda(), None if window is None else window.cuda())
    check("y", y, yc)
    y.backward(gy.cuda())
    check("dx", xd.grad, xc.grad)


@pytest.mark.parametrize("M,L,D,nh", [(3, 5, 16, 2), (6, 20, 128, 4), (4, 50, 32, 4), (2, 64, 32, 1)])
def test_mha_core(ops, M, L, D, nh):
    g = torch.Generator().manual_seed(M * L + nh)
    q, k, v = (torch.randn(M, L, D, generator=g) for _ in range(3))
    lens = torch.randint(0, L + 1, (M,), generator=g)
    lens[0] = L
    m = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()
    go = torch.randn(M, L, D, generator=g)
    dk = D // nh

    def ref(q, k, v):
        sp = lambda t: t.view(M, L, nh, dk).transpose(1, 2)
        s = torch.matmul(sp(q), sp(k).transpose(-2, -1)) / math.sqrt(dk)
        s = s.masked_fill(m.view(M, 1, L, 1) == 0, -1e9)
        return torch.matmul(torch.softmax(s, -1), sp(v)).transpose(1, 2).reshape(M, L, D)

    qc, kc, vc = cpu(q, True), cpu(k, True), cpu(v, True)
    ref(qc, kc, vc).backward(go)
    qd, kd, vd = dev(q, True), dev(k, True), dev(v, True)
    o = ops.mha_core(qd, kd, vd, m.cuda(), nh)
    check("out", o, ref(q, k, v))
    o.backward(go.cuda())
    check("dq", qd.grad, qc.grad)
    check("dk", kd.grad, kc.grad)
    check("dv", vd.grad, vc.grad)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("M,L,D,nh,p", [(5, 20, 128, 4, 0.0), (3, 50, 128, 4, 0.1), (4, 40, 128, 4, 0.1), (6, 13, 32, 4, 0.3), (2, 64, 64, 1, 0.1)])
def test_mha_core_on_fused_projections_equals_separate_tensors(ops, M, L, D, nh, p, bf16):
    """stage_mha_core_qkv_* (q | k | v as the thirds of ONE (M, L, 3D) tensor, row stride 3D; dq | dk | dv written into one gradient
    tensor) against the same kernels on three separate tensors with the same dropout seed: bit for bit, both storage types."""
    g = torch.Generator().manual_seed(M * 10 + L)
    dt = torch.bfloat16 if bf16 else torch.float32
    qkv = torch.randn(M, L, 3 * D, generator=g).to(dt)
    lens = torch.randint(1, L + 1, (M,), generator=g)
    lens[0] = L
    m = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float().cuda()
    go = torch.randn(M, L, D, generator=g).to(dt).cuda()
    assert ops.mha_core_qkv_supported(L, D, nh)
    x = qkv.cuda().requires_grad_()
    o = ops.mha_core_qkv(x, m, nh, p=p, seed=77)
    o.backward(go)
    parts = [qkv[..., j * D:(j + 1) * D].contiguous().cuda().requires_grad_() for j in range(3)]
    o2 = ops.mha_core(parts[0], parts[1], parts[2], m, nh, p=p, seed=77)
    o2.backward(go)
    assert torch.equal(o, o2)
    for j in range(3):
        assert torch.equal(x.grad[..., j * D:(j + 1) * D], parts[j].grad), j


@pytest.mark.parametrize("M,L,D,nh,p", [(5, 20, 128, 4, 0.0), (3, 50, 128, 4, 0.1), (4, 40, 128, 4, 0.1), (2, 64, 64, 1, 0.1),
                                        (6, 13, 32, 4, 0.3), (3, 33, 32, 2, 0.1)])
def test_mha_matrix_core_kernels_match_scalar(ops, M, L, D, nh, p):
    """csrc/mha_mfma.hip (one wave per sequence and head on v_mfma_f32_16x16x4_f32, probabilities recomputed in the
    backward) against the scalar kernels of csrc/mha.hip on the same inputs AND the same dropout seed: the two generate
    the identical (seed, element) mask, so outputs and all three gradients agree to rounding -- with ragged masks (padded
    QUERY rows attend uniformly, padded keys are not masked: the reference's quirk) and every tile count T = 1..4."""
    import os
    g = torch.Generator().manual_seed(M * 100 + L)
    q, k, v = (torch.randn(M, L, D, generator=g) for _ in range(3))
    lens = torch.randint(1, L + 1, (M,), generator=g)
    lens[0] = L
    m = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float().cuda()
    go = torch.randn(M, L, D, generator=g).cuda()
    assert ops._lib.load().stage_mha_core_recomputes(L, D, nh) == 1

    def run():
        qd, kd, vd = dev(q, True), dev(k, True), dev(v, True)
        o = ops.mha_core(qd, kd, vd, m, nh, p=p, seed=991)
        o.backward(go)
        return o.detach(), qd.grad, kd.grad, vd.grad

    fast = run()
    os.environ["STAGE_MHA_SCALAR"] = "1"
    try:
        slow = run()
    finally:
        del os.environ["STAGE_MHA_SCALAR"]
    for name, a, b in zip(("out", "dq", "dk", "dv"), fast, slow):
        check(name, a, b.cpu(), 2e-5)
    if p > 0:
        assert float((fast[0] - run()[0]).abs().max()) == 0.0      # same seed -> same mask, bit for bit


# ---------------------------------------------------------------------------------------------------------------
# K1 against the golden fixtures of the reference and against the oracle on ragged random inputs
# ---------------------------------------------------------------------------------------------------------------
def _k1_run(ops, C, Q, cm, qm, scale, gA=None, gS=None, gSn=None):
    N, NA, _, Lqa, D = C.shape
    _, _, Li, Lr, _ = Q.shape
    Cd, Qd = dev(C.view(N, NA, Lqa, D), True), dev(Q.view(N, Li, Lr, D), True)
    A, S, Sn = ops.structured_attention(Cd, Qd, cm.view(N, NA, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), scale)
    if gA is not None:
        loss = (A * gA.cuda()).sum() + (S * gS.cuda()).sum()
        if gSn is not None:
            loss = loss + (Sn * gSn.cuda()).sum()
        loss.backward()
    return A, S, Sn, Cd, Qd


@pytest.mark.parametrize("name", K1_CASES)
def test_k1_golden(ops, name):
    fx = Fixture(name)
    t = lambda k: torch.from_numpy(fx[k])
    # the model's use: gradients on A and on the raw scores only (expectation from the oracle)
    C, Q = t("C").requires_grad_(), t("Q").requires_grad_()
    Ao, So, Smo, Sno = O.structured_attention(C, Q, t("c_mask"), t("q_mask"), float(fx["scale"]))
    ((Ao * t("gA")).sum() + (So * t("gS")).sum()).backward()
    A, S, Sn, Cd, Qd = _k1_run(ops, t("C"), t("Q"), t("c_mask"), t("q_mask"), float(fx["scale"]), t("gA"), t("gS"))
    check("A", A, t("A"))
    check("S", S, t("S"))
    check("S_norm", Sn, t("S_norm"))
    # gradients: the scale-10 softmax backward dz = p*(dp - <p,dp>) cancels heavily; a 1-ulp change of S_ (x*(1/n)
    # instead of x/n in the forward) already moves dQ by 4.5e-4 of (1+|g|) on k1_sub -> held to the north star's 1e-3
    check("dC", Cd.grad.view_as(C), C.grad, 1e-3)
    check("dQ", Qd.grad.view_as(Q), Q.grad, 1e-3)
    # the public operator: a gradient on the normalised scores as well -- the REFERENCE's own gradients of the fixture
    # (model/context_query_attention.py:61 is differentiable; STAGE never uses it, a caller of ops.structured_attention may)
    if "dC" in fx.z.files and "gSn" in fx.z.files:
        _, _, _, Cd, Qd = _k1_run(ops, t("C"), t("Q"), t("c_mask"), t("q_mask"), float(fx["scale"]), t("gA"), t("gS"), t("gSn"))
        check("dC (with dS_norm)", Cd.grad.view_as(C), t("dC"), 1e-3)
        check("dQ (with dS_norm)", Qd.grad.view_as(Q), t("dQ"), 1e-3)


@pytest.mark.parametrize("N,Li,Lr,Lqa,D", [(2, 5, 4, 6, 16), (1, 9, 25, 13, 32), (2, 7, 20, 40, 128),
                                            (1, 3, 50, 40, 128), (1, 4, 33, 51, 64), (1, 2, 64, 3, 256)])
def test_k1_oracle_ragged(ops, N, Li, Lr, Lqa, D):
    from tvqaplus_amd.synth import make_batch
    g = torch.Generator().manual_seed(N * 7 + Li)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=Li + Lr, empty_frames=True)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g)
    Q = torch.randn(N, 1, Li, Lr, D, generator=g) * 3
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    gA = torch.randn(N, 5, Li, Lqa, D, generator=g)
    gS = torch.randn(N, 5, Li, Lqa, Lr, generator=g) * 0.1
    Cc, Qc = C.clone().requires_grad_(), Q.clone().requires_grad_()
    Ao, So, _, Sno = O.structured_attention(Cc, Qc, cm, qm, 10.0)
    ((Ao * gA).sum() + (So * gS).sum()).backward()
    A, S, Sn, Cd, Qd = _k1_run(ops, C, Q, cm, qm, 10.0, gA, gS)
    check("A", A, Ao)
    check("S", S, So)
    check("S_norm", Sn, Sno)
    check("dC", Cd.grad.view_as(C), Cc.grad, 1e-3)
    check("dQ", Qd.grad.view_as(Q), Qc.grad, 1e-3)


K1_SWEEP_LR = [1, 3, 4, 8, 13, 16, 17, 19, 21, 24, 26, 28, 31, 32, 33, 36, 40, 45, 48, 49, 52, 56, 61, 64]


@pytest.mark.parametrize("Lr", K1_SWEEP_LR)
def test_k1_d128_region_sweep(ops, Lr):
    """Every specialisation of the two D=128 forward kernels (register-resident: Lr <= 32, LDS-staged: Lr <= 64; full /
    permuted / 4x4-block last region tile, vector / scalar score stores) and of the D=128 backward kernels: eval mode
    against the oracle, training mode (dropout 0.3) against the generic kernels on the same seeds."""
    import os
    from tvqaplus_amd.synth import make_batch
    N, Li, Lqa, D = 1, 3, 23, 128
    g = torch.Generator().manual_seed(1000 + Lr)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=Lr, empty_frames=False)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g)
    Q = torch.randn(N, 1, Li, Lr, D, generator=g) * 2
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    gA = torch.randn(N, 5, Li, Lqa, D, generator=g)
    gS = torch.randn(N, 5, Li, Lqa, Lr, generator=g) * 0.1
    Cc, Qc = C.clone().requires_grad_(), Q.clone().requires_grad_()
    Ao, So, _, Sno = O.structured_attention(Cc, Qc, cm, qm, 10.0)
    ((Ao * gA).sum() + (So * gS).sum()).backward()
    A, S, Sn, Cd, Qd = _k1_run(ops, C, Q, cm, qm, 10.0, gA, gS)
    check("A", A, Ao)
    check("S", S, So)
    check("S_norm", Sn, Sno)
    check("dC", Cd.grad.view_as(C), Cc.grad, 1e-3)
    check("dQ", Qd.grad.view_as(Q), Qc.grad, 1e-3)

    def train_run():
        Cd, Qd = dev(C.view(N, 5, Lqa, D), True), dev(Q.view(N, Li, Lr, D), True)
        A, S, Sn = ops.structured_attention(Cd, Qd, cm.view(N, 5, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), 10.0, p=0.3,
                                            seed_c=77, seed_q=4242)
        ((A * gA.cuda()).sum() + (S * gS.cuda()).sum()).backward()
        return A, S, Sn, Cd.grad, Qd.grad
    fast = train_run()
    switches = ("STAGE_K1_GENERIC", "STAGE_K1_BWD_SCALAR", "STAGE_K1_DS_GENERIC")
    saved = {k: os.environ.get(k) for k in switches}
    try:
        for k in switches:
            os.environ[k] = "1"
        ref = train_run()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for name, x, y, tol in zip(("A", "S", "S_norm", "dC", "dQ"), fast, ref, (1e-4, 1e-4, 1e-4, 1e-3, 1e-3)):
        check("train " + name, x, y.cpu(), tol)
    assert float((fast[0] == 0).float().mean()) < 0.9   # not a degenerate all-masked case


@pytest.mark.parametrize("N,Li,Lr,Lqa,ext", [(2, 7, 20, 40, False), (2, 7, 20, 40, True), (1, 5, 50, 40, False),
                                              (1, 5, 50, 40, True), (1, 6, 36, 40, False), (1, 4, 24, 23, False),
                                              (2, 5, 8, 40, False), (1, 3, 64, 12, True), (2, 7, 20, 40, "valid"),
                                              (2, 7, 20, 40, "pad"), (1, 5, 50, 40, "valid"), (1, 5, 50, 40, "pad")])
def test_k1_backward_fused_paths(ops, N, Li, Lr, Lqa, ext):
    """Single-pass backward (csrc/str_attn_bwd_fused.hip), every dispatch: uniform row walk with dA kept in LDS (Lr = 20,
    8), uniform walk with the re-read from L2 (Lr = 50, 36, or STAGE_K1_BWD_NOLDSA), per-lane walk (Lqa = 23, 12); with
    the gradient on raw_s and without it (padded region tiles / empty frames skipped, their gradients are exact zeros).
    ext: True = dense gradient on raw_s (every column processed); "valid" = sparse, on valid regions only, as the
    supervised-attention loss produces it (the skips stay on); "pad" = a few non-zeros on padded regions and in an empty
    frame (those frames fall back to all columns).  Held to the oracle AND to the three-kernel path on the same inputs."""
    import os
    from tvqaplus_amd.synth import make_batch
    D = 128
    g = torch.Generator().manual_seed(31 * Lr + Lqa)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=Lr + Li, empty_frames=True)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g)
    Q = torch.randn(N, 1, Li, Lr, D, generator=g) * 2
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    gA = torch.randn(N, 5, Li, Lqa, D, generator=g)
    gS = torch.randn(N, 5, Li, Lqa, Lr, generator=g) * 0.1 if ext else None
    pair = (cm.view(N, 5, 1, Lqa, 1) * qm.view(N, 1, Li, 1, Lr))
    if ext == "valid":
        gS = gS * pair * (torch.rand(gS.shape, generator=g) < 0.05)
    elif ext == "pad":
        gS = gS * pair * (torch.rand(gS.shape, generator=g) < 0.05)
        nreg = qm.view(N, Li, Lr).sum(-1)
        short = (nreg > 0) & (nreg <= Lr - 4)            # frames with padded regions
        empty_f = nreg == 0
        assert bool(short.any()) and bool(empty_f.any())
        n0, i0 = [int(v[0]) for v in torch.nonzero(short, as_tuple=True)]
        n1, i1 = [int(v[0]) for v in torch.nonzero(empty_f, as_tuple=True)]
        gS[n0, 2, i0, 3, Lr - 1] = 0.7                   # last (padded) region column
        gS[n1, 0, i1, Lqa - 1, 1] = -0.4                 # a frame without any valid region
    Cc, Qc = C.clone().requires_grad_(), Q.clone().requires_grad_()
    Ao, So, _, _ = O.structured_attention(Cc, Qc, cm, qm, 10.0)
    ((Ao * gA).sum() + ((So * gS).sum() if ext else 0.0)).backward()

    def run():
        Cd, Qd = dev(C.view(N, 5, Lqa, D), True), dev(Q.view(N, Li, Lr, D), True)
        A, S, _ = ops.structured_attention(Cd, Qd, cm.view(N, 5, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), 10.0)
        ((A * gA.cuda()).sum() + ((S * gS.cuda()).sum() if ext else 0.0)).backward()
        return Cd.grad.cpu(), Qd.grad.cpu()

    fused = run()
    check("dC", fused[0].view_as(C), Cc.grad, 1e-3)
    check("dQ", fused[1].view_as(Q), Qc.grad, 1e-3)
    # frames without a valid region: the reference's gradient is exactly zero there when no gradient reaches raw_s
    if not ext or ext == "valid":
        empty = (qm.view(N, Li, Lr).sum(-1) == 0)
        assert bool(empty.any())
        assert float(fused[1].view(N, Li, Lr, D)[empty].abs().max()) == 0.0
    old_flag = ops._K1_BWD_UNFUSED
    ops._K1_BWD_UNFUSED = True
    try:
        three = run()
    finally:
        ops._K1_BWD_UNFUSED = old_flag
    check("dC fused vs three-kernel", fused[0], three[0], 2e-4)
    check("dQ fused vs three-kernel", fused[1], three[1], 2e-4)
    # the developer switches select kernels inside the library at first use (static) -- only the python-side switch above
    # can be toggled per call; STAGE_K1_BWD_NOLDSA is exercised by tools/k1_bwd_times.py
    again = run()
    assert torch.equal(fused[0], again[0]) and torch.equal(fused[1], again[1])   # run-to-run deterministic


def test_k1_full_size_vs_oracle(ops):
    """BASELINE.json config 2 video-stream shape (N=16, Li=300, Lr=20, Lqa=40, D=128): direct comparison plus the
    size-independent properties (valid rows of S_ sum to 1, padded rows/frames are exactly 0 / -1e10)."""
    from tvqaplus_amd.synth import make_batch
    N, Li, Lr, Lqa, D = 16, 300, 20, 40, 128
    g = torch.Generator().manual_seed(2018)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=2018)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g)
    Q = torch.randn(N, 1, Li, Lr, D, generator=g)
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    A, S, Sn, _, _ = _k1_run(ops, C, Q, cm, qm, 10.0)
    Ao, So, Smo, Sno = O.structured_attention(C, Q, cm, qm, 10.0)
    check("A", A, Ao)
    check("S", S, So)
    check("S_norm", Sn, Sno)
    rowsum = Sn.sum(-1).cpu()
    valid = (Smo.sum(-1) > 0)
    assert float((rowsum[valid] - 1).abs().max()) < 1e-5
    assert float(rowsum[~valid].abs().max()) == 0.0
    assert bool((S.cpu()[Smo == 0] == -1e10).all())
    A2, S2, Sn2, _, _ = _k1_run(ops, C, Q, cm, qm, 10.0)  # run-to-run determinism
    assert torch.equal(A, A2) and torch.equal(S, S2) and torch.equal(Sn, Sn2)


# ---------------------------------------------------------------------------------------------------------------
# dropout: statistical checks (the mask stream differs from torch's by construction)
# ---------------------------------------------------------------------------------------------------------------
def test_dropout_statistics_and_backward_mask(ops):
    K, rows, p = 128, 4096, 0.1
    x = torch.randn(rows, K).cuda().requires_grad_()
    w, b = torch.ones(K).cuda(), torch.zeros(K).cuda()
    y, _ = ops.layernorm(x, w, b, p=p, seed=1234)
    y0, _ = ops.layernorm(x, w, b)
    kept = (y != 0)
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 5e-3, rate
    check("scaled", y[kept], (y0 / (1 - p))[kept])
    y2, _ = ops.layernorm(x, w, b, p=p, seed=1234)
    assert torch.equal(y, y2)                       # same seed -> same mask
    y3, _ = ops.layernorm(x, w, b, p=p, seed=99)
    assert not torch.equal(y, y3)
    # backward regenerates the same mask: d/dx of sum(y * c) with c = 1 on dropped positions only must vanish
    c = (~kept).float()
    (y * c).sum().backward()
    assert float(x.grad.abs().max()) == 0.0


def test_dropout_stream_is_uniform_and_uncorrelated(ops):
    """The counter-based dropout stream (csrc/common.h: two 32-bit finalisers per group of four elements): keep rates per column,
    per row and overall; no correlation between neighbours (inside a hash group, across groups, across rows), between the thirds
    of a 3D-wide row, or between the masks of different seeds (neighbouring integers and successive states of STAGE's seed
    sequence).  Thresholds are ~5 sigma of the binomial noise."""
    rows, K, p = 8192, 384, 0.1
    x = torch.randn(rows, K).cuda()
    w, b = torch.ones(K).cuda(), torch.zeros(K).cuda()

    def mask(seed):
        y, _ = ops.layernorm(x, w, b, p=p, seed=seed)
        return (y != 0).float()
    s0 = 0x1234567890ABCDEF >> 1
    m = mask(s0)
    n = rows * K
    assert abs(float(m.mean()) - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5 + 1e-4
    assert float((m.mean(0) - (1 - p)).abs().max()) < 5 * (p * (1 - p) / rows) ** 0.5
    assert float((m.mean(1) - (1 - p)).abs().max()) < 5.5 * (p * (1 - p) / K) ** 0.5
    c = m - m.mean()
    var = float((c * c).mean())

    def corr(u, v):
        return abs(float((u * v).mean())) / var
    tol = 5.0 / (n ** 0.5)
    for lag in (1, 2, 3, 4, 5, 8, 32, 128):                 # along a row: inside a group of four, across groups, across the thirds
        assert corr(c[:, :-lag], c[:, lag:]) < tol * 1.1, lag
    for lag in (1, 2, 7):                                    # down the rows
        assert corr(c[:-lag], c[lag:]) < tol * 1.1, lag
    nxt = (s0 * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF     # STAGE._seed's next state
    for other in (s0 + 1, s0 ^ 1, s0 + (1 << 32), nxt >> 1, (nxt * 6364136223846793005 + 1442695040888963407 & 0xFFFFFFFFFFFFFFFF) >> 1):
        m2 = mask(other)
        assert not torch.equal(m, m2)
        assert corr(c, m2 - m2.mean()) < tol * 1.1, hex(other)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: long region rows, D = 256, bf16 storage with fp32 softmax accumulation
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ext", ["dense", "valid", "none"])
@pytest.mark.parametrize("N,Li,Lr,Lqa,D", [(1, 3, 100, 13, 32), (1, 2, 512, 40, 256), (2, 3, 77, 40, 128), (1, 2, 50, 40, 128),
                                            (1, 2, 20, 23, 16), (2, 5, 200, 24, 64)])
def test_k1_long_rows_fp32_vs_oracle(ops, N, Li, Lr, Lqa, D, ext):
    """csrc/str_attn_long.hip in fp32 storage: any Lr (16-region blocks, two-pass softmax; the backward's row term <P, dP> is
    taken as <dA, A>), ragged masks with empty frames, gradient on raw_s included.  Same tolerances as the specialised
    kernels; for Lr <= 64 it must also agree with them.  Region blocks behind a frame's last valid region are skipped by the
    forward always and by the backward unless the gradient on raw_s reaches into them: ``ext`` = that gradient everywhere
    (nothing may be skipped), on valid regions only (the supervised attention loss), or absent."""
    from tvqaplus_amd.synth import make_batch
    g = torch.Generator().manual_seed(Lr * 3 + D)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=Lr + 1, empty_frames=True)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g)
    Q = torch.randn(N, 1, Li, Lr, D, generator=g) * 2
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    gA = torch.randn(N, 5, Li, Lqa, D, generator=g)
    gS = torch.randn(N, 5, Li, Lqa, Lr, generator=g) * 0.1
    if ext == "valid":
        gS = gS * qm.view(N, 1, Li, 1, Lr)
    Cc, Qc = C.clone().requires_grad_(), Q.clone().requires_grad_()
    Ao, So, _, Sno = O.structured_attention(Cc, Qc, cm, qm, 10.0)
    ((Ao * gA).sum() + ((So * gS).sum() if ext != "none" else 0.0)).backward()
    Cd, Qd = dev(C.view(N, 5, Lqa, D), True), dev(Q.view(N, Li, Lr, D), True)
    A, S, Sn = ops.structured_attention_long(Cd, Qd, cm.view(N, 5, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), 10.0)
    ((A * gA.cuda()).sum() + ((S * gS.cuda()).sum() if ext != "none" else 0.0)).backward()
    check("A", A, Ao)
    check("S", S, So)
    check("S_norm", Sn, Sno)
    check("dC", Cd.grad.view_as(C), Cc.grad, 1e-3)
    check("dQ", Qd.grad.view_as(Q), Qc.grad, 1e-3)
    if Lr <= 64 and D % 16 == 0 and Lr % 2 == 0:
        C2, Q2 = dev(C.view(N, 5, Lqa, D), True), dev(Q.view(N, Li, Lr, D), True)
        A2, S2, Sn2 = ops.structured_attention(C2, Q2, cm.view(N, 5, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), 10.0)
        ((A2 * gA.cuda()).sum() + ((S2 * gS.cuda()).sum() if ext != "none" else 0.0)).backward()
        check("A vs specialised", A, A2.cpu(), 1e-5)
        check("dQ vs specialised", Qd.grad, Q2.grad.cpu(), 1e-4)


@pytest.mark.parametrize("N,Li,Lr,Lqa,D", [(2, 2, 512, 40, 256), (1, 3, 50, 40, 128), (1, 2, 77, 23, 64), (1, 2, 100, 13, 32),
                                            (1, 2, 70, 12, 16), (1, 2, 96, 40, 128)])
def test_k1_bf16_storage_vs_fp32_oracle(ops, N, Li, Lr, Lqa, D):
    """bf16 storage (C, Q in, A out, dA in), fp32 scores / softmax / accumulation.  Tolerance rule, stated up front: the
    inputs are first rounded to bf16 and the fp32 ORACLE is evaluated on those rounded values (so only the kernel's own
    roundings remain: the normalised operands and A are stored in bf16, relative error 2^-9 each).  Then
      raw scores (cosines in [-1, 1])  |dS| <= 1e-2        (D products of two 2^-9-accurate factors)
      normalised scores                |dS_| <= 3e-2      (softmax of 10 x cosine: d(logit) <= 0.1)
      A, dC, dQ                        <= 4e-2 * (1 + |ref|)
    -- an order of magnitude looser than the fp32 path's 1e-3, which is what 8 bits of mantissa buy."""
    from tvqaplus_amd.synth import make_batch
    g = torch.Generator().manual_seed(Lr + D)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=Lr + 3)
    bf = lambda t: t.to(torch.bfloat16)
    C = bf(torch.randn(N, 5, 1, Lqa, D, generator=g)).float()
    Q = bf(torch.randn(N, 1, Li, Lr, D, generator=g) * 2).float()
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    gA = bf(torch.randn(N, 5, Li, Lqa, D, generator=g)).float()
    Cc, Qc = C.clone().requires_grad_(), Q.clone().requires_grad_()
    Ao, So, _, Sno = O.structured_attention(Cc, Qc, cm, qm, 10.0)
    (Ao * gA).sum().backward()
    Cd = bf(C.view(N, 5, Lqa, D)).cuda().requires_grad_()
    Qd = bf(Q.view(N, Li, Lr, D)).cuda().requires_grad_()
    A, S, Sn = ops.structured_attention(Cd, Qd, cm.view(N, 5, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), 10.0)
    assert A.dtype == torch.bfloat16 and S.dtype == torch.float32
    (A.float() * gA.cuda()).sum().backward()
    valid = (cm.view(N, 5, 1, Lqa, 1) * qm.view(N, 1, Li, 1, Lr)) > 0
    assert float((S.cpu() - So)[valid.expand_as(So)].abs().max()) < 1e-2
    assert float((Sn.cpu() - Sno).abs().max()) < 3e-2
    check("A", A.float(), Ao, 4e-2)
    check("dC", Cd.grad.float().view_as(C), Cc.grad, 4e-2)
    check("dQ", Qd.grad.float().view_as(Q), Qc.grad, 4e-2)


def test_k1_forward_fp16_split_scores_any_context_scale():
    """Stage 1 of the D = 128 forward kernels (register-resident for Lr <= 32, LDS-staged above) runs as a two-way fp16 split
    with one power-of-two scale per CONTEXT row; the region rows are normalised inside the kernel.  Context rows of wildly different magnitude (1e-6 .. 1e4,
    nothing normalised them), a zero row and a row with one dominant element: raw scores against fp64 to fp32-class accuracy
    relative to |context row| (a cosine-like score is bounded by it); and the attended rows A for frames whose raw rows differ
    by 8 orders of magnitude."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    N, NA, Li, Lqa, D = 2, 5, 7, 40, 128
    for Lr in (20, 16, 32, 9, 50, 40, 64):        # <= 32: register-resident kernel, above: LDS-staged kernel
        Cn = torch.randn(N, NA, Lqa, D, generator=g) * 10.0 ** (torch.rand(N, NA, Lqa, 1, generator=g) * 10 - 6)
        Cn[0, 0, 3] = 0.0
        Cn[1, 2, 5, 17] = 3e6
        Q = torch.randn(N, Li, Lr, D, generator=g) * 10.0 ** torch.randint(-4, 5, (N, Li, 1, 1), generator=g).float()   # per-frame magnitude
        cm = torch.ones(N, NA, Lqa); qm = torch.ones(N, Li, Lr)
        dev = "cuda"
        Cd, Qd, cmd, qmd = Cn.to(dev), Q.to(dev), cm.to(dev), qm.to(dev)
        A = torch.empty(N, NA, Li, Lqa, D, device=dev); S = torch.empty(N, NA, Li, Lqa, Lr, device=dev); Sn = torch.empty_like(S)
        st = torch.cuda.current_stream().cuda_stream
        assert lib.stage_str_attn_fwd(Cd.data_ptr(), Qd.data_ptr(), cmd.data_ptr(), qmd.data_ptr(), A.data_ptr(), S.data_ptr(),
                                      Sn.data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0, 0.0, 0, st) == 0
        Qn = torch.nn.functional.normalize(Q.double(), dim=-1)
        ref = torch.einsum("nawd,nird->naiwr", Cn.double(), Qn)
        rown = Cn.double().norm(dim=-1)[:, :, None, :, None].clamp_min(1e-30)
        err = ((S.double().cpu() - ref).abs() / rown).max()
        assert bool(torch.isfinite(S).all()) and float(err) < 2e-6, (Lr, float(err))
        # stage 2 (fp16 split in the LDS-staged kernel: weights x raw rows, one scale per frame): A against the kernel's own
        # weights times the raw rows in fp64, relative to the frame's largest magnitude
        refA = torch.einsum("naiwr,nird->naiwd", Sn.double().cpu(), Q.double())
        fmax = Q.abs().amax(dim=(2, 3))[:, None, :, None, None].double()
        errA = ((A.double().cpu() - refA).abs() / fmax).max()
        assert bool(torch.isfinite(A).all()) and float(errA) < 2e-6, (Lr, float(errA))


@pytest.mark.parametrize("R,L,with_res", [(37, 40, True), (5, 1, True), (64, 7, False), (300, 40, True), (9, 33, False)])
def test_ln_masked_max_fused(ops, R, L, with_res):
    """LayerNorm + mask_logits + max over the sequence axis in one pass against the two separate operators of this package
    and against plain torch (forward and all gradients); groups with every row
    masked, ties (duplicated rows) and a single valid row included."""
    K = 128
    g = torch.Generator().manual_seed(R * 100 + L)
    x = torch.randn(R, L, K, generator=g)
    res = torch.randn(R, L, K, generator=g) if with_res else None
    if L > 2:
        x[0, 1] = x[0, 0]
        if with_res: res[0, 1] = res[0, 0]          # rows 0 and 1 of group 0 are identical: the first one must win
    gamma, beta = torch.randn(K, generator=g), torch.randn(K, generator=g)
    mask = (torch.rand(R, L, generator=g) > 0.3).float()
    mask[R - 1] = 0.0                                # a group without a valid row
    if R > 2: mask[1] = 0.0; mask[1, L - 1] = 1.0    # a group with one valid row
    gout = torch.randn(R, K, generator=g)

    def run(fused):
        xs = x.clone().cuda().requires_grad_(True)
        rs = res.clone().cuda().requires_grad_(True) if with_res else None
        gm, bt = gamma.clone().cuda().requires_grad_(True), beta.clone().cuda().requires_grad_(True)
        if fused:
            out = ops.ln_masked_max(xs, rs, gm, bt, mask.cuda())
        else:
            y, _ = ops.layernorm(xs, gm, bt, res=rs)
            out = ops.masked_max(y, mask.cuda())
        out.backward(gout.cuda())
        return [out.detach().cpu(), xs.grad.cpu(), rs.grad.cpu() if with_res else None, gm.grad.cpu(), bt.grad.cpu()]
    f, u = run(True), run(False)
    assert float((f[0] - u[0]).abs().max()) <= 1e-5 * (1.0 + float(u[0].abs().max()))   # same arithmetic up to FMA contraction
    for name, a, b in zip(("dx", "dres", "dgamma", "dbeta"), f[1:], u[1:]):
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-5 * (1.0 + float(b.abs().max())), name
    # plain torch reference
    xt = x.clone().double().requires_grad_(True)
    rt = res.clone().double().requires_grad_(True) if with_res else None
    gt, btt = gamma.clone().double().requires_grad_(True), beta.clone().double().requires_grad_(True)
    v = xt + rt if with_res else xt
    y = F.layer_norm(v, (K,), gt, btt, 1e-5)
    m = mask.double()[:, :, None]
    ref = (y * m + (1 - m) * (-1e10)).max(dim=1).values
    ref.backward(gout.double())
    check("out", f[0], ref.detach().float())
    check("dx", f[1], xt.grad.float())
    check("dgamma", f[3], gt.grad.float())
    check("dbeta", f[4], btt.grad.float())


@pytest.mark.parametrize("M,K,N,p", [(4200, 768, 300, 0.1), (5000, 300, 300, 0.0), (9000, 128, 72, 0.25), (4100, 260, 128, 0.1)])
def test_input_ln_linear_fused_backward(ops, M, K, N, p):
    """First layer of the input MLPs (LayerNorm -> Dropout -> Linear -> ReLU on features that need no gradient): the LayerNorm's
    gain / bias gradients reduced inside the dX GEMM's epilogue (stage_gemm_nt_lnparam + stage_dropout_keepmask) against the two
    separate operators with the same dropout seed -- same output, same parameter gradients."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda()
    gamma, beta = torch.randn(K, generator=g), torch.randn(K, generator=g)
    w, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    gout = torch.randn(M, N, generator=g).cuda()
    seed = 1234567
    assert ops.input_ln_linear_supported(x, w)

    def run(fused):
        gm, bt = gamma.clone().cuda().requires_grad_(True), beta.clone().cuda().requires_grad_(True)
        ww, bb = w.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
        if fused:
            h = ops.input_ln_linear(x, gm, bt, ww, bb, p=p, seed=seed)
        else:
            y, _ = ops.layernorm(x, gm, bt, p=p, seed=seed)
            h = ops.linear(y, ww, bb, relu=True)
        h.backward(gout)
        return [h.detach(), gm.grad, bt.grad, ww.grad, bb.grad]
    f, u = run(True), run(False)
    assert torch.equal(f[0], u[0])
    for name, a, c in zip(("dgamma", "dbeta", "dw", "db"), f[1:], u[1:]):
        err = float((a - c).abs().max()) / (1e-6 + float(c.abs().max()))
        assert err < 2e-5, (name, err)


def test_cpp_host_runs_the_c_abi(ops, tmp_path):
    """examples/k1_forward_host.cpp (C++ + HIP runtime, no torch) built here and run: its output sums equal the Python
    binding's on the same deterministic inputs."""
    import shutil, subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "k1_forward_host")
    libdir = os.path.join(root, "tvqaplus_amd")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "k1_forward_host.cpp"), "-L", libdir, "-lstage_hip", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.check_output([exe], env=env, timeout=120).decode().split()
    sa, ss, rows = float(out[1]), float(out[3]), int(out[5])
    N, NA, Li, Lqa, Lr, D = 2, 5, 6, 12, 20, 128
    assert rows == N * NA * Li * Lqa
    C = (torch.sin(0.37 * torch.arange(N * NA * Lqa * D, dtype=torch.float32)) + 0.1).view(N, NA, Lqa, D)
    Q = (torch.cos(0.11 * torch.arange(N * Li * Lr * D, dtype=torch.float32)) * 1.5).view(N, Li, Lr, D)
    cm = ((torch.arange(N * NA * Lqa) % Lqa) < 9).float().view(N, NA, Lqa)
    qm = ((torch.arange(N * Li * Lr) % Lr) < 17).float().view(N, Li, Lr)
    A, S, Sn = ops.structured_attention(C.cuda(), Q.cuda(), cm.cuda(), qm.cuda(), 10.0)
    ra, rs = float(A.double().sum()), float(Sn.double().sum())
    assert abs(sa - ra) < 2e-3 * (1 + abs(ra)) and abs(ss - rs) < 1e-4 * (1 + abs(rs)), (sa, ra, ss, rs)
    assert abs(rs - N * NA * Li * 9) < 1e-2          # every valid context row's weights sum to 1


def test_exact_fp32_gemm_fallback_runs_the_suite():
    """``STAGE_GEMM_F32=1`` (true fp32 MFMA products instead of the fp16 split; read once per process) is the documented exact
    fallback: the Linear / GEMM tests and two whole-model golden cases run under it in a fresh interpreter."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, STAGE_GEMM_F32="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_hip_ops.py"), os.path.join(ROOT, "tests", "test_hip_stage.py"),
           "-k", "test_linear or test_gemm_nt_gate or (test_golden_whole_model and (small_local_train or tiny_eval)) or test_golden_encoder"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


@pytest.mark.parametrize("Lr", [20, 50])
def test_k1_backward_fp16_pairs_are_fp32_class(ops, Lr):
    """The fused K1 backward runs all four products as fp16 pairs (hi + lo, three matrix instructions per fp32 product) under
    power-of-two scales per context row / per frame; against an fp64 evaluation of the oracle its gradients must be as accurate as the
    three-kernel fp32 path on the same inputs -- with output gradients whose rows differ by e^+-4 in magnitude (what the scales are for)."""
    from tvqaplus_amd.synth import make_batch
    N, Li, Lqa, D = 2, 12, 40, 128
    g = torch.Generator().manual_seed(7 + Lr)
    b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=3)
    C = torch.randn(N, 5, 1, Lqa, D, generator=g)
    Q = torch.randn(N, 1, Li, Lr, D, generator=g) * 2
    cm, qm = b.qas_mask.view(N, 5, 1, Lqa), b.vid_mask.view(N, 1, Li, Lr)
    gA = torch.randn(N, 5, Li, Lqa, D, generator=g) * torch.exp(2 * torch.randn(N, 5, Li, Lqa, 1, generator=g))
    Cc, Qc = C.double().requires_grad_(), Q.double().requires_grad_()
    Ao, _, _, _ = O.structured_attention(Cc, Qc, cm.double(), qm.double(), 10.0)
    (Ao * gA.double()).sum().backward()

    def run(unfused):
        old = ops._K1_BWD_UNFUSED
        ops._K1_BWD_UNFUSED = unfused
        try:
            Cd, Qd = dev(C.view(N, 5, Lqa, D), True), dev(Q.view(N, Li, Lr, D), True)
            A, _, _ = ops.structured_attention(Cd, Qd, cm.view(N, 5, Lqa).cuda(), qm.view(N, Li, Lr).cuda(), 10.0)
            (A * gA.cuda()).sum().backward()
            return Cd.grad.cpu().double().view_as(Cc.grad), Qd.grad.cpu().double().view_as(Qc.grad)
        finally:
            ops._K1_BWD_UNFUSED = old

    err = {}
    for name, unf in (("fused", False), ("three", True)):
        dC, dQ = run(unf)
        err[name] = (float((dC - Cc.grad).abs().max() / Cc.grad.abs().max()), float((dQ - Qc.grad).abs().max() / Qc.grad.abs().max()))
    assert err["fused"][0] < 2e-6 and err["fused"][1] < 2e-6, err
    assert err["fused"][0] <= 2 * err["three"][0] + 1e-7 and err["fused"][1] <= 2 * err["three"][1] + 1e-7, err


@pytest.mark.parametrize("Lr,Lqa", [(100, 12), (512, 40), (33, 7)])
def test_long_attention_bf16_dq_blocks_of_32_regions(ops, Lr, Lqa):
    """csrc/str_attn_long.hip: str_attn_long_bwd_dq32_kernel (bf16 storage, D = 256: 32 regions per wave on the bf16 matrix cores,
    weights as hi + lo pairs).  Through the C ABI with the dS workspace PRE-FILLED with NaN: the first backward kernel writes dS only for
    the 16-region blocks that can carry a gradient, so a 32-region block that straddles that limit must not read what lies behind it.
    dQraw = P^T dA and dQn = dS^T Cn against fp64 products of the same operands."""
    from tvqaplus_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    N, Li, NA, D = 2, 3, 5, 256
    g = torch.Generator().manual_seed(Lr + Lqa)
    bf = torch.bfloat16
    dA = torch.randn(N, NA, Li, Lqa, D, generator=g).to(bf).cuda()
    A = torch.randn(N, NA, Li, Lqa, D, generator=g).to(bf).cuda()
    Cn = torch.randn(N, NA, Lqa, D, generator=g).to(bf).cuda()
    Q = torch.randn(N, Li, Lr, D, generator=g).to(bf).cuda()
    Qn = torch.randn(N, Li, Lr, D, generator=g).to(bf).cuda()
    lens = torch.tensor([[1, 17, Lr], [min(Lr, 40), 0, min(Lr, 33)]])        # 0: a frame without a valid region
    qm = (torch.arange(Lr).view(1, 1, Lr) < lens.unsqueeze(-1)).float()
    logits = torch.randn(N, NA, Li, Lqa, Lr, generator=g) - 1e10 * (1 - qm.view(N, 1, Li, 1, Lr))
    Sn = (torch.softmax(logits, -1) * qm.view(N, 1, Li, 1, Lr)).cuda()
    qm = qm.cuda()
    dS_ws = torch.full_like(Sn, float("nan"))
    dQ = torch.full((N, Li, Lr, D), float("nan"), device="cuda")
    dQn = torch.full_like(dQ, float("nan"))
    dCn = torch.empty(N, NA, Lqa, D, device="cuda")
    wsb = lib.stage_str_attn_long_bwd_qm_ws_bytes(N, NA, Li, Lqa, D)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    rc = lib.stage_str_attn_long_bwd_qm(dA.data_ptr(), A.data_ptr(), None, Cn.data_ptr(), Q.data_ptr(), Qn.data_ptr(), Sn.data_ptr(),
                                        qm.data_ptr(), dS_ws.data_ptr(), dQ.data_ptr(), dQn.data_ptr(), dCn.data_ptr(), N, NA, Li, Lqa, Lr, D,
                                        1.0, 1, ws.data_ptr(), wsb, st)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(dQ).all() and torch.isfinite(dQn).all()
    dS = torch.nan_to_num(dS_ws, nan=0.0)                                    # what was not written is an exact zero of the gradient
    ref_raw = torch.einsum("nailr,naild->nird", Sn.double(), dA.double())
    ref_n = torch.einsum("nailr,nald->nird", dS.double(), Cn.double())
    ref_c = torch.einsum("nailr,nird->nald", dS.double(), Qn.double())        # dCn: the same hi + lo scheme in str_attn_long_bwd_ds_kernel
    for name, got, ref in (("dQraw", dQ, ref_raw), ("dQn", dQn, ref_n), ("dCn", dCn, ref_c)):
        err = float((got.double() - ref).abs().max())
        assert err <= 2e-5 * (1.0 + float(ref.abs().max())), (name, err, float(ref.abs().max()))   # bf16 hi + lo weights: 2^-16 per term
    dead = (qm == 0).view(N, Li, Lr, 1).expand_as(dQ)
    assert float(dQ[dead].abs().max()) == 0.0                                # padded regions: P = 0 exactly
