"""Autograd-aware host wrappers around the C-ABI HIP kernels (one ``torch.autograd.Function`` per fused group).

PyTorch is plumbing here: it owns device memory (``torch.empty``), the stream (``torch.cuda.current_stream``) and
the autograd tape.  All arithmetic of the hot path happens in ``libstage_hip.so`` through ``ctypes``; tensors cross
the boundary as raw device pointers.  CPU tensors are rejected -- there is no fallback path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

EPS_LN = 1e-5
EPS_L2 = 1e-12


# ---------------------------------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------------------------------
def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of the current stream.  ``torch.cuda.current_stream().cuda_stream`` builds a Stream object (2.7 us per
    call, several calls per op); the C accessor behind it costs 0.2 us.  The small-kernel tail of a step is host-bound."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.StageHipError("tvqaplus_amd.ops: %s is a %s tensor; the HIP path needs device tensors "
                                 "(there is no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s: expected %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


_BF16 = torch.bfloat16


def _act(t: torch.Tensor, name: str, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """An activation tensor: float32, or bfloat16 in the bf16 storage mode (BASELINE.json configs[4]: the tensors that
    stream through HBM between kernels are bf16, parameters / statistics / masks stay fp32).  All activations of one
    op share one type."""
    if t.dtype not in (torch.float32, _BF16):
        raise TypeError("%s: expected float32 or bfloat16, got %s" % (name, t.dtype))
    if like is not None and t.dtype != like.dtype:
        raise TypeError("%s: %s, but the op runs in %s" % (name, t.dtype, like.dtype))
    return _chk(t, name, t.dtype)


def _sfx(t: torch.Tensor) -> str:
    return "_bf16" if t.dtype == _BF16 else ""


def _on_device(fn):
    """Run an autograd forward / backward on the device of its tensors.  The reference driver moves the model with
    ``model.to(cuda:N)`` without ``torch.cuda.set_device`` (``--device_ids 1``): buffers then live on cuda:N while the
    current device is 0, and a launch on device 0's stream would be cross-device and unordered against torch's own work
    on cuda:N.  All tensor operands must share one device."""
    def wrapped(ctx, *args):
        dev = None
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device.index
                elif a.device.index != dev:
                    raise _lib.StageHipError("tvqaplus_amd.ops: operands on different devices (cuda:%d and cuda:%d)"
                                             % (dev, a.device.index))
        if dev is None or dev == torch.cuda.current_device():
            return fn(ctx, *args)
        with torch.cuda.device(dev):
            return fn(ctx, *args)
    wrapped.__name__ = fn.__name__
    wrapped.__doc__ = fn.__doc__
    return staticmethod(wrapped)


_WS = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Stream-ordered scratch, grown on demand (all ops of one process run on one stream)."""
    key = (device.index, _stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


_FN = {}


def _call(name: str, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib.load(), name)
    rc = fn(*args)
    if rc != 0:
        _lib.check(rc, name)


# ---------------------------------------------------------------------------------------------------------------
# Parameter-gradient sinks (tvqaplus_amd/groups.py: gate): a parameter alias handed out by ``groups.gate`` carries ``_stage_sink``;
# a backward that holds such an alias puts its gradient into the sink (first contribution kept, later ones added with one
# multi-tensor add per call) and returns None for it -- the K-groups do this, and so do the per-kernel LayerNorm / Linear /
# LayerNorm->dwconv ops below, so that a shared module applied once per stream and length bucket (the bf16 stress configuration:
# 220 AccumulateGrad additions per step) costs one fused add per application instead of one kernel per parameter and application.
# ---------------------------------------------------------------------------------------------------------------
def _sinks(params) -> list:
    return [getattr(w, "_stage_sink", None) for w in params]


def _deliver(sinks, grads) -> tuple:
    """Parameter gradients of a group call: into the sinks of gated parameters (returned as None), as they are otherwise."""
    if not any(sinks):
        return tuple(grads)
    out, per = [], {}
    for sk, g in zip(sinks, grads):
        if sk is None:
            out.append(g)
        else:
            e = per.setdefault(id(sk[0]), (sk[0], [], []))
            e[1].append(sk[1])
            e[2].append(g)
            out.append(None)
    for sink, idx, gs in per.values():
        sink.add(idx, gs)
    return tuple(out)



# ---------------------------------------------------------------------------------------------------------------
# LayerNorm (+ fused residual add / position table, + fused dropout)
# ---------------------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, res, gamma, beta, res_period: int, want_sum: bool, p: float, seed: int):
        x = _act(x, "x")
        K = x.shape[-1]
        rows = x.numel() // K
        if res is not None and res_period > 0 and res.dtype != x.dtype:
            res = res.to(x.dtype)          # the (L, D) position table follows the storage type
        res_c = None if res is None else _act(res, "res", x)
        ctx.sinks = _sinks((gamma, beta))
        gamma, beta = _chk(gamma, "gamma"), _chk(beta, "beta")
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        s = torch.empty_like(x) if (want_sum and res_c is not None) else None
        _call("stage_layernorm_fwd" + _sfx(x), _ptr(x), _ptr(res_c), int(res_period), _ptr(s), _ptr(gamma), _ptr(beta), _ptr(y),
              _ptr(mean), _ptr(rstd), rows, K, EPS_LN, float(p), int(seed), _stream())
        if res_c is not None and s is None:
            # backward needs the normalised input; recompute-free path requires the sum -> always keep it when training
            raise RuntimeError("internal: layernorm with residual requires want_sum=True")
        xin = s if res_c is not None else x
        ctx.save_for_backward(xin, mean, rstd, gamma)
        # the exported sum of the LAST LayerNorm of a residual chain has no consumer: without this autograd hands the
        # backward a materialised zero tensor for it (a full-size fill plus one more read as dx_add)
        ctx.set_materialize_grads(False)
        ctx.p, ctx.seed = float(p), int(seed)
        ctx.has_res = res_c is not None
        ctx.res_full = res_c is not None and res_period == 0
        if s is None:
            s = y.new_empty(0)
        return y, s

    @_on_device
    def backward(ctx, dy, dsum):
        xin, mean, rstd, gamma = ctx.saved_tensors
        K = xin.shape[-1]
        rows = xin.numel() // K
        if dy is None:       # only the exported sum was used downstream
            dy = torch.zeros_like(xin)
        dy = _act(dy, "dy", xin)
        x_needs = ctx.needs_input_grad[0]
        res_needs = ctx.needs_input_grad[1] and ctx.res_full
        need_dx = x_needs or res_needs
        dx = torch.empty_like(xin) if need_dx else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        lib = _lib.load()
        wsb = lib.stage_ln_bwd_ws_bytes(K)
        ws = _workspace(wsb, xin.device)
        dadd = None  # gradient arriving through the exported sum (next residual branch), fused into the dx store
        if dx is not None and ctx.has_res and dsum is not None and dsum.numel() == dx.numel():
            dadd = _act(dsum, "dsum", xin)
        _call("stage_layernorm_bwd" + _sfx(xin), _ptr(dy), _ptr(xin), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx), _ptr(dadd),
              _ptr(dgamma), _ptr(dbeta), rows, K, ctx.p, ctx.seed, _ptr(ws), wsb, _stream())
        dgamma, dbeta = _deliver(ctx.sinks, (dgamma, dbeta))
        return (dx if x_needs else None), (dx if res_needs else None), dgamma, dbeta, None, None, None, None


def layernorm(x, gamma, beta, p: float = 0.0, seed: int = 0, res=None, res_period: int = 0
              ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """y = drop(LN(x + res)); returns (y, x + res or None)."""
    y, s = _LayerNorm.apply(x, res, gamma, beta, res_period, res is not None, p, seed)
    return y, (s if res is not None else None)


# ---------------------------------------------------------------------------------------------------------------
# LayerNorm over cat([a, b, a*b])
# ---------------------------------------------------------------------------------------------------------------
class _Cat3LayerNorm(torch.autograd.Function):
    @_on_device
    def forward(ctx, a, b, gamma, beta, rep: int, inner: int, p: float, seed: int):
        b = _act(b, "b")
        a = _act(a, "a", b)
        D = b.shape[-1]
        rows = b.numel() // D
        gamma, beta = _chk(gamma, "gamma"), _chk(beta, "beta")
        assert a.numel() * rep == b.numel(), (a.shape, b.shape, rep)
        y = torch.empty(b.shape[:-1] + (3 * D,), dtype=b.dtype, device=b.device)
        mean = torch.empty(rows, dtype=torch.float32, device=b.device)
        rstd = torch.empty_like(mean)
        _call("stage_cat3_layernorm_fwd" + _sfx(b), _ptr(a), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd),
              rows, D, int(rep), int(inner), EPS_LN, float(p), int(seed), _stream())
        ctx.save_for_backward(a, b, mean, rstd, gamma)
        ctx.cfg = (int(rep), int(inner), float(p), int(seed))
        return y

    @_on_device
    def backward(ctx, dy):
        a, b, mean, rstd, gamma = ctx.saved_tensors
        rep, inner, p, seed = ctx.cfg
        D = b.shape[-1]
        rows = b.numel() // D
        dy = _act(dy, "dy", b)
        db = torch.empty_like(b)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        lib = _lib.load()
        d4 = D // 4
        if b.dtype == _BF16:
            if rep > 1 and D % 4 == 0 and 4 <= d4 <= 64 and (d4 & (d4 - 1)) == 0 and inner <= 64:
                da = torch.empty(a.shape, dtype=torch.float32, device=a.device)   # a sum over the frames: fp32, rounded once
                wsb = lib.stage_cat3_layernorm_bwd_reduced_ws_bytes(rows, D, rep, inner)
                ws = _workspace(wsb, b.device)
                rc = lib.stage_cat3_layernorm_bwd_reduced_bf16(_ptr(dy), _ptr(a), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                                               _ptr(da), _ptr(db), _ptr(dgamma), _ptr(dbeta), rows, D, rep, inner,
                                                               p, seed, _ptr(ws), wsb, _stream())
                if rc != _lib.STAGE_ERR_SHAPE:
                    _lib.check(rc, "stage_cat3_layernorm_bwd_reduced_bf16")
                    return da.to(_BF16), db, dgamma, dbeta, None, None, None, None
            # the unreduced da rows stay fp32 (scratch), reduced over the broadcast, rounded once
            da_full = torch.empty(b.shape, dtype=torch.float32, device=b.device)
            wsb = lib.stage_ln_bwd_ws_bytes(3 * D)
            ws = _workspace(wsb, b.device)
            _call("stage_cat3_layernorm_bwd_bf16", _ptr(dy), _ptr(a), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(gamma),
                  _ptr(da_full), _ptr(db), _ptr(dgamma), _ptr(dbeta), rows, D, rep, inner, p, seed, _ptr(ws), wsb, _stream())
            if rep > 1:
                da = torch.empty(a.shape, dtype=torch.float32, device=a.device)
                _call("stage_reduce_rep", _ptr(da_full), _ptr(da), a.numel() // (inner * D), rep, inner * D, _stream())
            else:
                da = da_full.view(a.shape)
            return da.to(_BF16), db, dgamma, dbeta, None, None, None, None
        if rep > 1 and D % 4 == 0 and 4 <= d4 <= 64 and (d4 & (d4 - 1)) == 0 and inner <= 64:
            # broadcast reduction fused into the backward kernel: no (rows, D) intermediate
            da = torch.empty_like(a)
            wsb = lib.stage_cat3_layernorm_bwd_reduced_ws_bytes(rows, D, rep, inner)
            ws = _workspace(wsb, b.device)
            rc = lib.stage_cat3_layernorm_bwd_reduced(_ptr(dy), _ptr(a), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(da),
                                                      _ptr(db), _ptr(dgamma), _ptr(dbeta), rows, D, rep, inner, p, seed,
                                                      _ptr(ws), wsb, _stream())
            if rc != _lib.STAGE_ERR_SHAPE:      # e.g. D = 256 with 40 inner rows: the generic path below
                _lib.check(rc, "stage_cat3_layernorm_bwd_reduced")
                return da, db, dgamma, dbeta, None, None, None, None
        da_full = torch.empty_like(b)
        wsb = lib.stage_ln_bwd_ws_bytes(3 * D)
        ws = _workspace(wsb, b.device)
        _call("stage_cat3_layernorm_bwd", _ptr(dy), _ptr(a), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(gamma),
              _ptr(da_full), _ptr(db), _ptr(dgamma), _ptr(dbeta), rows, D, rep, inner, p, seed, _ptr(ws), wsb,
              _stream())
        if rep > 1:
            da = torch.empty_like(a)
            groups = a.numel() // (inner * D)
            _call("stage_reduce_rep", _ptr(da_full), _ptr(da), groups, rep, inner * D, _stream())
        else:
            da = da_full.view(a.shape)
        return da, db, dgamma, dbeta, None, None, None, None


def cat3_layernorm(a, b, gamma, beta, rep: int = 1, inner: int = 1, p: float = 0.0, seed: int = 0):
    return _Cat3LayerNorm.apply(a, b, gamma, beta, rep, inner, p, seed)


# ---------------------------------------------------------------------------------------------------------------
# Linear (+bias, +ReLU) on the matrix cores
# ---------------------------------------------------------------------------------------------------------------
_WT_CACHE = {}   # (id(weight), stream) -> (data_ptr, version, transposed copy, weakref, epoch); entries die with their weight
_WT_EPOCH = 0


def new_step() -> None:
    """Called by STAGE.forward_main at the top of every forward: transposed-weight copies made during earlier steps are no
    longer trusted.  (The version counter alone misses in-place writes through ``p.data`` -- ``dist.broadcast(p.data)``,
    manual EMA, some third-party optimizers -- so a copy is only reused inside the forward/backward pair it was made in,
    where a shared module (the input encoder: QA, subtitle and video streams) asks for the same transpose several times.)"""
    global _WT_EPOCH
    _WT_EPOCH += 1


def invalidate_weight_cache() -> None:
    """Drop every cached transposed weight (for callers that rewrite weights between a forward and its backward)."""
    _WT_CACHE.clear()


def _transposed_weight(w, w2):
    """(K, N) copy of the (N, K) weight for the dX GEMM, cached per weight object for the current step (see ``new_step``)."""
    import weakref
    # one copy per stream: a shared module's backward runs on the stream of each of its applications (stage.py: branch streams), and a
    # copy made on one stream is not ordered against a reader on another
    key = (id(w), torch.cuda.current_stream(w2.device).cuda_stream if w2.is_cuda else 0)
    hit = _WT_CACHE.get(key)
    if (hit is not None and hit[4] == _WT_EPOCH and hit[0] == w2.data_ptr() and hit[1] == w._version
            and hit[2].shape == (w2.shape[1], w2.shape[0])):
        return hit[2]
    wt = w2.t().contiguous()
    try:
        ref = weakref.ref(w, lambda _r, k=key: _WT_CACHE.pop(k, None))
    except TypeError:
        return wt
    _WT_CACHE[key] = (w2.data_ptr(), w._version, wt, ref, _WT_EPOCH)
    return wt


class _Linear(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, w, bias, relu: bool):
        x = _act(x, "x")
        bf = x.dtype == _BF16
        ctx.sinks = _sinks((w, bias))
        w2 = _chk(w, "w").reshape(w.shape[0], -1)  # (N, K) ; pointwise Conv1d weights are (N, K, 1)
        N, K = w2.shape
        assert x.shape[-1] == K, (x.shape, w.shape)
        M = x.numel() // K
        bias_c = None if bias is None else _chk(bias, "bias")
        y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        lib = _lib.load()
        mask = None
        if bf:      # bf16 storage: bf16 operands, the fp32 weight is rounded to bf16 while it is staged, fp32 accumulation
            _call("stage_gemm_nt_bf16", _ptr(x), None, _ptr(w2), _ptr(bias_c), None, _ptr(y), M, N, K, int(relu), _stream())
            ctx.save_for_backward(x, w2, y if relu else None, None)
            ctx.relu, ctx.wshape, ctx.w_obj, ctx.has_bias = relu, w.shape, w, bias is not None
            return y
        if relu and lib.stage_gemm_mask_supported(M, N, K) and x.data_ptr() % 16 == 0 and w2.data_ptr() % 16 == 0:
            # Linear + ReLU on the streaming kernel: also emit the ReLU bit mask (1 bit per output) for the backward GEMMs
            mask = torch.empty((N + 31) // 32, M, dtype=torch.int32, device=x.device)   # word-major [ceil(N/32)][M]
            rc = lib.stage_gemm_nt_mask(_ptr(x), None, _ptr(w2), _ptr(bias_c), _ptr(y), _ptr(mask), M, N, K, 1, _stream())
            if rc == _lib.STAGE_ERR_SHAPE:
                mask = None
            else:
                _lib.check(rc, "stage_gemm_nt_mask")
        if mask is None:
            _call("stage_gemm_nt", _ptr(x), None, _ptr(w2), _ptr(bias_c), None, _ptr(y), M, N, K, int(relu), _stream())
        ctx.save_for_backward(x, w2, y if relu else None, mask)
        ctx.relu = relu
        ctx.wshape = w.shape
        ctx.w_obj = w
        ctx.has_bias = bias is not None
        return y

    @_on_device
    def backward(ctx, dy):
        x, w2, y, mask = ctx.saved_tensors
        N, K = w2.shape
        M = x.numel() // K
        dy = _act(dy, "dy", x)
        gate = y if ctx.relu else None
        lib = _lib.load()
        if x.dtype == _BF16:
            dx = None
            if ctx.needs_input_grad[0]:
                wt = _transposed_weight(ctx.w_obj, w2)
                dx = torch.empty_like(x)
                _call("stage_gemm_nt_bf16", _ptr(dy), _ptr(gate), _ptr(wt), None, None, _ptr(dx), M, K, N, 0, _stream())
            dw = torch.empty_like(w2)
            db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            wsb = lib.stage_gemm_tn_bf16_ws_bytes(M, N, K)
            ws = _workspace(wsb, x.device)
            _call("stage_gemm_tn_bf16", _ptr(dy), _ptr(gate), _ptr(x), _ptr(dw), _ptr(db), M, N, K, _ptr(ws), wsb, _stream())
            dwv, db = _deliver(ctx.sinks, (dw.view(ctx.wshape), db))
            return dx, dwv, db, None
        use_mask = mask is not None and dy.data_ptr() % 16 == 0
        dx = None
        if ctx.needs_input_grad[0]:
            wt = _transposed_weight(ctx.w_obj, w2)  # (K, N): dX = (dY .* gate) . W  ==  NT with the transposed weight
            dx = torch.empty_like(x)
            done = False
            if use_mask:
                rc = lib.stage_gemm_nt_mask(_ptr(dy), _ptr(mask), _ptr(wt), None, _ptr(dx), None, M, K, N, 0, _stream())
                if rc != _lib.STAGE_ERR_SHAPE:
                    _lib.check(rc, "stage_gemm_nt_mask")
                    done = True
            if not done:
                _call("stage_gemm_nt", _ptr(dy), _ptr(gate), _ptr(wt), None, None, _ptr(dx), M, K, N, 0, _stream())
        dw = torch.empty_like(w2)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        wsb = lib.stage_gemm_tn_ws_bytes(M, N, K)
        ws = _workspace(wsb, x.device)
        done = False
        if use_mask:
            rc = lib.stage_gemm_tn_mask(_ptr(dy), _ptr(mask), _ptr(x), _ptr(dw), _ptr(db), M, N, K, _ptr(ws), wsb, _stream())
            if rc != _lib.STAGE_ERR_SHAPE:
                _lib.check(rc, "stage_gemm_tn_mask")
                done = True
        if not done:
            _call("stage_gemm_tn", _ptr(dy), _ptr(gate), _ptr(x), _ptr(dw), _ptr(db), M, N, K, _ptr(ws), wsb, _stream())
        dwv, db = _deliver(ctx.sinks, (dw.view(ctx.wshape), db))
        return dx, dwv, db, None


def linear(x, w, bias=None, relu: bool = False):
    return _Linear.apply(x, w, bias, relu)


class _InputLnLinear(torch.autograd.Function):
    """relu(linear(drop(LN(x)))) for an input that needs NO gradient (first layer of the input MLPs): forward = the LayerNorm and
    the Linear kernels; backward = weight gradient + the LayerNorm's gain / bias gradients reduced inside the dX GEMM's epilogue
    (stage_gemm_nt_lnparam) -- the (M, K) gradient of the LayerNorm output is never written."""
    @_on_device
    def forward(ctx, x, gamma, beta, w, bias, p: float, seed: int):
        x = _chk(x, "x")
        K = x.shape[-1]
        M = x.numel() // K
        gamma, beta = _chk(gamma, "gamma"), _chk(beta, "beta")
        w2, bias_c = _chk(w, "w").reshape(w.shape[0], -1), _chk(bias, "bias")
        N = w2.shape[0]
        y = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _call("stage_layernorm_fwd", _ptr(x), None, 0, None, _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), M, K, EPS_LN,
              float(p), int(seed), _stream())
        h = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        mask = torch.empty((N + 31) // 32, M, dtype=torch.int32, device=x.device)
        _call("stage_gemm_nt_mask", _ptr(y), None, _ptr(w2), _ptr(bias_c), _ptr(h), _ptr(mask), M, N, K, 1, _stream())
        ctx.save_for_backward(x, mean, rstd, y, w2, mask)
        ctx.p, ctx.seed, ctx.wshape, ctx.w_obj = float(p), int(seed), w.shape, w
        return h

    @_on_device
    def backward(ctx, dh):
        x, mean, rstd, y, w2, mask = ctx.saved_tensors
        N, K = w2.shape
        M = x.numel() // K
        dh = _chk(dh, "dh")
        lib = _lib.load()
        dw, db = torch.empty_like(w2), torch.empty(N, dtype=torch.float32, device=x.device)
        wsb = lib.stage_gemm_tn_ws_bytes(M, N, K)
        ws = _workspace(wsb, x.device)
        _call("stage_gemm_tn_mask", _ptr(dh), _ptr(mask), _ptr(y), _ptr(dw), _ptr(db), M, N, K, _ptr(ws), wsb, _stream())
        keep = None
        if ctx.p > 0.0:
            keep = torch.empty((K + 31) // 32, M, dtype=torch.int32, device=x.device)
            _call("stage_dropout_keepmask", ctx.p, ctx.seed, _ptr(keep), M, K, _stream())
        wt = _transposed_weight(ctx.w_obj, w2)          # (K, N)
        dgamma, dbeta = torch.empty(K, dtype=torch.float32, device=x.device), torch.empty(K, dtype=torch.float32, device=x.device)
        wsb2 = lib.stage_gemm_nt_lnparam_ws_bytes(M, K)
        ws2 = _workspace(wsb2, x.device)
        _call("stage_gemm_nt_lnparam", _ptr(dh), _ptr(mask), _ptr(wt), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(keep), ctx.p, _ptr(dgamma),
              _ptr(dbeta), M, K, N, _ptr(ws2), wsb2, _stream())
        return None, dgamma, dbeta, dw.view(ctx.wshape), db, None, None


def input_ln_linear_supported(x, w) -> bool:
    """The fused backward of ``input_ln_linear`` applies: fp32 storage, x needs no gradient, shapes the streaming kernels take."""
    if x.dtype != torch.float32 or x.requires_grad or not torch.is_grad_enabled():
        return False
    K = x.shape[-1]
    M, N = x.numel() // K, w.shape[0]
    lib = _lib.load()
    return bool(lib.stage_gemm_mask_supported(M, N, K)) and bool(lib.stage_gemm_nt_lnparam_supported(M, K, N)) and x.data_ptr() % 16 == 0


def input_ln_linear(x, gamma, beta, w, bias, p: float = 0.0, seed: int = 0):
    """relu(linear(drop(LayerNorm(x)))) for a feature tensor that needs no gradient (see _InputLnLinear)."""
    return _InputLnLinear.apply(x, gamma, beta, w, bias, p, seed)


# ---------------------------------------------------------------------------------------------------------------
# depthwise Conv1d along L
# ---------------------------------------------------------------------------------------------------------------
class _DWConv(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, w, bias):
        x = _act(x, "x")  # (M, L, D)
        M, L, D = x.shape
        w, bias = _chk(w, "w"), _chk(bias, "bias")
        k = w.shape[-1]
        y = torch.empty_like(x)
        _call("stage_dwconv_fwd" + _sfx(x), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), M, L, D, k, _stream())
        ctx.save_for_backward(x, w)
        return y

    @_on_device
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        M, L, D = x.shape
        k = w.shape[-1]
        dy = _act(dy, "dy", x)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(D, dtype=torch.float32, device=x.device)
        lib = _lib.load()
        wsb = lib.stage_dwconv_bwd_ws_bytes(D, k)
        ws = _workspace(wsb, x.device)
        _call("stage_dwconv_bwd" + _sfx(x), _ptr(dy), _ptr(x), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), M, L, D, k, _ptr(ws), wsb,
              _stream())
        return dx, dw, db


def dwconv(x, w, bias):
    return _DWConv.apply(x, w, bias)


# ---------------------------------------------------------------------------------------------------------------
# fused LayerNorm (+ residual, + dropout) -> depthwise Conv1d (the LayerNorm output is never materialised)
# ---------------------------------------------------------------------------------------------------------------
def ln_dwconv_supported(D: int, k: int, dtype=torch.float32) -> bool:
    d4 = D // 4
    return dtype in (torch.float32, _BF16) and D % 4 == 0 and 4 <= d4 <= 64 and (d4 & (d4 - 1)) == 0 and 1 <= k <= 9 and k % 2 == 1


class _LnDwConv(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, res, gamma, beta, w, bias, res_period: int, p: float, seed: int):
        x = _act(x, "x")  # (M, L, D)
        M, L, D = x.shape
        if res is not None and res_period > 0 and res.dtype != x.dtype:
            res = res.to(x.dtype)          # the (L, D) position table follows the storage type
        res_c = None if res is None else _act(res, "res", x)
        ctx.sinks = _sinks((gamma, beta, w, bias))
        gamma, beta, w, bias = _chk(gamma, "gamma"), _chk(beta, "beta"), _chk(w, "w"), _chk(bias, "bias")
        k = w.shape[-1]
        h = torch.empty_like(x)
        mean = torch.empty(M * L, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        s = torch.empty_like(x) if res_c is not None else None
        _call("stage_ln_dwconv_fwd" + _sfx(x), _ptr(x), _ptr(res_c), int(res_period), _ptr(s), _ptr(gamma), _ptr(beta), _ptr(w),
              _ptr(bias), _ptr(h), _ptr(mean), _ptr(rstd), M, L, D, k, EPS_LN, float(p), int(seed), _stream())
        xin = s if res_c is not None else x
        ctx.save_for_backward(xin, mean, rstd, gamma, beta, w)
        ctx.set_materialize_grads(False)   # see _LayerNorm
        ctx.p, ctx.seed = float(p), int(seed)
        ctx.has_res = res_c is not None
        ctx.res_full = res_c is not None and res_period == 0
        if s is None:
            s = h.new_empty(0)
        return h, s

    @_on_device
    def backward(ctx, dh, dsum):
        xin, mean, rstd, gamma, beta, w = ctx.saved_tensors
        M, L, D = xin.shape
        k = w.shape[-1]
        if dh is None:
            dh = torch.zeros_like(xin)
        dh = _act(dh, "dh", xin)
        x_needs = ctx.needs_input_grad[0]
        res_needs = ctx.needs_input_grad[1] and ctx.res_full
        dx = torch.empty_like(xin) if (x_needs or res_needs) else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        dw = torch.empty_like(w)
        db = torch.empty(D, dtype=torch.float32, device=xin.device)
        lib = _lib.load()
        wsb = lib.stage_ln_dwconv_bwd_ws_bytes(D, k)
        ws = _workspace(wsb, xin.device)
        dadd = None  # gradient arriving through the exported sum (next residual branch), fused into the dx store
        if dx is not None and ctx.has_res and dsum is not None and dsum.numel() == dx.numel():
            dadd = _act(dsum, "dsum", xin)
        _call("stage_ln_dwconv_bwd" + _sfx(xin), _ptr(dh), _ptr(xin), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(w),
              _ptr(dx), _ptr(dadd), _ptr(dgamma), _ptr(dbeta), _ptr(dw), _ptr(db), M, L, D, k, ctx.p, ctx.seed,
              _ptr(ws), wsb, _stream())
        dgamma, dbeta, dw, db = _deliver(ctx.sinks, (dgamma, dbeta, dw, db))
        return (dx if x_needs else None), (dx if res_needs else None), dgamma, dbeta, dw, db, None, None, None


def ln_dwconv(x, gamma, beta, w, bias, p: float = 0.0, seed: int = 0, res=None, res_period: int = 0):
    """h = dwconv(drop(LN(x + res))); returns (h, x + res or None)."""
    h, s = _LnDwConv.apply(x, res, gamma, beta, w, bias, res_period, p, seed)
    return h, (s if res is not None else None)


# ---------------------------------------------------------------------------------------------------------------
# L2 normalisation of raw features (no gradient needed: inputs are data) -- model/stage.py:256
# ---------------------------------------------------------------------------------------------------------------
def l2norm(x: torch.Tensor, p: float = 0.0, seed: int = 0) -> torch.Tensor:
    x = _act(x, "x")
    K = x.shape[-1]
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _call("stage_l2norm_fwd" + _sfx(x), _ptr(x), _ptr(y), None, x.numel() // K, K, EPS_L2, float(p), int(seed), _stream())
    return y


# ---------------------------------------------------------------------------------------------------------------
# K1: StructuredAttention
# ---------------------------------------------------------------------------------------------------------------
import os as _os

_K1_BWD_UNFUSED = _os.environ.get("STAGE_K1_BWD_UNFUSED") is not None   # developer switch (cross-check in the tests)


def _fold_dsn(dS, dSn, Sn, scale):
    """A gradient that arrives on the NORMALISED scores (the model never sends one; the public operator may be used that way,
    model/context_query_attention.py:61 is differentiable in the reference): S_ = softmax(scale * S) * mask, so it reaches the
    raw scores as scale * S_ * (dS_ - <dS_, S_>) on the unmasked entries (masked entries have S_ = 0 and a zero gradient, fully
    masked rows are zeroed by the mask).  A few plumbing-sized ATen ops on a path the training step does not take."""
    if dSn is None:
        return dS
    g = scale * Sn * (dSn - (dSn * Sn).sum(-1, keepdim=True))
    return g if dS is None else dS + g


class _StrAttn(torch.autograd.Function):
    """The fast kernels (D = 128, Lr <= 64).  fp32, or bf16 storage: Q, A and dA are bf16 and are converted as the kernels load
    / store them; the context side (N*NA*Lqa rows: small) is normalised in fp32, the score maps are fp32."""

    @_on_device
    def forward(ctx, C, Q, c_mask, q_mask, scale: float, p: float, seed_c: int, seed_q: int):
        Q = _act(Q, "Q")
        C = _act(C, "C", Q)                                  # (N, NA, Lqa, D), (N, Li, Lr, D)
        bf = Q.dtype == _BF16
        c_mask, q_mask = _chk(c_mask, "c_mask"), _chk(q_mask, "q_mask")
        N, NA, Lqa, D = C.shape
        _, Li, Lr, _ = Q.shape
        Cf = C.float() if bf else C
        Cn = torch.empty_like(Cf)
        _call("stage_l2norm_fwd", _ptr(Cf), _ptr(Cn), None, N * NA * Lqa, D, EPS_L2, float(p), int(seed_c), _stream())
        A = torch.empty(N, NA, Li, Lqa, D, dtype=Q.dtype, device=C.device)
        S = torch.empty(N, NA, Li, Lqa, Lr, dtype=torch.float32, device=C.device)
        Sn = torch.empty_like(S)
        _call("stage_str_attn_fwd" + _sfx(Q), _ptr(Cn), _ptr(Q), _ptr(c_mask), _ptr(q_mask), _ptr(A), _ptr(S), _ptr(Sn), N, NA, Li,
              Lqa, Lr, D, float(scale), float(p), int(seed_q), _stream())
        ctx.save_for_backward(Cf, Q, Cn, Sn, q_mask)
        ctx.cfg = (float(scale), float(p), int(seed_c), int(seed_q))
        # raw S only receives a gradient when the supervised attention loss is on: without this autograd hands the
        # backward a materialised zero tensor (a 77 / 192 MB fill plus one more read in the dS kernel)
        ctx.set_materialize_grads(False)
        return A, S, Sn

    @_on_device
    def backward(ctx, dA, dS, dSn):
        Cf, Q, Cn, Sn, q_mask = ctx.saved_tensors
        scale, p, seed_c, seed_q = ctx.cfg
        dS = _fold_dsn(dS, dSn, Sn, scale)
        bf = Q.dtype == _BF16
        N, NA, Lqa, D = Cf.shape
        _, Li, Lr, _ = Q.shape
        dA = _act(dA, "dA", Q) if dA is not None else torch.zeros(N, NA, Li, Lqa, D, dtype=Q.dtype, device=Q.device)
        dS_ext = _chk(dS, "dS") if dS is not None else None
        Qn = torch.empty_like(Q)
        _call("stage_l2norm_fwd" + _sfx(Q), _ptr(Q), _ptr(Qn), None, N * Li * Lr, D, EPS_L2, p, seed_q, _stream())
        dQ = torch.empty(Q.shape, dtype=torch.float32, device=Q.device)   # receives the value-path gradient, then the normalised-path one on top
        dQn = torch.empty_like(dQ)
        dCn = torch.empty_like(Cf)
        lib = _lib.load()
        rc = _lib.STAGE_ERR_SHAPE
        if not _K1_BWD_UNFUSED:
            # one pass over dA, dS stays on chip (D = 128, even Lr); other shapes take the three-kernel path below
            wsb = lib.stage_str_attn_bwd_fused_ws_bytes(N, NA, Li, Lqa, D)
            ws = _workspace(wsb, Q.device)
            fn = lib.stage_str_attn_bwd_fused_bf16 if bf else lib.stage_str_attn_bwd_fused
            rc = fn(_ptr(dA), _ptr(dS_ext), _ptr(Cn), _ptr(Q), _ptr(Qn), _ptr(Sn), _ptr(q_mask), _ptr(dQ), _ptr(dQn), _ptr(dCn),
                    N, NA, Li, Lqa, Lr, D, scale, _ptr(ws), wsb, _stream())
            if rc != _lib.STAGE_ERR_SHAPE:
                _lib.check(rc, "stage_str_attn_bwd_fused")
        if rc == _lib.STAGE_ERR_SHAPE:
            # three-kernel path (fp32 operands; in the bf16 mode the casts are plumbing for the shapes the fused kernel rejects)
            dAf, Qf, Qnf = (dA.float(), Q.float(), Qn.float()) if bf else (dA, Q, Qn)
            dS_out = torch.empty_like(Sn)
            wsb = lib.stage_str_attn_bwd_ws_bytes(N, NA, Lqa, D)
            ws = _workspace(wsb, Q.device)
            _call("stage_str_attn_bwd", _ptr(dAf), _ptr(dS_ext), _ptr(Cn), _ptr(Qf), _ptr(Qnf), _ptr(Sn), _ptr(dS_out),
                  _ptr(dQ), _ptr(dQn), _ptr(dCn), N, NA, Li, Lqa, Lr, D, scale, _ptr(ws), wsb, _stream())
        dC = torch.empty_like(Cf)
        _call("stage_l2norm_bwd", _ptr(dCn), _ptr(Cf), _ptr(dC), N * NA * Lqa, D, EPS_L2, p, seed_c, 0, _stream())
        if bf:
            # one pass: dQ (bf16) = value-path gradient (fp32) + the normalisation's backward of dQn (fp32) at the bf16 rows of Q
            dQb = torch.empty_like(Q)
            _call("stage_l2norm_bwd_mixed_bf16", _ptr(dQn), _ptr(Q), _ptr(dQ), _ptr(dQb), N * Li * Lr, D, EPS_L2, p, seed_q, _stream())
            return dC.to(_BF16), dQb, None, None, None, None, None, None
        _call("stage_l2norm_bwd", _ptr(dQn), _ptr(Q), _ptr(dQ), N * Li * Lr, D, EPS_L2, p, seed_q, 1, _stream())
        return dC, dQ, None, None, None, None, None, None


class _StrAttnLong(torch.autograd.Function):
    """StructuredAttention for long region rows (any Lr) and / or bf16 storage (csrc/str_attn_long.hip).  C and Q are fp32 or
    bf16 (both the same); A comes back in that type, the score maps in fp32.  The L2 normalisation (+ dropout) of both
    sides runs in the fp32 kernels on an fp32 view (dtype casts are plumbing); the normalised operands are rounded to the
    storage type before the attention, like every other activation of a bf16 pipeline."""

    @_on_device
    def forward(ctx, C, Q, c_mask, q_mask, scale: float, p: float, seed_c: int, seed_q: int):
        dt = C.dtype
        if dt not in (torch.float32, torch.bfloat16) or Q.dtype != dt:
            raise TypeError("structured_attention_long: C and Q must both be float32 or bfloat16")
        C, Q = _chk(C, "C", dt), _chk(Q, "Q", dt)
        c_mask, q_mask = _chk(c_mask, "c_mask"), _chk(q_mask, "q_mask")
        N, NA, Lqa, D = C.shape
        _, Li, Lr, _ = Q.shape
        # normalised operands in the storage type (bf16: the 16-bit kernels, no fp32 copies of the big region tensor)
        Cn, Qn = torch.empty_like(C), torch.empty_like(Q)
        _call("stage_l2norm_fwd" + _sfx(C), _ptr(C), _ptr(Cn), None, N * NA * Lqa, D, EPS_L2, float(p), int(seed_c), _stream())
        _call("stage_l2norm_fwd" + _sfx(Q), _ptr(Q), _ptr(Qn), None, N * Li * Lr, D, EPS_L2, float(p), int(seed_q), _stream())
        A = torch.empty(N, NA, Li, Lqa, D, dtype=dt, device=C.device)
        S = torch.empty(N, NA, Li, Lqa, Lr, dtype=torch.float32, device=C.device)
        Sn = torch.empty_like(S)
        _call("stage_str_attn_long_fwd", _ptr(Cn), _ptr(Q), _ptr(Qn), _ptr(c_mask), _ptr(q_mask), _ptr(A), _ptr(S), _ptr(Sn),
              N, NA, Li, Lqa, Lr, D, float(scale), int(dt == torch.bfloat16), _stream())
        ctx.save_for_backward(C, Q, Cn, Qn, Sn, A, q_mask)
        ctx.cfg = (float(scale), float(p), int(seed_c), int(seed_q))
        ctx.set_materialize_grads(False)
        return A, S, Sn

    @_on_device
    def backward(ctx, dA, dS, dSn):
        C, Q, Cn, Qn, Sn, A, q_mask = ctx.saved_tensors
        scale, p, seed_c, seed_q = ctx.cfg
        dS = _fold_dsn(dS, dSn, Sn, scale)
        dt = C.dtype
        N, NA, Lqa, D = C.shape
        _, Li, Lr, _ = Q.shape
        dA = _chk(dA, "dA", dt) if dA is not None else torch.zeros_like(A)
        dS_ext = _chk(dS, "dS") if dS is not None else None
        dS_ws = torch.empty_like(Sn)
        dQ = torch.empty(Q.shape, dtype=torch.float32, device=Q.device)
        dQn, dCn = torch.empty_like(dQ), torch.empty(C.shape, dtype=torch.float32, device=C.device)
        lib = _lib.load()
        # region blocks behind a frame's last valid region are exact zeros and skipped (unless dS_ext reaches into them)
        wsb = lib.stage_str_attn_long_bwd_qm_ws_bytes(N, NA, Li, Lqa, D)
        ws = _workspace(wsb, C.device)
        _call("stage_str_attn_long_bwd_qm", _ptr(dA), _ptr(A), _ptr(dS_ext), _ptr(Cn), _ptr(Q), _ptr(Qn), _ptr(Sn), _ptr(q_mask),
              _ptr(dS_ws), _ptr(dQ), _ptr(dQn), _ptr(dCn), N, NA, Li, Lqa, Lr, D, scale, int(dt == torch.bfloat16), _ptr(ws), wsb,
              _stream())
        if dt == _BF16:
            dQb, dCb = torch.empty_like(Q), torch.empty_like(C)
            _call("stage_l2norm_bwd_mixed_bf16", _ptr(dQn), _ptr(Q), _ptr(dQ), _ptr(dQb), N * Li * Lr, D, EPS_L2, p, seed_q, _stream())
            _call("stage_l2norm_bwd_mixed_bf16", _ptr(dCn), _ptr(C), None, _ptr(dCb), N * NA * Lqa, D, EPS_L2, p, seed_c, _stream())
            return dCb, dQb, None, None, None, None, None, None
        _call("stage_l2norm_bwd", _ptr(dQn), _ptr(Q), _ptr(dQ), N * Li * Lr, D, EPS_L2, p, seed_q, 1, _stream())
        dC = torch.empty_like(C)
        _call("stage_l2norm_bwd", _ptr(dCn), _ptr(C), _ptr(dC), N * NA * Lqa, D, EPS_L2, p, seed_c, 0, _stream())
        return dC, dQ, None, None, None, None, None, None


def structured_attention_long(C, Q, c_mask, q_mask, scale: float, p: float = 0.0, seed_c: int = 0, seed_q: int = 0):
    """As ``structured_attention`` for any Lr and D in {16,32,64,128,256}; C, Q fp32 or bf16 (A in the same type)."""
    return _StrAttnLong.apply(C, Q, c_mask, q_mask, scale, p, seed_c, seed_q)


def structured_attention(C, Q, c_mask, q_mask, scale: float, p: float = 0.0, seed_c: int = 0, seed_q: int = 0):
    """C (N,NA,Lqa,D), Q (N,Li,Lr,D), c_mask (N,NA,Lqa), q_mask (N,Li,Lr) -> A (N,NA,Li,Lqa,D), raw S, normalised S."""
    # long rows (512 subtitle words) and bf16 shapes outside the fast kernels (D != 128): csrc/str_attn_long.hip
    if Q.shape[2] > 64 or (C.dtype == torch.bfloat16 and C.shape[-1] != 128):
        return _StrAttnLong.apply(C, Q, c_mask, q_mask, scale, p, seed_c, seed_q)
    return _StrAttn.apply(C, Q, c_mask, q_mask, scale, p, seed_c, seed_q)


# ---------------------------------------------------------------------------------------------------------------
# masked max over a sequence axis (optionally windowed)
# ---------------------------------------------------------------------------------------------------------------
class _MaskedMax(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, mask, window):
        x, mask = _act(x, "x"), _chk(mask, "mask")  # (R, L, D), (R, L)
        R, L, D = x.shape
        win = None if window is None else _chk(window, "window", torch.int32)
        out = torch.empty(R, D, dtype=x.dtype, device=x.device)
        idx = torch.empty(R, D, dtype=torch.int32, device=x.device)
        _call("stage_masked_max_fwd" + _sfx(x), _ptr(x), _ptr(mask), _ptr(win), _ptr(out), _ptr(idx), R, L, D, _stream())
        ctx.save_for_backward(idx, mask)
        ctx.shape = (R, L, D)
        return out

    @_on_device
    def backward(ctx, dout):
        idx, mask = ctx.saved_tensors
        R, L, D = ctx.shape
        dout = _act(dout, "dout")
        dx = torch.empty(R, L, D, dtype=dout.dtype, device=dout.device)
        _call("stage_masked_max_bwd" + _sfx(dout), _ptr(dout), _ptr(idx), _ptr(mask), _ptr(dx), R, L, D, 0, _stream())
        return dx, None, None


def masked_max(x, mask, window=None):
    return _MaskedMax.apply(x, mask, window)


class _LnMaskedMax(torch.autograd.Function):
    """max over the sequence axis of mask_logits(LayerNorm(x + res)) in one pass (csrc/rowops.hip: ln_mm_fwd_kernel)."""
    @_on_device
    def forward(ctx, x, res, gamma, beta, mask):
        x, mask = _act(x, "x"), _chk(mask, "mask")          # (R, L, K), (R, L)
        R, L, K = x.shape
        res_c = None if res is None else _act(res, "res", x)
        gamma, beta = _chk(gamma, "gamma"), _chk(beta, "beta")
        out = torch.empty(R, K, dtype=x.dtype, device=x.device)
        idx = torch.empty(R, K, dtype=torch.int32, device=x.device)
        mean = torch.empty(R * L, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        s = torch.empty_like(x) if res_c is not None else None
        _call("stage_ln_masked_max_fwd" + _sfx(x), _ptr(x), _ptr(res_c), _ptr(s), _ptr(gamma), _ptr(beta), _ptr(mask), _ptr(out), _ptr(idx),
              _ptr(mean), _ptr(rstd), R, L, K, EPS_LN, _stream())
        ctx.save_for_backward(s if res_c is not None else x, mean, rstd, gamma, idx, mask)
        ctx.shape = (R, L, K)
        ctx.has_res = res_c is not None
        return out

    @_on_device
    def backward(ctx, dout):
        xin, mean, rstd, gamma, idx, mask = ctx.saved_tensors
        R, L, K = ctx.shape
        dout = _act(dout, "dout", xin)
        dx = torch.empty_like(xin)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        lib = _lib.load()
        wsb = lib.stage_ln_bwd_ws_bytes(K)
        ws = _workspace(wsb, xin.device)
        _call("stage_ln_masked_max_bwd" + _sfx(xin), _ptr(dout), _ptr(idx), _ptr(mask), _ptr(xin), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx),
              _ptr(dgamma), _ptr(dbeta), R, L, K, _ptr(ws), wsb, _stream())
        return (dx if ctx.needs_input_grad[0] else None), (dx if (ctx.has_res and ctx.needs_input_grad[1]) else None), dgamma, dbeta, None


def ln_masked_max_supported(x, L: int, K: int) -> bool:
    return bool(_lib.load().stage_ln_masked_max_supported(int(L), int(K)))


def ln_masked_max(x, res, gamma, beta, mask):
    """(R, L, K) x (+ res), (R, L) mask -> (R, K): masked max over L of LayerNorm(x + res); the normalised tensor is never materialised."""
    return _LnMaskedMax.apply(x, res, gamma, beta, mask)


# ---------------------------------------------------------------------------------------------------------------
# multi-head attention core
# ---------------------------------------------------------------------------------------------------------------
class _MHACore(torch.autograd.Function):
    @_on_device
    def forward(ctx, q, k, v, mask, nh: int, p: float, seed: int):
        q = _act(q, "q")
        k, v, mask = _act(k, "k", q), _act(v, "v", q), _chk(mask, "mask")
        M, L, D = q.shape
        out = torch.empty_like(q)
        if q.dtype == _BF16:      # bf16 storage: the matrix-core kernels only (probabilities recomputed, never stored)
            _call("stage_mha_core_fwd_bf16", _ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(out), M, L, D, nh, float(p), int(seed),
                  _stream())
            ctx.save_for_backward(q, k, v, None, mask)
            ctx.cfg = (nh, float(p), int(seed))
            return out
        # the matrix-core kernels recompute the probabilities in the backward: no (M, nh, L, L) tensor (614 MB for the
        # classifier encoder of the full config); only the scalar fallback shapes keep it
        probs = None
        if _os.environ.get("STAGE_MHA_SCALAR") is not None or not _lib.load().stage_mha_core_recomputes(L, D, nh):
            probs = torch.empty(M, nh, L, L, dtype=torch.float32, device=q.device)
        _call("stage_mha_core_fwd", _ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(out), _ptr(probs), M, L, D, nh, float(p),
              int(seed), _stream())
        ctx.save_for_backward(q, k, v, probs, mask)
        ctx.cfg = (nh, float(p), int(seed))
        return out

    @_on_device
    def backward(ctx, dout):
        q, k, v, probs, mask = ctx.saved_tensors
        nh, p, seed = ctx.cfg
        M, L, D = q.shape
        dout = _act(dout, "dout", q)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        if q.dtype == _BF16:
            _call("stage_mha_core_bwd_bf16", _ptr(dout), _ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(dq), _ptr(dk), _ptr(dv),
                  M, L, D, nh, p, seed, _stream())
            return dq, dk, dv, None, None, None, None
        _call("stage_mha_core_bwd", _ptr(dout), _ptr(q), _ptr(k), _ptr(v), _ptr(probs), _ptr(mask), _ptr(dq), _ptr(dk),
              _ptr(dv), M, L, D, nh, p, seed, _stream())
        return dq, dk, dv, None, None, None, None


def mha_core(q, k, v, mask, nh: int, p: float = 0.0, seed: int = 0):
    return _MHACore.apply(q, k, v, mask, nh, p, seed)


class _MHACoreQKV(torch.autograd.Function):
    """The same core on the FUSED projections: qkv (M, L, 3D) = Linear(D -> 3D)(x) with the weight [W_q; W_k; W_v]; the backward
    writes dq | dk | dv into one (M, L, 3D) tensor, the output gradient of that Linear (stage_mha_core_qkv_*)."""

    @_on_device
    def forward(ctx, qkv, mask, nh: int, p: float, seed: int):
        qkv = _act(qkv, "qkv")
        mask = _chk(mask, "mask")
        M, L, D3 = qkv.shape
        D = D3 // 3
        out = torch.empty(M, L, D, dtype=qkv.dtype, device=qkv.device)
        _call("stage_mha_core_qkv_fwd", _ptr(qkv), _ptr(mask), _ptr(out), M, L, D, nh, float(p), int(seed), int(qkv.dtype == _BF16),
              _stream())
        ctx.save_for_backward(qkv, mask)
        ctx.cfg = (nh, float(p), int(seed), D)
        return out

    @_on_device
    def backward(ctx, dout):
        qkv, mask = ctx.saved_tensors
        nh, p, seed, D = ctx.cfg
        M, L, _ = qkv.shape
        dout = _act(dout, "dout", qkv)
        dqkv = torch.empty_like(qkv)
        _call("stage_mha_core_qkv_bwd", _ptr(dout), _ptr(qkv), _ptr(mask), _ptr(dqkv), M, L, D, nh, p, seed, int(qkv.dtype == _BF16),
              _stream())
        return dqkv, None, None, None, None


def mha_core_qkv_supported(L: int, D: int, nh: int) -> bool:
    return _os.environ.get("STAGE_MHA_SCALAR") is None and _os.environ.get("STAGE_NO_FUSED_QKV") is None and \
        bool(_lib.load().stage_mha_core_recomputes(L, D, nh)) and D % 8 == 0


def mha_core_qkv(qkv, mask, nh: int, p: float = 0.0, seed: int = 0):
    return _MHACoreQKV.apply(qkv, mask, nh, p, seed)
