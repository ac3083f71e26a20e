"""K-groups: one ``torch.autograd.Function`` per fused-op GROUP of STAGE, each backed by one forward and one backward entry point of
``libstage_hip.so`` that sequences the group's kernels on the C side (csrc/groups.hip; SURVEY.md section 8b, last bullet).

The per-op path (``tvqaplus_amd.ops``: one Function and ~4 ``torch.empty`` per kernel) makes ~140 library calls per training step
from Python; here the interpreter sees one call per group -- input MLP, encoder block, QA<->context attention + down-projection,
two-stream fusion, temporal head, span scores / proposal pooling + classifier, the two losses -- ~31 calls per step.  The kernels
and their arguments are the per-op path's except for the fused ``[a, b, a*b]`` kernels (``tests/test_hip_groups.py`` holds the two
paths equal, dropout on): what changes is that the step no longer depends on how fast the host can issue launches.

Memory: the caller (this module) owns everything.  ``arena`` = one buffer per group call that the C side carves into what the
backward needs; ``tmp`` = backward-only scratch, freed when the backward returns (stream-ordered, like every torch temporary).
A group that does not take a shape raises ``Unsupported`` BEFORE launching anything; STAGE then runs that group per-op.
fp32 storage only (the bf16 storage mode stays on the per-op path).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from .ops import _chk, _deliver, _on_device, _sinks, _stream

_U8 = torch.uint8


class Unsupported(Exception):
    """The C side declined the shape (STAGE_ERR_SHAPE) before launching anything."""


def _ptrs(ts: Sequence[Optional[torch.Tensor]]):
    return (ctypes.c_void_p * max(1, len(ts)))(*[None if t is None else t.data_ptr() for t in ts])


def _u64(seeds: Sequence[int]):
    return (ctypes.c_ulonglong * max(1, len(seeds)))(*[int(s) for s in seeds])


def _flags():
    return (ctypes.c_int * 16)()


def _rc(rc: int, what: str):
    if rc == _lib.STAGE_ERR_SHAPE:
        raise Unsupported(what)
    if rc != 0:
        _lib.check(rc, what)


_SIZES = {}


def _size(fn_name: str, *dims) -> int:
    """Arena / scratch sizes depend on the dimensions only: one C call per distinct shape."""
    key = (fn_name,) + dims
    v = _SIZES.get(key)
    if v is None:
        v = _SIZES[key] = int(getattr(_lib.load(), fn_name)(*dims))
    return v


_QUANT = 32 << 20


def _buf(nbytes: int, device) -> torch.Tensor:
    # large buffers in 32 MB steps: with ragged token rows the row count changes from batch to batch, and a caching allocator
    # that sees a new size every step keeps going back to hipMalloc
    n = max(int(nbytes), 256)
    if n > _QUANT:
        n = (n + _QUANT - 1) // _QUANT * _QUANT
    return torch.empty(n, dtype=_U8, device=device)


def _rows(rows: int, D: int, device) -> torch.Tensor:
    """(rows, D) fp32, carved from an allocation whose row count is rounded up (same reason)."""
    cap = (rows + 16383) // 16384 * 16384 if rows > 16384 else rows
    return torch.empty(cap, D, dtype=torch.float32, device=device)[:rows]


def _grad_views(params: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """One allocation for all parameter gradients of a group call; 256-byte aligned slices shaped like the parameters."""
    offs, total = [], 0
    for w in params:
        offs.append(total)
        total += (w.numel() + 63) // 64 * 64
    flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
    return [flat[o: o + w.numel()].view(w.shape) for o, w in zip(offs, params)]


def _params(params) -> List[torch.Tensor]:
    return [_chk(w, "param") for w in params]


# ---------------------------------------------------------------------------------------------------------------
# Shared modules: one gradient per parameter leaves the graph
# ---------------------------------------------------------------------------------------------------------------
# The input embedding, the first encoder and the down-projection are applied to two or three streams per step
# (model/stage.py:226-269).  With one autograd node per application, autograd sums their parameter gradients with one
# ``add`` kernel per parameter and extra use (31 launches per step).  ``gate(params)`` hands out aliases of the parameters for one
# step; a group that is called with such an alias puts its gradient into the gate's sink (the first contribution is kept as is, later
# ones are added with ONE multi-tensor add per group call) and returns None for it; the gate's own backward -- which autograd
# runs after every consumer, whatever they returned -- hands the totals to the parameters.
class _Sink:
    """The contributions arrive OUTSIDE autograd's graph edges, so autograd does not order them: with branch streams (stage.py:
    use_streams) the three applications of a shared module run their backward on three streams.  Every access to ``acc`` therefore
    waits for the event of the previous one on the stream it runs on, records its own, and tells the allocator that the accumulators
    are in use on that stream (an unordered first version failed one model test in ten)."""

    def __init__(self, n: int):
        self.acc: List[Optional[torch.Tensor]] = [None] * n
        self.ev = None

    def _enter(self, tensors):
        if not tensors or not tensors[0].is_cuda:
            return None
        cur = torch.cuda.current_stream(tensors[0].device)
        if self.ev is not None:
            cur.wait_event(self.ev)
        return cur

    def _leave(self, cur, tensors):
        if cur is None:
            return
        for t in tensors:
            if t is not None:
                t.record_stream(cur)
        self.ev = torch.cuda.Event()
        self.ev.record(cur)

    def add(self, idx: Sequence[int], grads: Sequence[torch.Tensor]):
        cur = self._enter(list(grads))
        accs, news = [], []
        for i, g in zip(idx, grads):
            if self.acc[i] is None:
                self.acc[i] = g
            else:
                accs.append(self.acc[i])
                news.append(g)
        if accs:
            torch._foreach_add_(accs, news)
        self._leave(cur, [self.acc[i] for i in idx])


class _ParamGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sink, *params):
        ctx.sink = sink
        ctx.set_materialize_grads(False)
        outs = tuple(w.detach() for w in params)
        for i, o in enumerate(outs):
            o._stage_sink = (sink, i)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        sink = ctx.sink
        live = [a for a in sink.acc if a is not None]
        cur = sink._enter(live)
        res = []
        for a, g in zip(sink.acc, gouts):         # g: whatever reached the alias through ordinary autograd (a per-kernel fallback)
            res.append(a if g is None else (g if a is None else a + g))
        sink._leave(cur, live)
        sink.acc = [None] * len(res)
        sink.ev = None
        return (None,) + tuple(res)


def gate(params: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Aliases of ``params`` for ONE training step (see above); use them wherever the parameters go into a group call."""
    return list(_ParamGate.apply(_Sink(len(params)), *params))


# ---------------------------------------------------------------------------------------------------------------
# G1 input MLP
# ---------------------------------------------------------------------------------------------------------------
class _InputMLP(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, l2: int, p: float, seeds, *params):
        x = _chk(x, "x")
        ctx.sinks = _sinks(params)
        params = _params(params)
        K0, H, D = x.shape[-1], params[2].shape[0], params[6].shape[0]
        M = x.numel() // K0
        lib = _lib.load()
        ab = _size("stage_grp_input_mlp_arena_bytes", M, K0, H, D, int(l2))
        arena = _buf(ab, x.device)
        out = torch.empty(x.shape[:-1] + (D,), dtype=torch.float32, device=x.device)
        flags = _flags()
        _rc(lib.stage_grp_input_mlp_fwd(x.data_ptr(), _ptrs(params), out.data_ptr(), arena.data_ptr(), ab, flags, M, K0, H, D, int(l2),
                                        float(p), _u64(seeds), _stream()), "stage_grp_input_mlp_fwd")
        ctx.save_for_backward(x, arena, *params)
        ctx.cfg = (M, K0, H, D, int(l2), float(p), tuple(seeds), flags, ab)
        return out

    @_on_device
    def backward(ctx, dout):
        x, arena, *params = ctx.saved_tensors
        M, K0, H, D, l2, p, seeds, flags, ab = ctx.cfg
        dout = _chk(dout, "dout")
        lib = _lib.load()
        grads = _grad_views(params)
        tb = _size("stage_grp_input_mlp_bwd_tmp_bytes", M, K0, H, D)
        tmp = _buf(tb, x.device)
        _rc(lib.stage_grp_input_mlp_bwd(dout.data_ptr(), x.data_ptr(), _ptrs(params), _ptrs(grads), arena.data_ptr(), ab, flags,
                                        tmp.data_ptr(), tb, M, K0, H, D, l2, p, _u64(seeds), _stream()), "stage_grp_input_mlp_bwd")
        return (None, None, None, None) + _deliver(ctx.sinks, grads)


def input_mlp(x, l2: bool, p: float, seeds, params):
    """x (..., K0) features -> (..., D).  params: ln0.w ln0.b fc1.w fc1.b ln1.w ln1.b fc2.w fc2.b ln2.w ln2.b."""
    return _InputMLP.apply(x, int(bool(l2)), p, tuple(seeds), *params)


# ---------------------------------------------------------------------------------------------------------------
# G2 encoder block (no self-attention)
# ---------------------------------------------------------------------------------------------------------------
class _Encoder(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, pe, pool_mask, k: int, p: float, seeds, *params):
        x = _chk(x, "x")                       # (M, L, D)
        pe = _chk(pe, "pe")
        ctx.sinks = _sinks(params)
        params = _params(params)
        M, L, D = x.shape
        n_conv = (len(params) - 2) // 6
        pooled = pool_mask is not None
        pm = _chk(pool_mask, "pool_mask") if pooled else None
        lib = _lib.load()
        ab = _size("stage_grp_encoder_arena_bytes", M, L, D, n_conv, int(pooled))
        arena = _buf(ab, x.device)
        out = torch.empty((M, D) if pooled else (M, L, D), dtype=torch.float32, device=x.device)
        flags = _flags()
        _rc(lib.stage_grp_encoder_fwd(x.data_ptr(), pe.data_ptr(), None if pm is None else pm.data_ptr(), _ptrs(params), out.data_ptr(),
                                      arena.data_ptr(), ab, flags, M, L, D, n_conv, int(k), float(p), _u64(seeds), _stream()),
            "stage_grp_encoder_fwd")
        ctx.save_for_backward(x, pm, arena, *params)
        ctx.cfg = (M, L, D, n_conv, int(k), float(p), tuple(seeds), flags, ab, pooled)
        return out

    @_on_device
    def backward(ctx, dout):
        x, pm, arena, *params = ctx.saved_tensors
        M, L, D, n_conv, k, p, seeds, flags, ab, pooled = ctx.cfg
        dout = _chk(dout, "dout")
        lib = _lib.load()
        grads = _grad_views(params)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        tb = _size("stage_grp_encoder_bwd_tmp_bytes", M, L, D, k, int(pooled))
        tmp = _buf(tb, x.device)
        _rc(lib.stage_grp_encoder_bwd(dout.data_ptr(), x.data_ptr(), None if pm is None else pm.data_ptr(), _ptrs(params), _ptrs(grads),
                                      None if dx is None else dx.data_ptr(), arena.data_ptr(), ab, flags, tmp.data_ptr(), tb, M, L, D,
                                      n_conv, k, p, _u64(seeds), _stream()), "stage_grp_encoder_bwd")
        return (dx, None, None, None, None, None) + _deliver(ctx.sinks, grads)


def encoder_block(x, pe, pool_mask, k: int, p: float, seeds, params):
    """x (M, L, D) -> (M, L, D), or (M, D) = masked max over L when ``pool_mask`` (M, L) is given.
    params: per conv (ln.w ln.b dw.w dw.b pw.w pw.b), then final_ln.w final_ln.b."""
    return _Encoder.apply(x, pe, pool_mask, k, p, tuple(seeds), *params)


# ---------------------------------------------------------------------------------------------------------------
# G3 QA <-> context attention + down-projection
# ---------------------------------------------------------------------------------------------------------------
class _QaCtx(torch.autograd.Function):
    @_on_device
    def forward(ctx, qa, cx, qa_mask, cx_mask, scale: float, p: float, seeds, *params):
        qa, cx = _chk(qa, "qa"), _chk(cx, "ctx")                # (N, NA, Lqa, D), (N, Li, Lr, D)
        qa_mask, cx_mask = _chk(qa_mask, "qa_mask"), _chk(cx_mask, "ctx_mask")
        ctx.sinks = _sinks(params)
        params = _params(params)
        N, NA, Lqa, D = qa.shape
        _, Li, Lr, _ = cx.shape
        lib = _lib.load()
        ab = _size("stage_grp_qa_ctx_arena_bytes", N, NA, Li, Lqa, D)
        arena = _buf(ab, qa.device)
        mixed = torch.empty(N, NA, Li, Lqa, D, dtype=torch.float32, device=qa.device)
        S = torch.empty(N, NA, Li, Lqa, Lr, dtype=torch.float32, device=qa.device)
        Sn = torch.empty_like(S)
        flags = _flags()
        _rc(lib.stage_grp_qa_ctx_fwd(qa.data_ptr(), cx.data_ptr(), qa_mask.data_ptr(), cx_mask.data_ptr(), _ptrs(params), mixed.data_ptr(),
                                     S.data_ptr(), Sn.data_ptr(), arena.data_ptr(), ab, flags, N, NA, Li, Lqa, Lr, D, float(scale), float(p),
                                     _u64(seeds), _stream()), "stage_grp_qa_ctx_fwd")
        ctx.save_for_backward(qa, cx, cx_mask, mixed, Sn, arena, *params)
        ctx.cfg = (N, NA, Li, Lqa, Lr, D, float(scale), float(p), tuple(seeds), flags, ab)
        ctx.set_materialize_grads(False)      # raw S only receives a gradient with the supervised attention loss
        return mixed, S, Sn

    @_on_device
    def backward(ctx, d_mixed, dS, dSn):
        from .ops import _fold_dsn
        qa, cx, cx_mask, mixed, Sn, arena, *params = ctx.saved_tensors
        N, NA, Li, Lqa, Lr, D, scale, p, seeds, flags, ab = ctx.cfg
        dS = _fold_dsn(dS, dSn, Sn, scale)
        d_mixed = _chk(d_mixed, "d_mixed") if d_mixed is not None else torch.zeros_like(mixed)
        dS = _chk(dS, "dS") if dS is not None else None
        lib = _lib.load()
        grads = _grad_views(params)
        d_qa, d_cx = torch.empty_like(qa), torch.empty_like(cx)
        tb = _size("stage_grp_qa_ctx_bwd_tmp_bytes", N, NA, Li, Lqa, Lr, D)
        tmp = _buf(tb, qa.device)
        _rc(lib.stage_grp_qa_ctx_bwd(d_mixed.data_ptr(), None if dS is None else dS.data_ptr(), qa.data_ptr(), cx.data_ptr(),
                                     cx_mask.data_ptr(), mixed.data_ptr(), Sn.data_ptr(), _ptrs(params), _ptrs(grads), d_qa.data_ptr(),
                                     d_cx.data_ptr(), arena.data_ptr(), ab, flags, tmp.data_ptr(), tb, N, NA, Li, Lqa, Lr, D, scale, p,
                                     _u64(seeds), _stream()), "stage_grp_qa_ctx_bwd")
        return (d_qa, d_cx, None, None, None, None, None) + _deliver(ctx.sinks, grads)


def qa_ctx(qa, cx, qa_mask, cx_mask, scale: float, p: float, seeds, params):
    """-> mixed (N,NA,Li,Lqa,D), raw scores, normalised scores (N,NA,Li,Lqa,Lr).  params: ln.w ln.b fc.w fc.b."""
    return _QaCtx.apply(qa, cx, qa_mask, cx_mask, scale, p, tuple(seeds), *params)


# ---------------------------------------------------------------------------------------------------------------
# G3r / G2r: the same groups on RAGGED TOKEN ROWS (tvqaplus_amd/ragged.py, csrc/groups.hip "RAGGED TOKEN ROWS")
# ---------------------------------------------------------------------------------------------------------------
def qa_ctx_rag_supported(N, NA, Li, Lqa, Lr, D, lay) -> bool:
    return bool(_lib.load().stage_grp_qa_ctx_rag_supported(N, NA, Li, Lqa, Lr, D, lay.U, lay.Fc))


class _QaCtxRag(torch.autograd.Function):
    @_on_device
    def forward(ctx, qa, cx, qa_mask, cx_mask, lay, clay, scale: float, p: float, seeds, *params):
        qa, cx = _chk(qa, "qa"), _chk(cx, "ctx")
        qa_mask, cx_mask = _chk(qa_mask, "qa_mask"), _chk(cx_mask, "ctx_mask")
        ctx.sinks = _sinks(params)
        params = _params(params)
        N, NA, Lqa, D = qa.shape
        _, Li, Lr = cx_mask.shape
        Uc = cx.numel() // D                       # rows of the context stream: N * Li * Lr, or its compact rows (clay)
        assert Uc == (clay.U if clay is not None else N * Li * Lr)
        T5 = lay.attention_tables(clay)
        lib = _lib.load()
        ab = _size("stage_grp_qa_ctx_rag_arena_bytes", N, NA, Lqa, D, lay.Ucap, lay.Fc)
        arena = _buf(ab, qa.device)
        mixed = torch.empty(lay.Ucap, D, dtype=torch.float32, device=qa.device)[:lay.U]
        S = torch.empty(N, NA, Li, Lqa, Lr, dtype=torch.float32, device=qa.device)
        Sn = torch.empty_like(S)
        flags = _flags()
        _rc(lib.stage_grp_qa_ctx_rag_fwd(qa.data_ptr(), cx.data_ptr(), qa_mask.data_ptr(), cx_mask.data_ptr(), _ptrs(params),
                                         mixed.data_ptr(), S.data_ptr(), Sn.data_ptr(), T5, arena.data_ptr(), ab, flags, N, NA, Li, Lqa, Lr,
                                         D, lay.U, lay.Ucap, lay.Fc, Uc, float(scale), float(p), _u64(seeds), _stream()),
            "stage_grp_qa_ctx_rag_fwd")
        ctx.save_for_backward(qa, cx, cx_mask, mixed, Sn, arena, *params)
        ctx.cfg = (N, NA, Li, Lqa, Lr, D, float(scale), float(p), tuple(seeds), flags, ab, Uc)
        ctx.lay, ctx.clay = lay, clay
        ctx.spent = False
        ctx.set_materialize_grads(False)
        return mixed, S, Sn

    @_on_device
    def backward(ctx, d_mixed, dS, dSn):
        from .ops import _fold_dsn
        if ctx.spent:      # the backward overwrites the saved attention output with its gradient (csrc/groups.hip G3r)
            raise RuntimeError("tvqaplus_amd: the ragged attention group can be differentiated once per forward "
                               "(retain_graph + a second backward: set STAGE_NO_RAGGED=1)")
        ctx.spent = True
        qa, cx, cx_mask, mixed, Sn, arena, *params = ctx.saved_tensors
        N, NA, Li, Lqa, Lr, D, scale, p, seeds, flags, ab, Uc = ctx.cfg
        lay = ctx.lay
        dS = _fold_dsn(dS, dSn, Sn, scale)
        d_mixed = _chk(d_mixed, "d_mixed") if d_mixed is not None else torch.zeros_like(mixed)
        dS = _chk(dS, "dS") if dS is not None else None
        lib = _lib.load()
        grads = _grad_views(params)
        d_qa, d_cx = torch.empty_like(qa), torch.empty_like(cx)
        cap = ctx.clay.Ucap if ctx.clay is not None else Uc
        tb = _size("stage_grp_qa_ctx_rag_bwd_tmp_bytes", N, NA, Li, Lqa, Lr, D, lay.Ucap, cap)
        tmp = _buf(tb, qa.device)
        _rc(lib.stage_grp_qa_ctx_rag_bwd(d_mixed.data_ptr(), None if dS is None else dS.data_ptr(), qa.data_ptr(), cx.data_ptr(),
                                         cx_mask.data_ptr(), mixed.data_ptr(), Sn.data_ptr(), _ptrs(params), _ptrs(grads), d_qa.data_ptr(),
                                         d_cx.data_ptr(), lay.attention_tables(ctx.clay), arena.data_ptr(), ab, flags, tmp.data_ptr(), tb, N, NA, Li,
                                         Lqa, Lr, D, lay.U, lay.Ucap, lay.Fc, Uc, scale, p, _u64(seeds), _stream()),
            "stage_grp_qa_ctx_rag_bwd")
        return (d_qa, d_cx, None, None, None, None, None, None, None) + _deliver(ctx.sinks, grads)


def qa_ctx_rag(qa, cx, qa_mask, cx_mask, lay, clay, scale: float, p: float, seeds, params):
    """As ``qa_ctx``; ``mixed`` comes back as the (U, D) compact rows of ``lay`` (a ``ragged.RaggedLayout``), the score maps dense.
    ``clay`` (a ``ragged.CtxLayout`` or None): ``cx`` holds the compact rows of the context stream instead of (N, Li, Lr, D)."""
    return _QaCtxRag.apply(qa, cx, qa_mask, cx_mask, lay, clay, scale, p, tuple(seeds), *params)


class _EncoderRag(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, pe, qa_mask, lay, k: int, p: float, seeds, *params):
        x = _chk(x, "x")                       # (U, D) compact
        pe = _chk(pe, "pe")
        qa_mask = _chk(qa_mask, "qa_mask") if qa_mask is not None else None
        ctx.sinks = _sinks(params)
        params = _params(params)
        U, D = x.shape
        assert U == lay.U
        n_conv = (len(params) - 2) // 6
        Rd = lay.out_rows if qa_mask is not None else 0
        lib = _lib.load()
        ab = _size("stage_grp_encoder_rag_arena_bytes", lay.Ucap, Rd, D, n_conv)
        arena = _buf(ab, x.device)
        out = torch.empty(Rd, D, dtype=torch.float32, device=x.device) if qa_mask is not None else _rows(U, D, x.device)
        flags = _flags()
        _rc(lib.stage_grp_encoder_rag_fwd(x.data_ptr(), pe.data_ptr(), None if qa_mask is None else qa_mask.data_ptr(), _ptrs(params),
                                          out.data_ptr(), lay.T,
                                          arena.data_ptr(), ab, flags, lay.U, lay.Ucap, lay.S, Rd, lay.Lqa, D, n_conv, int(k), float(p),
                                          _u64(seeds), _stream()), "stage_grp_encoder_rag_fwd")
        ctx.save_for_backward(qa_mask, arena, *params)
        ctx.cfg = (U, D, Rd, n_conv, int(k), float(p), tuple(seeds), flags, ab)
        ctx.lay = lay
        return out

    @_on_device
    def backward(ctx, dout):
        qa_mask, arena, *params = ctx.saved_tensors
        U, D, Rd, n_conv, k, p, seeds, flags, ab = ctx.cfg
        lay = ctx.lay
        dout = _chk(dout, "dout")
        lib = _lib.load()
        grads = _grad_views(params)
        dx = torch.empty(lay.Ucap, D, dtype=torch.float32, device=dout.device)[:U] if ctx.needs_input_grad[0] else None
        tb = _size("stage_grp_encoder_rag_bwd_tmp_bytes", lay.Ucap, D, k)
        tmp = _buf(tb, dout.device)
        _rc(lib.stage_grp_encoder_rag_bwd(dout.data_ptr(), None if qa_mask is None else qa_mask.data_ptr(), _ptrs(params), _ptrs(grads),
                                          None if dx is None else dx.data_ptr(), lay.T, arena.data_ptr(), ab, flags, tmp.data_ptr(), tb,
                                          lay.U, lay.Ucap, lay.S, Rd, lay.Lqa, D, n_conv, k, p, _u64(seeds), _stream()),
            "stage_grp_encoder_rag_bwd")
        return (dx, None, None, None, None, None, None) + _deliver(ctx.sinks, grads)


def encoder_block_rag(x, pe, qa_mask, lay, k: int, p: float, seeds, params):
    """x (U, D) compact rows of ``lay`` -> (N*NA*Li, D): the classifier encoder block + the masked max over the words of every
    (example, candidate, frame) (model/stage.py:502-503); qa_mask (N*NA, Lqa).  qa_mask None: no pooling, (U, D) comes back -- an
    encoder block over the ragged sequences of a context stream (``lay`` a ``ragged.CtxLayout``)."""
    return _EncoderRag.apply(x, pe, qa_mask, lay, k, p, tuple(seeds), *params)


class _InputMLPRag(torch.autograd.Function):
    @_on_device
    def forward(ctx, x, clay, l2: int, p: float, seeds, *params):
        x = _chk(x, "x")                       # the padded feature tensor, (..., K0)
        ctx.sinks = _sinks(params)
        params = _params(params)
        K0, H, D = x.shape[-1], params[2].shape[0], params[6].shape[0]
        M = clay.U
        lib = _lib.load()
        ab = _size("stage_grp_input_mlp_arena_bytes", clay.Ucap, K0, H, D, int(l2))
        arena = _buf(ab, x.device)
        out = _rows(M, D, x.device)
        flags = _flags()
        _rc(lib.stage_grp_input_mlp_rag_fwd(x.data_ptr(), clay.src_rows.data_ptr(), _ptrs(params), out.data_ptr(), arena.data_ptr(), ab, flags,
                                            M, K0, H, D, int(l2), float(p), _u64(seeds), _stream()), "stage_grp_input_mlp_rag_fwd")
        ctx.save_for_backward(x, arena, *params)
        ctx.cfg = (M, K0, H, D, int(l2), float(p), tuple(seeds), flags, ab)
        ctx.clay = clay
        return out

    @_on_device
    def backward(ctx, dout):
        x, arena, *params = ctx.saved_tensors
        M, K0, H, D, l2, p, seeds, flags, ab = ctx.cfg
        dout = _chk(dout, "dout")
        lib = _lib.load()
        grads = _grad_views(params)
        tb = _size("stage_grp_input_mlp_bwd_tmp_bytes", ctx.clay.Ucap, K0, H, D)
        tmp = _buf(tb, x.device)
        _rc(lib.stage_grp_input_mlp_rag_bwd(dout.data_ptr(), x.data_ptr(), ctx.clay.src_rows.data_ptr(), _ptrs(params), _ptrs(grads),
                                            arena.data_ptr(), ab, flags, tmp.data_ptr(), tb, M, K0, H, D, l2, p, _u64(seeds), _stream()),
            "stage_grp_input_mlp_rag_bwd")
        return (None, None, None, None, None) + _deliver(ctx.sinks, grads)


def input_mlp_rag(x, clay, l2: bool, p: float, seeds, params):
    """As ``input_mlp`` on the compact rows of a ragged context stream: x the padded (..., K0) features, ``clay.src_rows`` the rows of it
    that exist; -> (clay.U, D)."""
    return _InputMLPRag.apply(x, clay, int(bool(l2)), p, tuple(seeds), *params)


# ---------------------------------------------------------------------------------------------------------------
# G4 two-stream fusion
# ---------------------------------------------------------------------------------------------------------------
class _ConcatFc(torch.autograd.Function):
    @_on_device
    def forward(ctx, s, v, p: float, seeds, *params):
        s, v = _chk(s, "s"), _chk(v, "v")
        params = _params(params)
        D = s.shape[-1]
        U = s.numel() // D
        lib = _lib.load()
        ab = _size("stage_grp_concat_fc_arena_bytes", U, D)
        arena = _buf(ab, s.device)
        out = _rows(U, D, s.device)
        flags = _flags()
        _rc(lib.stage_grp_concat_fc_fwd(s.data_ptr(), v.data_ptr(), _ptrs(params), out.data_ptr(), arena.data_ptr(), ab, flags, U, D,
                                        float(p), _u64(seeds), _stream()), "stage_grp_concat_fc_fwd")
        ctx.save_for_backward(s, v, arena, *params)
        ctx.cfg = (U, D, float(p), tuple(seeds), flags, ab)
        return out

    @_on_device
    def backward(ctx, dout):
        s, v, arena, *params = ctx.saved_tensors
        U, D, p, seeds, flags, ab = ctx.cfg
        dout = _chk(dout, "dout")
        lib = _lib.load()
        grads = _grad_views(params)
        ds, dv = _rows(U, D, s.device), _rows(U, D, s.device)
        tb = _size("stage_grp_concat_fc_bwd_tmp_bytes", U, D)
        tmp = _buf(tb, s.device)
        _rc(lib.stage_grp_concat_fc_bwd(dout.data_ptr(), s.data_ptr(), v.data_ptr(), _ptrs(params), _ptrs(grads), ds.data_ptr(),
                                        dv.data_ptr(), arena.data_ptr(), ab, flags, tmp.data_ptr(), tb, U, D, p, _u64(seeds), _stream()),
            "stage_grp_concat_fc_bwd")
        return (ds, dv, None, None) + tuple(grads)


def concat_fc(s, v, p: float, seeds, params):
    """params: ln3.w ln3.b fc.w fc.b ln.w ln.b."""
    return _ConcatFc.apply(s, v, p, tuple(seeds), *params)


# ---------------------------------------------------------------------------------------------------------------
# G5 temporal head (layer 0)
# ---------------------------------------------------------------------------------------------------------------
class _TemporalHead(torch.autograd.Function):
    @_on_device
    def forward(ctx, enc, p: float, seeds, *params):
        enc = _chk(enc, "enc")                        # (R, D)
        params = _params(params)
        R, D = enc.shape
        lib = _lib.load()
        ab = _size("stage_grp_temporal_head_arena_bytes", R, D)
        arena = _buf(ab, enc.device)
        first = torch.empty_like(enc)
        t_st = torch.empty(R, 1, dtype=torch.float32, device=enc.device)
        t_ed = torch.empty_like(t_st)
        flags = _flags()
        _rc(lib.stage_grp_temporal_head_fwd(enc.data_ptr(), _ptrs(params), first.data_ptr(), t_st.data_ptr(), t_ed.data_ptr(),
                                            arena.data_ptr(), ab, flags, R, D, float(p), _u64(seeds), _stream()),
            "stage_grp_temporal_head_fwd")
        ctx.save_for_backward(enc, first, arena, *params)
        ctx.cfg = (R, D, float(p), tuple(seeds), flags, ab)
        ctx.set_materialize_grads(False)
        return first, t_st, t_ed

    @_on_device
    def backward(ctx, d_first, d_st, d_ed):
        enc, first, arena, *params = ctx.saved_tensors
        R, D, p, seeds, flags, ab = ctx.cfg
        d_st = _chk(d_st, "d_st") if d_st is not None else torch.zeros(R, 1, dtype=torch.float32, device=enc.device)
        d_ed = _chk(d_ed, "d_ed") if d_ed is not None else torch.zeros(R, 1, dtype=torch.float32, device=enc.device)
        d_first = _chk(d_first, "d_first") if d_first is not None else None
        lib = _lib.load()
        grads = _grad_views(params)
        d_enc = torch.empty_like(enc)
        tb = _size("stage_grp_temporal_head_bwd_tmp_bytes", R, D)
        tmp = _buf(tb, enc.device)
        _rc(lib.stage_grp_temporal_head_bwd(None if d_first is None else d_first.data_ptr(), d_st.data_ptr(), d_ed.data_ptr(), enc.data_ptr(),
                                            first.data_ptr(), _ptrs(params), _ptrs(grads), d_enc.data_ptr(), arena.data_ptr(), ab, flags,
                                            tmp.data_ptr(), tb, R, D, p, _u64(seeds), _stream()), "stage_grp_temporal_head_bwd")
        return (d_enc, None, None) + tuple(grads)


def temporal_head(enc, p: float, seeds, params):
    """enc (R, D) -> first = enc + h (R, D), start / end scores (R, 1) each.
    params: proj(ln.w ln.b fc.w fc.b) st(ln.w ln.b fc.w fc.b) ed(ln.w ln.b fc.w fc.b)."""
    return _TemporalHead.apply(enc, p, tuple(seeds), *params)


# ---------------------------------------------------------------------------------------------------------------
# Head glue: temporal scores, span proposal, pooling + classifier, auxiliary losses (csrc/groups.hip, "head glue")
# ---------------------------------------------------------------------------------------------------------------
class _TScores(torch.autograd.Function):
    @_on_device
    def forward(ctx, t_st, t_ed, tm, N: int, NA: int, Li: int):
        t_st, t_ed, tm = _chk(t_st, "t_st"), _chk(t_ed, "t_ed"), _chk(tm, "frame mask")
        out = torch.empty(N, NA, Li, 2, dtype=torch.float32, device=t_st.device)
        _rc(_lib.load().stage_tscores_fwd(t_st.data_ptr(), t_ed.data_ptr(), tm.data_ptr(), out.data_ptr(), N, NA, Li, _stream()),
            "stage_tscores_fwd")
        ctx.save_for_backward(tm)
        ctx.cfg = (N, NA, Li, t_st.shape, t_ed.shape)
        return out

    @_on_device
    def backward(ctx, dout):
        (tm,) = ctx.saved_tensors
        N, NA, Li, s1, s2 = ctx.cfg
        dout = _chk(dout, "dout")
        d_st = torch.empty(s1, dtype=torch.float32, device=dout.device)
        d_ed = torch.empty(s2, dtype=torch.float32, device=dout.device)
        _rc(_lib.load().stage_tscores_bwd(dout.data_ptr(), tm.data_ptr(), d_st.data_ptr(), d_ed.data_ptr(), N, NA, Li, _stream()),
            "stage_tscores_bwd")
        return d_st, d_ed, None, None, None, None


def tscores(t_st, t_ed, frame_mask, N: int, NA: int, Li: int):
    """mask_logits(cat(t_st, t_ed), frame mask) -> (N, NA, Li, 2)   (model/stage.py:515-521)."""
    return _TScores.apply(t_st, t_ed, frame_mask, N, NA, Li)


def gt_spans(t_scores, target, lab_st, lab_ed):
    """(6, N) device floats: predicted start, end, confidence of the ground-truth candidate's span, the label's start, end and the
    answer index (model/stage.py:408-418; no gradient)."""
    t = _chk(t_scores.detach(), "t_scores")
    N, NA, Li, _ = t.shape
    target, lab_st, lab_ed = (_chk(v, "labels", torch.int64) for v in (target, lab_st, lab_ed))
    spans = torch.empty(6, N, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _rc(_lib.load().stage_gt_spans(t.data_ptr(), target.data_ptr(), lab_st.data_ptr(), lab_ed.data_ptr(), spans.data_ptr(), N, NA, Li,
                                       _stream()), "stage_gt_spans")
    return spans


def masked_max_raw(x, mask):
    """(R, L, D), (R, L) -> max (R, D), argmax (R, D) int32; no autograd node (the pooling group owns this gradient)."""
    x, mask = _chk(x.detach(), "x"), _chk(mask, "mask")
    R, L, D = x.shape
    out = torch.empty(R, D, dtype=torch.float32, device=x.device)
    idx = torch.empty(R, D, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _rc(_lib.load().stage_masked_max_fwd(x.data_ptr(), mask.data_ptr(), None, out.data_ptr(), idx.data_ptr(), R, L, D, _stream()),
            "stage_masked_max_fwd")
    return out, idx


class _PoolCls(torch.autograd.Function):
    @_on_device
    def forward(ctx, first, mask, glob, idx_g, meta, dims, p: float, seeds, *params):
        first, mask, glob = _chk(first, "first"), _chk(mask, "mask"), _chk(glob, "glob")
        idx_g, meta = _chk(idx_g, "idx_g", torch.int32), _chk(meta, "meta", torch.int32)
        params = _params(params)
        N, NA, Li, D, P = dims
        lib = _lib.load()
        ab = _size("stage_grp_pool_cls_arena_bytes", P, NA, D)
        arena = _buf(ab, first.device)
        logits = torch.empty(P * NA, 1, dtype=torch.float32, device=first.device)
        _rc(lib.stage_grp_pool_cls_fwd(first.data_ptr(), mask.data_ptr(), glob.data_ptr(), meta.data_ptr(), _ptrs(params), logits.data_ptr(),
                                       arena.data_ptr(), ab, N, NA, Li, D, P, float(p), _u64(seeds), _stream()), "stage_grp_pool_cls_fwd")
        ctx.save_for_backward(first, mask, idx_g, meta, arena, *params)
        ctx.cfg = (dims, float(p), tuple(seeds), ab)
        return logits

    @_on_device
    def backward(ctx, d_logits):
        first, mask, idx_g, meta, arena, *params = ctx.saved_tensors
        (N, NA, Li, D, P), p, seeds, ab = ctx.cfg
        d_logits = _chk(d_logits, "d_logits")
        lib = _lib.load()
        grads = _grad_views(params)
        d_first = torch.empty_like(first)
        tb = _size("stage_grp_pool_cls_bwd_tmp_bytes", P, NA, D)
        tmp = _buf(tb, first.device)
        _rc(lib.stage_grp_pool_cls_bwd(d_logits.data_ptr(), mask.data_ptr(), idx_g.data_ptr(), meta.data_ptr(), _ptrs(params), _ptrs(grads),
                                       d_first.data_ptr(), arena.data_ptr(), ab, tmp.data_ptr(), tb, N, NA, Li, D, P, p, _u64(seeds),
                                       _stream()), "stage_grp_pool_cls_bwd")
        return (d_first, None, None, None, None, None, None, None) + tuple(grads)


def pool_classifier(first, mask, glob, idx_g, meta, dims, p: float, seeds, params):
    """first (N*NA, Li, D) -> logits (P*NA, 1): local window max + global max of every proposal, LayerNorm(2D) + dropout,
    Linear(2D -> 1).  meta (device int32): src[P] | win[2P] | inv[2N].  params: ln.w ln.b fc.w fc.b."""
    return _PoolCls.apply(first, mask, glob, idx_g, meta, tuple(dims), p, tuple(seeds), *params)


class _TsLoss(torch.autograd.Function):
    @_on_device
    def forward(ctx, t_scores, target, lab_st, lab_ed, cand_offset: int, na_total: int = 0):
        t = _chk(t_scores, "t_scores")
        N, NA, Li, _ = t.shape
        target, lab_st, lab_ed = (_chk(v, "labels", torch.int64) for v in (target, lab_st, lab_ed))
        loss = torch.empty((), dtype=torch.float32, device=t.device)
        grad = torch.empty_like(t)
        scratch = torch.empty(N, dtype=torch.float32, device=t.device)
        _rc(_lib.load().stage_ts_loss(t.data_ptr(), target.data_ptr(), lab_st.data_ptr(), lab_ed.data_ptr(), loss.data_ptr(), grad.data_ptr(),
                                      scratch.data_ptr(), N, NA, Li, int(cand_offset), int(na_total), _stream()), "stage_ts_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


def ts_loss(t_scores, target, lab_st, lab_ed, cand_offset: int = 0, na_total: int = 0):
    """0.5 * (CE_sum(start scores of the ground-truth candidate, st) + CE_sum(end scores, ed))   (model/stage.py:539-555)."""
    return _TsLoss.apply(t_scores, target, lab_st, lab_ed, cand_offset, na_total)


class _TrainLoss(torch.autograd.Function):
    @_on_device
    def forward(ctx, logits, targets, att_loss, t_loss, scale, att_w: float, ts_w: float):
        x = _chk(logits, "logits")
        P, C = x.shape
        targets = _chk(targets, "targets", torch.int64)
        dev = x.device
        att = _chk(att_loss.reshape(1), "att_loss") if torch.is_tensor(att_loss) else None
        ts = _chk(t_loss.reshape(1), "t_loss") if torch.is_tensor(t_loss) else None
        sdev = _chk(scale.reshape(1), "scale") if torch.is_tensor(scale) else None
        loss = torch.empty((), dtype=torch.float32, device=dev)
        grad = torch.empty_like(x)
        _rc(_lib.load().stage_train_loss(x.data_ptr(), targets.data_ptr(), att.data_ptr() if att is not None else None,
                                         ts.data_ptr() if ts is not None else None, sdev.data_ptr() if sdev is not None else None,
                                         0.0 if sdev is not None else float(scale), float(att_w), float(ts_w), loss.data_ptr(),
                                         grad.data_ptr(), P, C, _stream()), "stage_train_loss")
        ctx.save_for_backward(grad)
        ctx.w = (float(att_w) if att is not None else None, float(ts_w) if ts is not None else None)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        aw, tw = ctx.w
        return grad * g, None, (g * aw if aw is not None else None), (g * tw if tw is not None else None), None, None, None


def train_loss(logits, targets, att_loss, t_loss, scale, att_w: float, ts_w: float):
    """CE_sum(logits, targets) * scale + att_w * att_loss + ts_w * t_loss (main.py:55-60) and the cross entropy's gradient in ONE
    launch.  att_loss / t_loss: 0-d tensors, or anything else (a Python 0) = absent; scale: float or a 0-d device tensor."""
    return _TrainLoss.apply(logits, targets, att_loss, t_loss, scale, att_w, ts_w)


class _AttLoss(torch.autograd.Function):
    @_on_device
    def forward(ctx, scores, flat, M: int, hinge: bool, alpha: float, margin: float):
        scores = _chk(scores, "scores")
        flat = _chk(flat, "pair indices", torch.int64)
        coef = torch.empty(M, dtype=torch.float32, device=scores.device)
        loss = torch.empty((), dtype=torch.float32, device=scores.device)
        _rc(_lib.load().stage_att_loss_fwd(scores.data_ptr(), flat.data_ptr(), M, int(bool(hinge)), float(alpha), float(margin),
                                           coef.data_ptr(), loss.data_ptr(), _stream()), "stage_att_loss_fwd")
        ctx.save_for_backward(flat, coef)
        ctx.cfg = (M, scores.shape)
        return loss

    @_on_device
    def backward(ctx, g):
        flat, coef = ctx.saved_tensors
        M, shape = ctx.cfg
        g = _chk(g.reshape(1), "g")
        dS = torch.empty(shape, dtype=torch.float32, device=coef.device)
        _rc(_lib.load().stage_att_loss_bwd(flat.data_ptr(), coef.data_ptr(), g.data_ptr(), M, dS.data_ptr(), dS.numel(), _stream()),
            "stage_att_loss_bwd")
        return dS, None, None, None, None, None


def att_loss(scores, flat, M: int, loss_type: str, alpha: float, margin: float):
    """Supervised attention loss over M (positive, negative) pairs of a contiguous score tensor (model/stage.py:738-745)."""
    if loss_type not in ("lse", "hinge"):
        raise NotImplementedError("Only support hinge and lse")
    return _AttLoss.apply(scores, flat, M, loss_type == "hinge", alpha, margin)
