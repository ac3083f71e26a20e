"""MI355X-native STAGE: drop-in for ``model.stage.STAGE`` of jayleicn/TVQAplus (model/stage.py:55-348).

Same constructor (``STAGE(opt)``), same parameter / buffer names and shapes (``load_state_dict(strict=True)`` of a
reference checkpoint works, inference.py:87-89), same ``forward(batch)`` return conventions (model/stage.py:192-197,
297-312, 345-348).  The tensor path runs entirely on the HIP kernels of ``libstage_hip.so`` via ``tvqaplus_amd.ops``;
the torch ``nn`` modules below are parameter *containers* only (they are never called), which is what gives the
reference's state_dict keys and default initialisers for free.

Scope: the model forward/backward.  The host-side supervised-attention loss and box prediction
(model/stage.py:557-806) live in ``tvqaplus_amd.att_host`` and are pure host logic on top of ``vid_raw_s``.
"""
from __future__ import annotations

import copy

import math
import contextlib
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import groups, ops, ragged

NEG = -1e10


def _opt(batch, key, default=None):
    """Optional batch field (candidate-sharded batches of tvqaplus_amd.parallel.CandidateLayout carry extra ones)."""
    if isinstance(batch, dict):
        return batch.get(key, default)
    return getattr(batch, key, default)


# ---------------------------------------------------------------------------------------------------------------
# parameter containers (names mirror the reference's attribute names -> identical state_dict keys)
# ---------------------------------------------------------------------------------------------------------------
_SIDE_STREAMS: Dict[torch.device, Tuple[torch.cuda.Stream, torch.cuda.Stream]] = {}    # branch streams (STAGE.use_streams), per device


class _ScatterBuckets(torch.autograd.Function):
    """(frames_b, Lb, D) bucket outputs -> one zero-filled (M, L, D) tensor; the backward hands every bucket its slice of the
    gradient.  (Plain in-place index_copy_ on views would make autograd clone the whole (M, L, D) gradient once per bucket.)"""

    @staticmethod
    def forward(ctx, M, L, idxs, *ys):
        out = torch.zeros(M, L, ys[0].shape[-1], device=ys[0].device, dtype=ys[0].dtype)
        for ix, y in zip(idxs, ys):
            out[:, :y.shape[1]].index_copy_(0, ix, y)
        ctx.idxs, ctx.lens = idxs, [y.shape[1] for y in ys]
        return out

    @staticmethod
    def backward(ctx, g):
        return (None, None, None) + tuple(g[:, :lb].index_select(0, ix) for ix, lb in zip(ctx.idxs, ctx.lens))


class _PositionTable(nn.Module):
    """model/position_encoding.py:19-31: fixed sinusoid table registered as buffer ``pe`` (max_len, D)."""

    def __init__(self, n_filters: int, max_len: int = 500):
        super().__init__()
        self.register_buffer("pe", self.table(max_len, n_filters))
        self._ext: Optional[torch.Tensor] = None

    @staticmethod
    def table(L: int, D: int) -> torch.Tensor:
        pos = torch.arange(0, L).float().unsqueeze(1)
        div = torch.exp(torch.arange(0, D, 2).float() * -(math.log(10000.0) / D))
        pe = torch.zeros(L, D)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        return pe

    def rows(self, L: int) -> torch.Tensor:
        """First L rows; beyond the 500 registered positions (where the reference crashes, SURVEY.md note 3) the
        same closed form is continued and cached."""
        if L <= self.pe.shape[0]:
            return self.pe
        if self._ext is None or self._ext.shape[0] < L or self._ext.device != self.pe.device:
            self._ext = self.table(L, self.pe.shape[1]).to(self.pe.device)
        return self._ext


class _DWSepConv(nn.Module):
    """model/cnn.py:23-28 parameters: depthwise Conv1d(C,C,k,groups=C,pad=k//2) + pointwise Conv1d(C,C_out,1)."""

    def __init__(self, in_ch: int, out_ch: int, k: int):
        super().__init__()
        self.depthwise_conv = nn.Conv1d(in_ch, in_ch, kernel_size=k, groups=in_ch, padding=k // 2)
        self.pointwise_conv = nn.Conv1d(in_ch, out_ch, kernel_size=1, padding=0)


class _MHAParams(nn.Module):
    """model/self_attention.py:19-33: four Linear(D, D); attention-probability dropout fixed at 0.1."""

    def __init__(self, nh: int, d_model: int):
        super().__init__()
        assert d_model % nh == 0
        self.nh = nh
        # model/self_attention.py:26 `clones(nn.Linear(d_model, d_model), 4)` deep-copies ONE initialised layer: the four
        # projections start identical and consume a single draw of the generator
        first = nn.Linear(d_model, d_model)
        self.linears = nn.ModuleList([first] + [copy.deepcopy(first) for _ in range(3)])
        self.p_attn_drop = 0.1


class _EncoderBlockParams(nn.Module):
    """model/encoder.py:11-27."""

    def __init__(self, n_conv: int, kernel_size: int, n_filters: int, num_heads: int):
        super().__init__()
        self.n_conv, self.num_heads = n_conv, num_heads
        self.position_encoding = _PositionTable(n_filters)
        self.layer_norm = nn.ModuleList([nn.LayerNorm(n_filters) for _ in range(n_conv)])
        self.final_layer_norm = nn.LayerNorm(n_filters)
        self.conv = nn.ModuleList([_DWSepConv(n_filters, n_filters, kernel_size) for _ in range(n_conv)])
        if num_heads != 0:
            self.multi_head_attn = _MHAParams(num_heads, n_filters)
            self.attn_layer_norm = nn.LayerNorm(n_filters)


class _StackedEncoderParams(nn.Module):
    """model/encoder.py:55-64."""

    def __init__(self, n_blocks, n_conv, kernel_size, hidden_size, num_heads):
        super().__init__()
        self.stacked_encoderBlocks = nn.ModuleList(
            [_EncoderBlockParams(n_conv, kernel_size, hidden_size, num_heads) for _ in range(n_blocks)])


class _LinearWrapperParams(nn.Module):
    """model/stage.py:15-25: ``conv`` = Sequential(LayerNorm, Dropout, Linear)."""

    def __init__(self, in_hsz, out_hsz, dropout, relu):
        super().__init__()
        self.relu = relu
        self.conv = nn.Sequential(nn.LayerNorm(in_hsz), nn.Dropout(dropout), nn.Linear(in_hsz, out_hsz))


class _ConvLinearParams(nn.Module):
    """model/stage.py:35-48: ``conv`` = Sequential(LayerNorm, Dropout, DepthwiseSeparableConv(k=3))."""

    def __init__(self, in_hsz, out_hsz, dropout):
        super().__init__()
        self.conv = nn.Sequential(nn.LayerNorm(in_hsz), nn.Dropout(dropout), _DWSepConv(in_hsz, out_hsz, 3))


# ---------------------------------------------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------------------------------------------
def reference_loss(outputs, targets, att_loss, temporal_loss, n_examples: int, att_weight: float = 0.1, ts_weight: float = 0.5, scale=None):
    """The training loss of the reference's driver (main.py:55-60) for the outputs of ``STAGE.forward`` in training mode:

        criterion(outputs, targets) * (len(qids) / len(targets)) + att_weight * att_loss + ts_weight * temporal_loss

    (criterion = CrossEntropyLoss(reduction="sum"), main.py:188) -- in ONE kernel launch forward and one small multiply backward
    (csrc/groups.hip: train_loss_kernel) instead of the ~20 launches of the eager expression and its autograd nodes.  It matters because
    of WHERE they sit: right behind the proposal read-back of ``get_proposals``, the one point of the step where the device has nothing
    queued and waits for the host call by call (profiles/r06_step_idle_gaps.txt).  A drop-in for those four lines of a training loop,
    optional: the eager lines give the same value (tests/test_hip_groups.py::test_reference_loss_matches_the_eager_lines).
    ``scale``: overrides len(qids) / len(targets) (multi-GPU: ``parallel.global_loss_scale(..., as_tensor=True)``)."""
    if scale is None:
        scale = float(n_examples) / len(targets)
    return groups.train_loss(outputs, targets, att_loss, temporal_loss, scale, att_weight, ts_weight)


def _block_params(blk):
    """[ln.w, ln.b, dw.w, dw.b, pw.w, pw.b] per conv layer + the final LayerNorm's pair of an encoder block, in the order the encoder
    K-groups take them.  Cached on the block (plain attribute): the Parameter OBJECTS never change (``.to()`` / the optimizer update them
    in place), and walking ``nn.Module.__getattr__`` / ``ModuleList.__getitem__`` for 14 parameters per call is host time the small-batch
    step is bound by."""
    cached = blk.__dict__.get("_stage_plist")
    if cached is not None:
        ps, owners = cached
        # every entry is checked against the module that owns it (dict look-ups, no __getattr__ walk): a replaced Parameter object
        # (per-layer surgery, parametrize, a swapped nn.Parameter) must not leave the K-groups computing with the old tensor
        for p, (params, key) in zip(ps, owners):
            if params[key] is not p:
                cached = None
                break
    if cached is None:
        ps, owners = [], []
        for i in range(blk.n_conv):
            c = blk.conv[i]
            for m in (blk.layer_norm[i], c.depthwise_conv, c.pointwise_conv):
                for key in ("weight", "bias"):
                    ps.append(m._parameters[key])
                    owners.append((m._parameters, key))
        for key in ("weight", "bias"):
            ps.append(blk.final_layer_norm._parameters[key])
            owners.append((blk.final_layer_norm._parameters, key))
        blk.__dict__["_stage_plist"] = (ps, owners)
    return ps


class STAGE(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.inference_mode = False
        self._qkv_cache = {}
        self.sub_flag = opt.sub_flag
        self.vfeat_flag = opt.vfeat_flag
        self.vfeat_size = opt.vfeat_size
        self.t_iter = opt.t_iter
        self.extra_span_length = opt.extra_span_length
        self.add_local = opt.add_local
        self.use_sup_att = opt.use_sup_att
        self.num_negatives = opt.num_negatives
        self.negative_pool_size = opt.negative_pool_size
        self.num_hard = opt.num_hard
        self.drop_topk = opt.drop_topk
        self.margin = opt.margin
        self.att_loss_type = opt.att_loss_type
        self.scale = opt.scale
        self.alpha = opt.alpha
        self.dropout = opt.dropout
        self.hsz = opt.hsz
        self.bsz = None
        self.num_seg = None
        self.num_a = 5
        self.flag_cnt = self.sub_flag + self.vfeat_flag
        self.wd_size = opt.embedding_size
        self.bridge_hsz = 300

        def bridge(in_size):  # model/stage.py:85-91 / 98-104
            return nn.Sequential(nn.LayerNorm(in_size), nn.Dropout(self.dropout), nn.Linear(in_size, self.bridge_hsz),
                                 nn.ReLU(True), nn.LayerNorm(self.bridge_hsz))

        self.bert_word_encoding_fc = bridge(self.wd_size)
        if self.sub_flag:
            print("Activate sub branch")
        if self.vfeat_flag:
            print("Activate vid branch")
            self.vid_fc = bridge(self.vfeat_size)
        if self.flag_cnt == 2:  # model/stage.py:106-113
            self.concat_fc = nn.Sequential(nn.LayerNorm(3 * self.hsz), nn.Dropout(self.dropout),
                                           nn.Linear(3 * self.hsz, self.hsz), nn.ReLU(True), nn.LayerNorm(self.hsz))
        self.input_embedding = nn.Sequential(nn.Dropout(self.dropout), nn.Linear(self.bridge_hsz, self.hsz),
                                             nn.ReLU(True), nn.LayerNorm(self.hsz))
        self.input_encoder = _StackedEncoderParams(opt.input_encoder_n_blocks, opt.input_encoder_n_conv,
                                                   opt.input_encoder_kernel_size, self.hsz,
                                                   opt.input_encoder_n_heads)
        # str_attn has no parameters (model/stage.py:129-131)
        self.c2q_down_projection = nn.Sequential(nn.LayerNorm(3 * self.hsz), nn.Dropout(self.dropout),
                                                 nn.Linear(3 * self.hsz, self.hsz), nn.ReLU(True))
        self.cls_encoder = _StackedEncoderParams(opt.cls_encoder_n_blocks, opt.cls_encoder_n_conv,
                                                 opt.cls_encoder_kernel_size, self.hsz, opt.cls_encoder_n_heads)
        self.cls_projection_layers = nn.ModuleList(
            [_LinearWrapperParams(self.hsz, self.hsz, self.dropout, relu=True)]
            + [_ConvLinearParams(self.hsz, self.hsz, self.dropout) for _ in range(self.t_iter)])
        self.temporal_scoring_st_layers = nn.ModuleList(
            [_LinearWrapperParams(self.hsz, 1, self.dropout, relu=False) for _ in range(self.t_iter + 1)])
        self.temporal_scoring_ed_layers = nn.ModuleList(
            [_LinearWrapperParams(self.hsz, 1, self.dropout, relu=False) for _ in range(self.t_iter + 1)])
        self.temporal_criterion = nn.CrossEntropyLoss(reduction="sum")
        self.classifier = _LinearWrapperParams(self.hsz * 2 if self.add_local else self.hsz, 1, self.dropout,
                                               relu=False)
        # developer switch: False (or STAGE_NO_FUSE_LN_DWCONV=1) = separate LayerNorm and depthwise-conv kernels
        self.fuse_ln_dwconv = os.environ.get("STAGE_NO_FUSE_LN_DWCONV") is None
        self.fuse_ln_max = os.environ.get("STAGE_NO_FUSE_LN_MAX") is None
        # opt-in: correct, but 0.6 ms SLOWER per step as built (DESIGN.md finding 25) -- the register budget forces the variant
        # without the weight-chunk prefetch
        self.fuse_input_ln = os.environ.get("STAGE_FUSE_INPUT_LN") is not None
        # launch sequencing: True = one C call per fused-op GROUP (tvqaplus_amd/groups.py, csrc/groups.hip: ~25 host calls per
        # training step); False (or STAGE_NO_GROUPS=1) = one call per kernel (tvqaplus_amd/ops.py, ~360).  Same kernels, same order,
        # same dropout streams: tests/test_hip_groups.py holds the two equal.  Groups cover fp32 storage and encoder blocks
        # without self-attention; everything else falls back to the per-op path group by group.
        self.use_groups = os.environ.get("STAGE_NO_GROUPS") is None
        self.gate_shared = os.environ.get("STAGE_NO_PARAM_GATE") is None   # groups.gate for the modules applied to several streams
        self._gate_map = {}
        # ragged token rows (tvqaplus_amd/ragged.py): the (N, 5, Li, Lqa, .) kernels run on the rows that can reach an output or a
        # gradient -- live frames x (valid words + the classifier encoder's convolution halo).  False / STAGE_NO_RAGGED=1: every padded
        # row is computed, as the reference does.  Needs the K-group path at hsz = 128 (otherwise the dense path runs, silently: it is
        # the same function).  ``last_ragged``: the layout of the last forward (None = dense), for tests and the bench record.
        self.use_ragged = os.environ.get("STAGE_NO_RAGGED") is None
        # ... and the context streams in front of the attention on their valid words / regions + the input encoder's halo
        # (STAGE_NO_RAGGED_CTX=1: dense context streams, ragged statement rows)
        self.use_ragged_ctx = os.environ.get("STAGE_NO_RAGGED_CTX") is None
        # padded statement rows (N * 5 * Li * Lqa) below which the ragged layout is not worth its host work (_ragged_layout)
        self.ragged_min_rows = int(os.environ.get("STAGE_RAGGED_MIN_ROWS", "200000"))
        # context streams the ragged group path does not take (rows longer than 64, bf16 storage, hsz != 128): length BUCKETS -- the
        # frames are sorted into a few dense (frames, Lb, .) batches by valid length + halo and every bucket runs the ordinary path on
        # its Lb positions; dead frames run nowhere (ragged.bucket_plan).  ``last_buckets``: {stream: [(frames, Lb), ...]} of the
        # last forward.  STAGE_NO_CTX_BUCKETS=1 / use_ctx_buckets = False: one dense (frames, L, .) batch, as the reference.
        self.use_ctx_buckets = os.environ.get("STAGE_NO_CTX_BUCKETS") is None
        # Branch streams (STAGE_STREAMS=0..4 / use_streams; default 3 since round 5, 2 in round 4): the three branches of the forward -- statements, subtitles,
        # video -- are independent up to the attention (statements <-> context) and the fusion (subtitles <-> video).  Level 1 runs the
        # statement branch (input MLP + encoder over N*5 statements: ~90 latency-bound launches forward + backward, 0.66 ms of device
        # time that uses a few CUs) on a side stream next to the subtitle branch; level 2 also the video input MLP / encoder; level 3
        # the video attention as well, its forward fenced behind the subtitle attention (only the backward overlaps: -1..2 % on level 2 with the K1 forward kernels
        # still alone on their streams -- round 5, two A/B runs on one box: 13.65 / 13.61 ms at level 2, 13.50 / 13.35 ms at level 3);
        # level 4 without the fence (two saturating kernels side by side: fastest step, -2..3 % on level 2, but a kernel's duration is
        # then no statement about that kernel).  The backward of every op runs on its forward's stream (autograd), so the branches overlap there too.
        # Joins: events in front of the attention / the fusion; tensors that cross streams are registered with the allocator.
        self.use_streams = int(os.environ.get("STAGE_STREAMS", "3") or 0)
        self.last_buckets: Dict[str, list] = {}
        self._mask_info = None
        self.last_ragged: Optional[ragged.RaggedLayout] = None
        self.last_ragged_ctx: Dict[str, ragged.CtxLayout] = {}
        self._rag_stage = None
        self._ctx_stage: Dict[str, object] = {}
        # storage type of the activations between kernels: fp32 (the reference's), or bf16 with ``opt.storage_dtype = "bf16"``
        # (BASELINE.json configs[4]: bf16 weights / activations, fp32 softmax / statistics / accumulation; parameters stay
        # fp32 master copies, a weight is rounded to bf16 when a GEMM stages it; scores, losses and logits are fp32)
        sd = str(_opt(opt, "storage_dtype", "fp32")).lower()
        if sd not in ("fp32", "float32", "bf16", "bfloat16"):
            raise ValueError("opt.storage_dtype must be 'fp32' or 'bf16', got %r" % sd)
        self.storage = torch.bfloat16 if sd in ("bf16", "bfloat16") else torch.float32
        self._span_host = None      # pinned landing buffer of the per-step proposal spans (get_proposals)
        self._meta_stage = None     # pinned staging of the per-step proposal bookkeeping (_proposals_grouped)
        # counter-based dropout stream (csrc/common.h): seeded lazily from the seed of torch's default generator at the first
        # use (so torch.manual_seed() before training takes effect, as for the reference's nn.Dropout) and mixed with the
        # process rank (identical seeds on every data-parallel rank would drop the same units everywhere).  The stream
        # position is not part of state_dict(): like the reference, a resumed run does not replay the masks.
        self._seed_state: Optional[int] = None
        self._dropout_rank: Optional[int] = None
        self.mha_dropout_override: Optional[float] = None  # tests: the reference's fixed 0.1 can be zeroed

    # ---- dropout bookkeeping ---------------------------------------------------------------------------------
    def _p(self) -> float:
        return float(self.dropout) if self.training else 0.0

    def _seed(self) -> int:
        if self._seed_state is None:
            # the seed of torch's default generator as of NOW (the latest torch.manual_seed), read without consuming a draw:
            # the reference's negative sampling (get_att_loss) draws from that generator and must see the same sequence
            draw = int(torch.initial_seed())
            # ranks of one candidate group (parallel.CandidateLayout) replicate the context encoders of the SAME examples: they
            # share the stream of their example block (batch.dropout_rank), so the replicated work drops the same units
            rank = int(os.environ.get("RANK", "0")) if self._dropout_rank is None else int(self._dropout_rank)
            self._seed_state = (draw * 0x9E3779B97F4A7C15 + 0x1234567 + rank * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self._seed_state = (self._seed_state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._seed_state >> 1

    def _seeds(self, n: int) -> List[int]:
        return [self._seed() for _ in range(n)]

    def _grouped(self) -> bool:
        return self.use_groups and self.storage == torch.float32

    def _try_group(self, fn, n_seeds: int, *args):
        """Run one K-group; ``None`` when the C side declines the shape (nothing was launched; the dropout stream is rewound
        so that the per-op path draws the same seeds)."""
        state = self._seed_state
        seeds = self._seeds(n_seeds)
        try:
            return fn(seeds, *args)
        except groups.Unsupported:
            self._seed_state = state
            return None

    # ---- building blocks ---------------------------------------------------------------------------------------
    def _ln(self, x, ln: nn.LayerNorm, drop: bool = False, res=None, res_period: int = 0):
        return ops.layernorm(x, self._g(ln.weight), self._g(ln.bias), p=self._p() if drop else 0.0, seed=self._seed() if drop else 0,
                             res=res, res_period=res_period)

    def _encoder_block(self, x, mask, blk: _EncoderBlockParams, pool_mask=None):
        """model/encoder.py:29-52.  x (M, L, D).  Residual adds are deferred into the next LayerNorm's prologue.
        ``pool_mask`` (M, L): the caller only needs the masked max of the block's output over L (the classifier head,
        model/stage.py:503) -- the final LayerNorm and the max run as one pass and (M, D) is returned."""
        M, L, D = x.shape
        if (self._grouped() and blk.num_heads == 0 and blk.n_conv >= 1 and x.dtype == torch.float32 and self.fuse_ln_dwconv
                and self.fuse_ln_max):
            k = blk.conv[0].depthwise_conv.weight.shape[-1]
            params = [self._g(w) for w in _block_params(blk)]
            y = self._try_group(lambda seeds: groups.encoder_block(x, blk.position_encoding.rows(L), pool_mask, k, self._p(), seeds,
                                                                   params), (blk.n_conv + 1) // 2)
            if y is not None:
                return y
        pending, cur, period = x, blk.position_encoding.rows(L), L  # first LN sees x + pe[:L]
        for i in range(blk.n_conv):
            c = blk.conv[i]
            ln, drop = blk.layer_norm[i], (i % 2 == 0)
            if self.fuse_ln_dwconv and ops.ln_dwconv_supported(D, c.depthwise_conv.weight.shape[-1], pending.dtype):
                # LayerNorm output only feeds the depthwise conv: one fused pass, never materialised
                h, cur = ops.ln_dwconv(pending, self._g(ln.weight), self._g(ln.bias), self._g(c.depthwise_conv.weight),
                                       self._g(c.depthwise_conv.bias),
                                       p=self._p() if drop else 0.0, seed=self._seed() if drop else 0, res=cur,
                                       res_period=period)
            else:
                y, cur = self._ln(pending, ln, drop=drop, res=cur, res_period=period)
                h = ops.dwconv(y, c.depthwise_conv.weight, c.depthwise_conv.bias)
            period = 0
            pending = ops.linear(h, self._g(c.pointwise_conv.weight), self._g(c.pointwise_conv.bias), relu=True)
        if blk.num_heads != 0:
            y, cur = self._ln(pending, blk.attn_layer_norm, res=cur, res_period=period)
            period = 0
            mha = blk.multi_head_attn
            p_attn = mha.p_attn_drop if self.mha_dropout_override is None else self.mha_dropout_override
            p_attn = p_attn if self.training else 0.0
            if ops.mha_core_qkv_supported(L, D, mha.nh):
                # the three projections of model/self_attention.py:35-44 as ONE Linear(D -> 3D) on the stacked weight: one forward GEMM,
                # one dX GEMM (no sum of three input gradients), one weight-gradient GEMM; the attention core reads / writes the thirds
                # of the fused tensors in place (round 5; the stacked weight is a 192 KB torch.cat whose backward hands each
                # Linear its slice of the gradient)
                # (stacked ONCE per step and module -- _open_gates, on the main stream before the branches fork: the shared encoder
                # is applied to three streams and several length buckets)
                ck = self._qkv_cache.get(id(mha))
                if ck is None:
                    ck = (torch.cat([self._g(mha.linears[j].weight) for j in range(3)], dim=0),
                          torch.cat([self._g(mha.linears[j].bias) for j in range(3)], dim=0))
                    self._qkv_cache[id(mha)] = ck
                w_qkv, b_qkv = ck
                qkv = ops.linear(y, w_qkv, b_qkv)
                a = ops.mha_core_qkv(qkv, mask, mha.nh, p=p_attn, seed=self._seed() if p_attn > 0 else 0)
            else:
                q = ops.linear(y, self._g(mha.linears[0].weight), self._g(mha.linears[0].bias))
                k = ops.linear(y, self._g(mha.linears[1].weight), self._g(mha.linears[1].bias))
                v = ops.linear(y, self._g(mha.linears[2].weight), self._g(mha.linears[2].bias))
                a = ops.mha_core(q, k, v, mask, mha.nh, p=p_attn, seed=self._seed() if p_attn > 0 else 0)
            pending = ops.linear(a, self._g(mha.linears[3].weight), self._g(mha.linears[3].bias))
        if pool_mask is not None:
            if self.fuse_ln_max and period == 0 and ops.ln_masked_max_supported(pending, L, D):
                return ops.ln_masked_max(pending, cur, blk.final_layer_norm.weight, blk.final_layer_norm.bias, pool_mask)
            y, _ = self._ln(pending, blk.final_layer_norm, res=cur, res_period=period)
            return ops.masked_max(y, pool_mask)
        y, _ = self._ln(pending, blk.final_layer_norm, res=cur, res_period=period)
        return y

    def _stacked_encoder(self, x, mask, enc: _StackedEncoderParams, pool_mask=None):
        blocks = list(enc.stacked_encoderBlocks)
        if pool_mask is not None and not blocks:
            return ops.masked_max(x, pool_mask)
        for j, blk in enumerate(blocks):
            x = self._encoder_block(x, mask, blk, pool_mask=pool_mask if j == len(blocks) - 1 else None)
        return x

    def base_encoder(self, data, data_mask, init_encoder, downsize_encoder, input_encoder, l2_normalize=False, clay=None):
        """model/stage.py:350-363 (+ the F.normalize of :256 when l2_normalize)."""
        M, L, _ = data.shape
        if clay is not None:
            # ragged context rows (tvqaplus_amd/ragged.py: CtxLayout): the MLP reads the live rows of the padded features in place,
            # everything behind it -- and the attention -- runs on the compact rows
            params = [init_encoder[0].weight, init_encoder[0].bias, init_encoder[2].weight, init_encoder[2].bias,
                      init_encoder[4].weight, init_encoder[4].bias, downsize_encoder[1].weight, downsize_encoder[1].bias,
                      downsize_encoder[3].weight, downsize_encoder[3].bias]
            y = groups.input_mlp_rag(data, clay, l2_normalize, self._p(), self._seeds(2), [self._g(w) for w in params])
            if clay.tab.halo >= L:
                # whole frames (self-attention in the input encoder): S sequences of L positions, the ordinary encoder path
                mc = data_mask.index_select(0, clay.live_frames.long()).contiguous()
                return self._stacked_encoder(y.view(clay.S, L, -1), mc, input_encoder).reshape(clay.U, -1)
            for blk in input_encoder.stacked_encoderBlocks:
                bp = _block_params(blk)
                k = blk.conv[0].depthwise_conv.weight.shape[-1]
                y = groups.encoder_block_rag(y, blk.position_encoding.rows(L), None, clay, k, self._p(),
                                             self._seeds((blk.n_conv + 1) // 2), [self._g(w) for w in bp])
            return y
        if data.dtype != self.storage:
            data = data.to(self.storage)      # bf16 storage: features are rounded once on entry
        if self._grouped() and not self.fuse_input_ln and data.dtype == torch.float32:
            params = [init_encoder[0].weight, init_encoder[0].bias, init_encoder[2].weight, init_encoder[2].bias,
                      init_encoder[4].weight, init_encoder[4].bias, downsize_encoder[1].weight, downsize_encoder[1].bias,
                      downsize_encoder[3].weight, downsize_encoder[3].bias]
            params = [self._g(w) for w in params]
            y = self._try_group(lambda seeds: groups.input_mlp(data, l2_normalize, self._p(), seeds, params), 2)
            if y is not None:
                return self._stacked_encoder(y.view(M, L, -1), data_mask, input_encoder)
        if l2_normalize:
            data = ops.l2norm(data)
        if self.fuse_input_ln and ops.input_ln_linear_supported(data, init_encoder[2].weight):
            # features need no gradient: the first LayerNorm's gain / bias gradients come out of the dX GEMM's epilogue
            y = ops.input_ln_linear(data, init_encoder[0].weight, init_encoder[0].bias, init_encoder[2].weight, init_encoder[2].bias,
                                    p=self._p(), seed=self._seed())
        else:
            y, _ = self._ln(data, init_encoder[0], drop=True)
            y = ops.linear(y, self._g(init_encoder[2].weight), self._g(init_encoder[2].bias), relu=True)
        y, _ = self._ln(y, init_encoder[4], drop=True)            # LN(300) then input_embedding's Dropout
        y = ops.linear(y, self._g(downsize_encoder[1].weight), self._g(downsize_encoder[1].bias), relu=True)
        y, _ = self._ln(y, downsize_encoder[3])
        return self._stacked_encoder(y.view(M, L, -1), data_mask, input_encoder)

    def qa_ctx_attention(self, qa_embed, ctx_embed, qa_mask, ctx_mask, lay=None, clay=None):
        """model/stage.py:365-387.  qa_embed (N,5,Lqa,D), ctx_embed (N,Li,Lr,D), qa_mask (N,5,Lqa), ctx_mask (N,Li,Lr).
        ``lay`` (ragged.RaggedLayout): the mixed rows come back compact, (U, D)."""
        N, NA, Lqa, D = qa_embed.shape
        Li = ctx_mask.shape[1]
        p = self._p()
        # (s_mask.sum(-1) != 0) with s_mask = qa_mask (x) ctx_mask
        mixed_mask = ((qa_mask != 0).view(N, NA, 1, Lqa) & (ctx_mask.sum(-1) != 0).view(N, 1, Li, 1)).float()
        if lay is not None:
            proj = self.c2q_down_projection
            res = groups.qa_ctx_rag(qa_embed, ctx_embed, qa_mask, ctx_mask, lay, clay, self.scale, p, self._seeds(3),
                                    [self._g(w) for w in (proj[0].weight, proj[0].bias, proj[2].weight, proj[2].bias)])
            return res[0], mixed_mask, res[1], res[2]
        if self._grouped() and qa_embed.dtype == torch.float32 and ctx_embed.shape[2] <= 64:
            proj = self.c2q_down_projection
            res = self._try_group(lambda seeds: groups.qa_ctx(qa_embed, ctx_embed, qa_mask, ctx_mask, self.scale, p, seeds,
                                                              [self._g(w) for w in (proj[0].weight, proj[0].bias, proj[2].weight,
                                                                                    proj[2].bias)]), 3)
            if res is not None:
                return res[0], mixed_mask, res[1], res[2]
        u_a, raw_s, s_norm = ops.structured_attention(qa_embed, ctx_embed, qa_mask, ctx_mask, self.scale, p=p,
                                                      seed_c=self._seed(), seed_q=self._seed())
        proj = self.c2q_down_projection
        z = ops.cat3_layernorm(qa_embed.reshape(N * NA * Lqa, D), u_a.view(-1, D), proj[0].weight, proj[0].bias,
                               rep=Li, inner=Lqa, p=p, seed=self._seed())
        mixed = ops.linear(z, proj[2].weight, proj[2].bias, relu=True).view(N, NA, Li, Lqa, D)
        return mixed, mixed_mask, raw_s, s_norm

    # ---- span proposals (model/stage.py:389-467, model/model_utils.py:37-123) ----------------------------------
    @staticmethod
    def _best_span(p_st, p_ed):
        """Upper-triangular arg max of p_st (x) p_ed per row, on device.  (R, Li) x2 -> st, ed, conf (R,)."""
        R, Li = p_st.shape
        prod = torch.triu(p_st.unsqueeze(2) * p_ed.unsqueeze(1))
        conf, flat = prod.view(R, -1).max(dim=1)
        return flat // Li, flat % Li, conf

    def get_proposals(self, max_statement, max_statement_mask, temporal_scores, targets, ts_labels,
                      iou_thd=0.5, ce_prob_thd=0.01, extra_span_length=3, gt_scores_fn=None):
        N, NA, Li, D = max_statement.shape
        x = max_statement.reshape(N * NA, Li, D)
        m = max_statement_mask.reshape(N * NA, Li)
        if self.training:
            # ground-truth candidate's scores (:408-409); with the candidates of an example spread over a rank group they
            # come from the rank that holds it (parallel.CandidateLayout.gt_scores)
            gt = (gt_scores_fn(temporal_scores, targets) if gt_scores_fn is not None
                  else temporal_scores[torch.arange(N, device=targets.device), targets])
            ca = F.softmax(gt.detach(), dim=1)
            st, ed, conf = self._best_span(ca[:, :, 0], ca[:, :, 1])
            # one small D2H copy per step: the number of proposals (N_new) is data dependent by construction.  It is
            # requested BEFORE the global masked max is queued and awaited through an event: the host resumes as soon as the
            # 5 x N floats have arrived and issues the proposal kernels while that 0.1 ms kernel still runs.
            dev_spans = torch.stack([st.float(), ed.float(), conf, ts_labels["st"].float(), ts_labels["ed"].float()])
            if dev_spans.is_cuda:
                if self._span_host is None or self._span_host.shape != dev_spans.shape:
                    self._span_host = torch.empty(dev_spans.shape, dtype=torch.float32, pin_memory=True)
                with torch.cuda.device(dev_spans.device):    # the model may live on a device that is not the current one
                    self._span_host.copy_(dev_spans, non_blocking=True)
                    arrived = torch.cuda.Event()
                    arrived.record(torch.cuda.current_stream(dev_spans.device))
                    glob = ops.masked_max(x, m)                                   # (N*5, D)
                    arrived.synchronize()
                host = self._span_host.tolist()
            else:   # CPU tensors only reach this point in host-logic tests
                glob = ops.masked_max(x, m)
                host = dev_spans.tolist()
            src, wins = [], []
            for n in range(N):
                gs, ge = int(host[3][n]), int(host[4][n]) + 1
                spans = [(gs, ge)]
                if host[2][n] >= ce_prob_thd:
                    ps, pe = int(host[0][n]), int(host[1][n]) + 1
                    inter = max(0, min(pe, ge) - max(ps, gs))
                    union = max(pe, ge) - min(ps, gs)
                    if union != 0 and inter / union >= iou_thd:
                        spans.append((ps, pe))
                for (s, e) in spans:
                    src.append(n)
                    wins.append((max(0, s - extra_span_length), e + extra_span_length))
            src_t = torch.tensor(src, device=x.device, dtype=torch.long)
            win_t = torch.tensor(wins, device=x.device, dtype=torch.int32)       # (N_new, 2)
            # index_select (backward: one index_add; at most two contributions per slot, so the order cannot matter)
            # instead of advanced indexing (backward: sort-based index_put, five small kernels each)
            xg = max_statement.index_select(0, src_t).reshape(-1, Li, D)          # (N_new*5, Li, D)
            mg = max_statement_mask.reshape(N, NA, Li).index_select(0, src_t).reshape(-1, Li)
            wg = win_t.unsqueeze(1).expand(-1, NA, -1).reshape(-1, 2).contiguous()
            loc = ops.masked_max(xg, mg, wg).view(-1, NA, D)
            pooled = torch.cat([loc, glob.view(N, NA, D).index_select(0, src_t)], dim=-1)   # (N_new, 5, 2D)
            return pooled, targets.index_select(0, src_t)
        glob = ops.masked_max(x, m)                                               # (N*5, D)
        ts = F.softmax(temporal_scores, dim=2).view(N * NA, Li, 2)
        st, ed, _ = self._best_span(ts[:, :, 0], ts[:, :, 1])
        win = torch.stack([(st - extra_span_length).clamp(min=0), ed + 1 + extra_span_length], dim=1).int().contiguous()
        loc = ops.masked_max(x, m, win)
        return torch.cat([loc, glob], dim=-1).view(N, NA, 2 * D), targets

    def _linear_wrapper(self, x, lw: _LinearWrapperParams, res=None):
        y, s = self._ln(x, lw.conv[0], drop=True, res=res)
        return ops.linear(y, lw.conv[2].weight, lw.conv[2].bias, relu=lw.relu), s

    def classfier_head_multi_proposal(self, statement, statement_mask, targets, ts_labels, ts_labels_mask,
                                      extra_span_length=3, gt_scores_fn=None, pool_mask_factors=None, lay=None, qa_mask=None):
        """model/stage.py:484-537.  ``lay``: ``statement`` holds the compact rows of a ragged layout, (U, D)."""
        N, NA, Li, Lqa = statement_mask.shape
        D = statement.shape[-1]
        m = statement_mask.reshape(N * NA * Li, Lqa).contiguous()
        if lay is not None and lay.tab.halo >= Lqa:
            # every word of a live frame is kept (self-attention in the classifier encoder mixes all of them, or several blocks): the
            # compact rows are S whole sequences of Lqa words -- the ordinary encoder path on (S, Lqa, D), dead frames filled in behind
            seq = lay.seq.view(-1, 4)
            seq_g, seq_out = seq[:, 2].long(), seq[:, 3].long()
            ms = qa_mask.reshape(N * NA, Lqa).index_select(0, seq_g).contiguous()
            mx_c = self._stacked_encoder(statement.view(lay.S, Lqa, D), ms, self.cls_encoder, pool_mask=ms)
            mx = torch.full((N * NA * Li, D), NEG, dtype=mx_c.dtype, device=mx_c.device).index_copy(0, seq_out, mx_c)
        elif lay is not None:
            blk = self.cls_encoder.stacked_encoderBlocks[0]
            params = _block_params(blk)
            k = blk.conv[0].depthwise_conv.weight.shape[-1]
            mx = groups.encoder_block_rag(statement, blk.position_encoding.rows(Lqa), qa_mask.reshape(N * NA, Lqa).contiguous(), lay, k,
                                          self._p(), self._seeds((blk.n_conv + 1) // 2), [self._g(w) for w in params])
        else:
            x = statement.reshape(N * NA * Li, Lqa, D)
            mx = self._stacked_encoder(x, m, self.cls_encoder, pool_mask=m)            # encoder + :503 (max over the words)
        if pool_mask_factors is not None:
            # :504 from the factors of the statement mask (qa word mask x frame validity): any word valid AND the frame valid --
            # two reductions over KBs instead of one over the 154 MB (N, 5, Li, Lqa) mask
            qa_any, frame_any = pool_mask_factors
            mx_mask = (qa_any.view(N, NA, 1) & frame_any.view(N, 1, Li)).float().view(N, NA, Li, 1)
        else:
            mx_mask = (m.sum(1) != 0).float().view(N, NA, Li, 1)                         # :504
        enc = mx.view(N * NA * Li, D)
        # residual_temporal_predictor, layer 0 (:469-482).  Layers >= 1 (t_iter > 0) never reach any output or
        # gradient because of the `[:1]` slice at :516 (0.5*(t0 + mean([t0])) == t0 exactly), so they are skipped.
        st_lw, ed_lw = self.temporal_scoring_st_layers[0], self.temporal_scoring_ed_layers[0]
        res = None
        if self._grouped() and enc.dtype == torch.float32:
            pj = self.cls_projection_layers[0]
            res = self._try_group(lambda seeds: groups.temporal_head(
                enc, self._p(), seeds, [pj.conv[0].weight, pj.conv[0].bias, pj.conv[2].weight, pj.conv[2].bias,
                                        st_lw.conv[0].weight, st_lw.conv[0].bias, st_lw.conv[2].weight, st_lw.conv[2].bias,
                                        ed_lw.conv[0].weight, ed_lw.conv[0].bias, ed_lw.conv[2].weight, ed_lw.conv[2].bias]), 3)
        if res is not None:
            first, t_st, t_ed = res
        else:
            h, _ = self._linear_wrapper(enc, self.cls_projection_layers[0])
            t_st, first = self._linear_wrapper(h, st_lw, res=enc)                        # first = enc + h
            t_ed, _ = self._linear_wrapper(first, ed_lw)
        # the head-glue kernels (csrc/groups.hip) take Li <= 2048 frames and 2 * hsz <= 1024 columns; past that the per-op branches
        # below run (pre-checked here: those entry points would decline the shape in the middle of the step)
        grouped_tail = res is not None and Li <= 2048 and 2 * D <= 1024
        if grouped_tail:
            t_scores = groups.tscores(t_st, t_ed, ts_labels_mask.reshape(N, Li), N, NA, Li)  # cat + :521 mask_logits, one kernel
        else:
            t_scores = torch.cat([t_st, t_ed], dim=-1).view(N, NA, Li, 2).float()   # scores / losses are fp32 in every storage mode
            tm = ts_labels_mask.view(N, 1, Li, 1)
            t_scores = t_scores * tm + (1 - tm) * NEG                                    # :521 mask_logits
        first = first.view(N, NA, Li, D)
        if (grouped_tail and self.add_local and self.training and gt_scores_fn is None and NA == self.num_a
                and first.is_cuda):
            logits, targets = self._proposals_grouped(first, mx_mask, t_scores, targets, ts_labels, extra_span_length)
            return logits.view(-1, NA), targets, t_scores
        if self.add_local:
            pooled, targets = self.get_proposals(first, mx_mask, t_scores, targets, ts_labels,
                                                 extra_span_length=extra_span_length, gt_scores_fn=gt_scores_fn)
        else:
            pooled = ops.masked_max(first.view(N * NA, Li, D), mx_mask.view(N * NA, Li)).view(N, NA, D)
        logits, _ = self._linear_wrapper(pooled.reshape(-1, pooled.shape[-1]), self.classifier)
        return logits.view(-1, NA).float(), targets, t_scores

    def _proposals_grouped(self, first, mx_mask, t_scores, targets, ts_labels, extra_span_length,
                           iou_thd=0.5, ce_prob_thd=0.01):
        """get_proposals (training, model/stage.py:406-438) + classifier (:536) on the K-group path: one span kernel before the
        per-step read-back, ONE pooling + classifier group after it (csrc/groups.hip, G6), the proposal bookkeeping
        (source example, frame window, inverse map) uploaded as one int32 tensor through pinned staging."""
        from .att_host import PinnedStage
        N, NA, Li, D = first.shape
        dev = first.device
        spans = groups.gt_spans(t_scores, targets, ts_labels["st"], ts_labels["ed"])         # (6, N) device floats
        if self._span_host is None or self._span_host.shape != spans.shape:
            self._span_host = torch.empty(spans.shape, dtype=torch.float32, pin_memory=True)
        x = first.reshape(N * NA, Li, D)
        m = mx_mask.reshape(N * NA, Li)
        with torch.cuda.device(dev):
            self._span_host.copy_(spans, non_blocking=True)
            arrived = torch.cuda.Event()
            arrived.record(torch.cuda.current_stream(dev))
            glob, idx_g = groups.masked_max_raw(x, m)       # queued behind the copy: runs while the host reads the spans
            arrived.synchronize()
        host = self._span_host.tolist()
        src, wins, inv = [], [], [-1] * (2 * N)
        for n in range(N):
            gs, ge = int(host[3][n]), int(host[4][n]) + 1
            cand = [(gs, ge)]
            if host[2][n] >= ce_prob_thd:
                ps, pe = int(host[0][n]), int(host[1][n]) + 1
                inter = max(0, min(pe, ge) - max(ps, gs))
                union = max(pe, ge) - min(ps, gs)
                if union != 0 and inter / union >= iou_thd:
                    cand.append((ps, pe))
            for j, (s0, e0) in enumerate(cand):
                inv[2 * n + j] = len(src)
                src.append(n)
                wins += [max(0, s0 - extra_span_length), e0 + extra_span_length]
        P = len(src)
        if self._meta_stage is None:
            self._meta_stage = PinnedStage()
        # the repeated targets travel with the bookkeeping (ATen's index_select switches kernels at 16 indices: the first step
        # with a 17th proposal paid ~9 ms of lazy kernel loading inside the training loop)
        tgt = [int(host[5][n]) for n in src]
        meta = self._meta_stage.upload(torch.tensor(src + wins + inv + tgt, dtype=torch.int32), dev)
        cl = self.classifier
        seeds = self._seeds(1)
        logits = groups.pool_classifier(x, m, glob, idx_g, meta, (N, NA, Li, D, P), self._p(), seeds,
                                        [cl.conv[0].weight, cl.conv[0].bias, cl.conv[2].weight, cl.conv[2].bias])
        return logits, meta[3 * P + 2 * N:].long()

    def get_ts_loss(self, temporal_scores, ts_labels, answer_indices, cand_offset: int = 0):
        """model/stage.py:539-555.  ``cand_offset``: global index of local candidate 0 when the candidates of an example
        are spread over ranks -- only examples whose ground-truth candidate is local contribute here (the sum over the
        ranks of the group is the full loss)."""
        bsz = len(answer_indices)
        NA_loc, Li = temporal_scores.shape[1:3]
        if self._grouped() and temporal_scores.is_cuda and temporal_scores.dtype == torch.float32 and 1 <= Li <= 2048 and bsz > 0:
            # loss and its gradient in one pass (csrc/groups.hip: ts_loss_kernel) instead of gather + 2 x (log-softmax, nll) + add
            try:
                return groups.ts_loss(temporal_scores, answer_indices, ts_labels["st"], ts_labels["ed"], cand_offset, self.num_a)
            except groups.Unsupported:
                pass
        local = answer_indices - cand_offset
        if cand_offset == 0 and NA_loc == self.num_a:
            ca = temporal_scores.gather(1, local.view(bsz, 1, 1, 1).expand(bsz, 1, Li, 2)).squeeze(1)   # [n, target_n]
            loss_st = self.temporal_criterion(ca[:, :, 0], ts_labels["st"])
            loss_ed = self.temporal_criterion(ca[:, :, 1], ts_labels["ed"])
            return (loss_st + loss_ed) / 2.
        here = ((local >= 0) & (local < NA_loc))
        idx = local.clamp(0, NA_loc - 1)
        ca = temporal_scores.gather(1, idx.view(bsz, 1, 1, 1).expand(bsz, 1, Li, 2)).squeeze(1)
        per = (F.cross_entropy(ca[:, :, 0], ts_labels["st"], reduction="none")
               + F.cross_entropy(ca[:, :, 1], ts_labels["ed"], reduction="none"))
        return (per * here.to(per.dtype)).sum() / 2.

    # ---- ragged token rows --------------------------------------------------------------------------------------
    def _get_mask_info(self, batch, N, NA, Lqa, Li, names):
        """{"qas": (N, NA, Lqa) bool, "<stream>_len": (N, Li) last valid position + 1} of this batch: the loader's host copies
        (batch.mask_host) or ONE read-back of the device masks; cached for the forward."""
        if self._mask_info is not None:
            return self._mask_info
        info = ragged.host_info(batch)
        if info is not None and (info["qas"].shape != (N, NA, Lqa)
                                 or any(info.get(k + "_len") is None or info[k + "_len"].shape != (N, Li) for k in names)):
            info = None                                  # no / stale host copies (a batch sliced by foreign code): read the masks
        if info is None:
            info = ragged.info_from_device(batch.qas_mask.view(N, NA, Lqa),
                                           {k: (batch.sub_mask if k == "sub" else batch.vid_mask).view(N, Li, -1) for k in names})
        self._mask_info = info
        return info

    def _ctx_buckets(self, batch, name, N, NA, Lqa, Li, L):
        """Length buckets of context stream ``name`` (ragged.bucket_plan) for the dense paths, or None: switched off, short rows
        (<= 64: the ragged group path's territory, and too few positions to sort), an input encoder with self-attention."""
        if not (self.use_ragged_ctx and self.use_ctx_buckets) or L <= 64:
            return None
        iblocks = list(self.input_encoder.stacked_encoderBlocks)
        if not iblocks or any(b.num_heads != 0 or b.n_conv < 1 for b in iblocks):
            return None
        halo = sum(ragged.conv_halo(1, b.n_conv, b.conv[0].depthwise_conv.weight.shape[-1]) for b in iblocks)
        names = (["sub"] if self.sub_flag else []) + (["vid"] if self.vfeat_flag else [])
        info = self._get_mask_info(batch, N, NA, Lqa, Li, names)
        step = 64 if L >= 256 else 32
        plan = ragged.bucket_plan(info[name + "_len"], int(L), halo, step)
        if not plan or (len(plan) == 1 and plan[0][1] == L and len(plan[0][0]) == N * Li):
            return None if plan else []
        return plan

    def _base_encoder_buckets(self, plan, data, data_mask, init_encoder, downsize_encoder, input_encoder, l2_normalize):
        """base_encoder over the length buckets of ``plan``: (M, L, D) with zeros where nothing is computed (rows behind a frame's
        bucket length, frames without a valid position -- the attention masks every one of them, and no gradient comes back)."""
        M, L, _ = data.shape
        idxs, ys = [], []
        for idx, Lb in plan:
            idx_d = torch.from_numpy(idx).to(data.device, non_blocking=True)
            xb = data[:, :Lb].index_select(0, idx_d)                # (Mb, Lb, F): only these positions are read
            mb = data_mask[:, :Lb].index_select(0, idx_d)
            yb = self.base_encoder(xb, mb, init_encoder, downsize_encoder, input_encoder, l2_normalize=l2_normalize)
            idxs.append(idx_d)
            ys.append(yb.view(len(idx), Lb, -1))
        if not ys:
            return torch.zeros(M, L, self.hsz, device=data.device, dtype=self.storage)
        return _ScatterBuckets.apply(M, L, idxs, *ys)

    def _ragged_layout(self, batch, qas_mask, a_embed):
        """The ragged layout of this batch (tvqaplus_amd/ragged.py), or None when the dense path runs: switched off, a configuration
        the ragged kernels do not cover (decided BEFORE anything is launched: hsz = 128 on the K-group path, one classifier-encoder
        block without self-attention, <= 40 QA words, even region / word counts <= 64), or no live row at all."""
        none = (None, {})
        if not (self.use_ragged and self._grouped() and self.fuse_ln_dwconv and self.fuse_ln_max and a_embed.is_cuda
                and a_embed.dtype == torch.float32):
            return none
        N, NA, Lqa, D = a_embed.shape
        blocks = list(self.cls_encoder.stacked_encoderBlocks)
        # Small batches (a rank of a strong-scaled job holds 1-2 examples): the step is bound by the HOST there, and the ragged layout
        # costs the host its tables, three uploads and six small launches (0.9 ms of a 5.4 ms step at 2 examples, measured) to save device
        # time nobody waits for -- the padded rows run instead (the same function: tests hold the two paths equal).
        li_any = batch.vid.shape[1] if self.vfeat_flag else batch.sub_bert.shape[1]
        if N * NA * li_any * Lqa < self.ragged_min_rows:
            return none
        if D != 128 or not (4 <= Lqa <= 40) or self.bridge_hsz % 4 or self.bridge_hsz > 1024:
            return none          # (the ragged input-MLP group's own limits: stage_grp_input_mlp_rag_fwd declines other widths)
        # words kept behind the last valid one: the classifier encoder's convolution halo -- or every word of a live frame when that
        # encoder is not a single conv-only block (self-attention mixes all words; the frames that are dead are still skipped)
        cls_halo = Lqa
        if len(blocks) == 1 and blocks[0].num_heads == 0 and 1 <= blocks[0].n_conv <= 8:
            k = blocks[0].conv[0].depthwise_conv.weight.shape[-1]
            if k % 2 == 1 and k <= 9:
                cls_halo = min(Lqa, ragged.conv_halo(1, blocks[0].n_conv, k))
        streams = []
        if self.sub_flag:
            streams.append((batch.sub_mask, batch.sub_bert.shape[1], batch.sub_bert.shape[2]))
        if self.vfeat_flag:
            streams.append((batch.vid_mask, batch.vid.shape[1], batch.vid.shape[2]))
        if not streams or any(Li != streams[0][1] for _, Li, _ in streams):
            return none
        Li = streams[0][1]
        # the statement mask's frame side comes from the video stream when there is one (model/stage.py:283-289)
        frame_stream = "vid" if self.vfeat_flag else "sub"
        names = (["sub"] if self.sub_flag else []) + (["vid"] if self.vfeat_flag else [])
        info = self._get_mask_info(batch, N, NA, Lqa, Li, names)
        tab = ragged.RaggedTables(info["qas"], info[frame_stream + "_len"] > 0, cls_halo)
        if tab.U == 0:
            return none
        lib = groups._lib.load()
        lib_ok = all(bool(lib.stage_grp_qa_ctx_rag_supported(N, NA, Li, Lqa, int(Lr), D, tab.U, tab.Fc)) for _, _, Lr in streams)
        # ... and the fused [a, b, a*b] kernels on gathered rows, which the attention and the fusion groups of this layout are built on
        # (switched off by STAGE_NO_CAT3_FUSED: the dense path then runs -- decided here, before anything is launched)
        lib_ok = lib_ok and bool(lib.stage_cat3_ln_gemm_fwd_rag_supported(tab.U, N * NA * Lqa, tab.Fc, D))
        if not lib_ok:
            return none
        lay = ragged.RaggedLayout(tab, a_embed.device, self._rag_stage)
        self._rag_stage = lay.stage
        # the context streams themselves: valid words / regions + the halo of the INPUT encoder's convolutions (no self-attention there,
        # feature widths the gathered LayerNorm takes)
        clays = {}
        iblocks = list(self.input_encoder.stacked_encoderBlocks)
        ctx_ok = self.use_ragged_ctx and not self.fuse_input_ln
        conv_only = len(iblocks) >= 1 and all(b.num_heads == 0 and 1 <= b.n_conv <= 8 for b in iblocks)
        if conv_only:
            ik = iblocks[0].conv[0].depthwise_conv.weight.shape[-1]
            conv_only = ik % 2 == 1 and ik <= 9 and all(b.conv[0].depthwise_conv.weight.shape[-1] == ik for b in iblocks)
        if ctx_ok:
            # conv-only input encoder: valid positions + its halo; otherwise (self-attention) whole frames, dead frames skipped
            halo = sum(ragged.conv_halo(1, b.n_conv, ik) for b in iblocks) if conv_only else 1 << 20
            for (mask, _, L), name in zip(streams, names):
                feat = batch.sub_bert if name == "sub" else batch.vid
                if feat.dtype != torch.float32 or feat.shape[-1] % 4 or feat.shape[-1] > 1024 or not feat.is_contiguous():
                    continue
                ct = ragged.CtxTables(info[name + "_len"], int(L), halo)
                if ct.U == 0:
                    continue
                st = self._ctx_stage.get(name)
                clays[name] = ragged.CtxLayout(ct, a_embed.device, st)
                self._ctx_stage[name] = clays[name].stage
        return lay, clays

    # ---- forward ----------------------------------------------------------------------------------------------
    def forward(self, batch):
        if getattr(self, "_is_replica", False):
            # nn.DataParallel (main.py:204-206) replicates the module into THREADS of one process; the HIP path keeps per-process
            # launch state (scratch buffers, ticket rings) and is built for one process per GPU over RCCL instead
            raise RuntimeError("tvqaplus_amd.STAGE does not run under nn.DataParallel: launch one process per GPU "
                               "(python -m torch.distributed.run ...) and use tvqaplus_amd.parallel (INTEGRATION.md section 4)")
        if self.inference_mode:
            return self.forward_main(batch)
        out, att_loss, att_predictions, temporal_loss, temporal_predictions, _ = self.forward_main(batch)
        return out, att_loss, att_predictions, temporal_loss, temporal_predictions

    def _g(self, w):
        """The parameter, or its alias of this step when its module is gated (groups.gate)."""
        return self._gate_map.get(id(w), w)

    def _stack_qkv(self):
        """The stacked Q / K / V projection weights of every attention block of this step (model/self_attention.py:35-44 as one
        Linear(D -> 3D)): built here, on the main stream, from the step's gate aliases."""
        self._qkv_cache = {}
        for enc in (self.input_encoder, self.cls_encoder):
            for blk in enc.stacked_encoderBlocks:
                if blk.num_heads != 0:
                    mha = blk.multi_head_attn
                    self._qkv_cache[id(mha)] = (torch.cat([self._g(mha.linears[j].weight) for j in range(3)], dim=0),
                                                torch.cat([self._g(mha.linears[j].bias) for j in range(3)], dim=0))

    def _open_gates(self):
        # modules applied to two or three streams (model/stage.py:226-269): their parameter gradients leave the graph once
        self._gate_map = {}
        # (both paths: the K-groups deliver into the sinks, and so do the per-kernel LayerNorm / Linear / LayerNorm->dwconv ops --
        # the bf16 storage mode runs the shared encoder once per stream and length bucket)
        if not (self.gate_shared and self.training and torch.is_grad_enabled()):
            return
        mods = [self.bert_word_encoding_fc, self.input_embedding, self.input_encoder]
        if self.flag_cnt == 2:
            mods.append(self.c2q_down_projection)
        for m in mods:
            # (cached: m.parameters() walks named_modules() on every call.  Keyed on the parameter COUNT of the module's direct
            # children and the requires_grad flags: a parameter added, replaced by surgery or (un)frozen rebuilds the list)
            cached = m.__dict__.get("_stage_gated")
            ps = None
            if cached is not None:
                ps, owners = cached
                for w, (params, key) in zip(ps, owners):
                    if params.get(key) is not w or not w.requires_grad:
                        ps = None
                        break
                if ps is not None and m.__dict__.get("_stage_gated_frozen"):
                    ps = None if any(w.requires_grad for w in m.__dict__["_stage_gated_frozen"]) else ps
            if ps is None:
                ps, owners, frozen = [], [], []
                for sub in m.modules():
                    for key, w in sub._parameters.items():
                        if w is None:
                            continue
                        if w.requires_grad:
                            ps.append(w)
                            owners.append((sub._parameters, key))
                        else:
                            frozen.append(w)
                m.__dict__["_stage_gated"] = (ps, owners)
                m.__dict__["_stage_gated_frozen"] = frozen
            if ps:
                for w, a in zip(ps, groups.gate(ps)):
                    self._gate_map[id(w)] = a

    def forward_main(self, batch):
        """model/stage.py:199-348."""
        try:
            self._open_gates()
            self._stack_qkv()
            return self._forward_main(batch)
        finally:
            self._gate_map = {}
            self._qkv_cache = {}

    def _forward_main(self, batch):
        ops.new_step()
        self.bsz = len(batch.qid)
        N, D = self.bsz, self.hsz
        NA = batch.qas_bert.shape[1]          # 5 (self.num_a), or the local candidates of a candidate-sharded batch
        cand_offset = int(_opt(batch, "cand_offset", 0) or 0)
        gt_scores_fn = _opt(batch, "gt_scores_fn", None)
        dr = _opt(batch, "dropout_rank", None)
        if dr is not None and (self._seed_state is None or int(dr) != self._dropout_rank):
            # the example block of this rank (parallel.CandidateLayout) -- it may change when the layout is rebuilt for another batch
            # size: the ranks that now share a block must share a dropout stream again, so the stream is re-derived from the seed
            if self._seed_state is not None:
                self._seed_state = None
            self._dropout_rank = int(dr)
        qas_mask = batch.qas_mask.view(N, NA, -1).float()
        # (all masks converted HERE, on the caller's stream and in front of the first join: a cast issued later on the main stream would
        # not be ordered against a branch stream that reads it)
        sub_mask_f = batch.sub_mask.float() if self.sub_flag else None
        vid_mask_f = batch.vid_mask.float() if self.vfeat_flag else None
        dev = batch.qas_bert.device
        streams = int(self.use_streams) if dev.type == "cuda" else 0
        if streams > 3 and os.environ.get("STAGE_STREAMS_UNSAFE4") is None and not (self._grouped() and self.storage == torch.float32 and self.input_encoder.stacked_encoderBlocks[0].num_heads == 0):
            # level 4 (two attention forwards side by side) only on the fp32 K-group path, where steps reproduce level 0 bit for bit; on the
            # per-kernel path the stress config's loss differed from run to run at level 4 (an ordering problem that was not found), while
            # levels <= 3 reproduce level 0 there
            streams = 3
        main = s_qa = s_vid = None
        if streams:
            main = torch.cuda.current_stream(dev)
            if dev not in _SIDE_STREAMS:      # (per device, not per model: a Stream object inside the module would not survive copy.deepcopy)
                _SIDE_STREAMS[dev] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
            s_qa, s_vid = _SIDE_STREAMS[dev]
            s_qa.wait_stream(main)                 # the batch (and last step's parameter update) is ready on the main stream
        with (torch.cuda.stream(s_qa) if streams else contextlib.nullcontext()):
            a_embed = self.base_encoder(batch.qas_bert.view(N * NA, -1, self.wd_size), qas_mask.view(N * NA, -1),
                                        self.bert_word_encoding_fc, self.input_embedding, self.input_encoder)
            a_embed = a_embed.view(N, NA, -1, D)
        attended_sub = attended_vid = attended_vid_mask = attended_sub_mask = None
        other_outputs: Dict[str, torch.Tensor] = {}
        self._mask_info = None
        lay, clays = self._ragged_layout(batch, qas_mask, a_embed)
        self.last_ragged, self.last_ragged_ctx = lay, clays
        self.last_buckets = {}
        if streams:
            s_vid.wait_stream(main)                # the layouts' tables are uploaded / expanded on the main stream
            # Everything allocated on the main stream that a branch stream reads -- in the forward or, through saved tensors, in the
            # backward (autograd runs a node on its forward's stream): the float masks and the layouts' device tables.  Registered with
            # the allocator for both side streams, so that a block whose last reference is dropped by a side-stream node is not handed
            # to the main stream's next allocation while that node's kernel still reads it (the statement / video
            # embeddings and the level-3 outputs are registered where they cross).
            shared = [qas_mask, sub_mask_f, vid_mask_f]
            if lay is not None:
                shared += lay.device_tensors()
            for cl_ in clays.values():
                shared += cl_.device_tensors()
            for t in shared:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(s_qa)
                    t.record_stream(s_vid)
        if self.sub_flag:
            Li, Lw = batch.sub_bert.shape[1:3]
            sub_mask = sub_mask_f.view(N, Li, Lw)
            cl = clays.get("sub")
            plan = self._ctx_buckets(batch, "sub", N, NA, qas_mask.shape[-1], Li, Lw) if cl is None else None
            if plan is not None:
                self.last_buckets["sub"] = [(len(ix), lb) for ix, lb in plan]
                sub_embed = self._base_encoder_buckets(plan, batch.sub_bert.view(N * Li, Lw, -1), sub_mask.view(N * Li, Lw),
                                                       self.bert_word_encoding_fc, self.input_embedding, self.input_encoder, False)
            else:
                sub_embed = self.base_encoder(batch.sub_bert.view(N * Li, Lw, -1), sub_mask.view(N * Li, Lw),
                                              self.bert_word_encoding_fc, self.input_embedding, self.input_encoder, clay=cl)
            if streams:
                main.wait_stream(s_qa)             # the statement embedding
                a_embed.record_stream(main)
            attended_sub, attended_sub_mask, raw, norm = self.qa_ctx_attention(
                a_embed, sub_embed if cl is not None else sub_embed.view(N, Li, Lw, D), qas_mask, sub_mask, lay, cl)
            other_outputs["sub_normalized_s"], other_outputs["sub_raw_s"] = norm, raw
        if self.vfeat_flag:
            Li, Lr = batch.vid.shape[1:3]
            vid_mask = vid_mask_f.view(N, Li, Lr)
            cl = clays.get("vid")
            plan = self._ctx_buckets(batch, "vid", N, NA, qas_mask.shape[-1], Li, Lr) if cl is None else None
            with (torch.cuda.stream(s_vid) if streams >= 2 else contextlib.nullcontext()):
                if plan is not None:
                    self.last_buckets["vid"] = [(len(ix), lb) for ix, lb in plan]
                    vid_embed = self._base_encoder_buckets(plan, batch.vid.view(N * Li, Lr, -1), vid_mask.view(N * Li, Lr), self.vid_fc,
                                                           self.input_embedding, self.input_encoder, True)
                else:
                    vid_embed = self.base_encoder(batch.vid.view(N * Li, Lr, -1), vid_mask.view(N * Li, Lr), self.vid_fc,
                                                  self.input_embedding, self.input_encoder, l2_normalize=True, clay=cl)
                if streams >= 3:
                    s_vid.wait_stream(s_qa)
                    if streams == 3:
                        s_vid.wait_stream(main)    # forward fenced behind the subtitle attention; the backward still overlaps
                    a_embed.record_stream(s_vid)
                    attended_vid, attended_vid_mask, raw, norm = self.qa_ctx_attention(
                        a_embed, vid_embed if cl is not None else vid_embed.view(N, Li, Lr, D), qas_mask, vid_mask, lay, cl)
            if streams >= 2:
                main.wait_stream(s_vid)
            if streams:
                main.wait_stream(s_qa)
                a_embed.record_stream(main)
            if streams >= 3:
                for t in (attended_vid, attended_vid_mask, raw, norm):
                    if torch.is_tensor(t):
                        t.record_stream(main)
            else:
                if streams == 2:
                    vid_embed.record_stream(main)
                attended_vid, attended_vid_mask, raw, norm = self.qa_ctx_attention(
                    a_embed, vid_embed if cl is not None else vid_embed.view(N, Li, Lr, D), qas_mask, vid_mask, lay, cl)
            other_outputs["vid_normalized_s"], other_outputs["vid_raw_s"] = norm, raw
        if self.flag_cnt == 2:
            fc = self.concat_fc
            z = None
            if self._grouped() and attended_sub.dtype == torch.float32:
                z = self._try_group(lambda seeds: groups.concat_fc(
                    attended_sub.view(-1, D), attended_vid.view(-1, D), self._p(), seeds,
                    [fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias]), 1)
            if z is None:
                z = ops.cat3_layernorm(attended_sub.view(-1, D), attended_vid.view(-1, D), fc[0].weight, fc[0].bias,
                                       p=self._p(), seed=self._seed())
                z = ops.linear(z, fc[2].weight, fc[2].bias, relu=True)
                z, _ = self._ln(z, fc[4])
            statement, statement_mask = z.view(attended_vid.shape), attended_vid_mask
        elif self.sub_flag:
            statement, statement_mask = attended_sub, attended_sub_mask
        elif self.vfeat_flag:
            statement, statement_mask = attended_vid, attended_vid_mask
        else:
            raise NotImplementedError
        # supervised attention loss, random-negative mode: the (positive, negative) index pairs need no scores.  They are
        # built HERE -- the encoders, both attentions and the fusion (most of the forward's device time) are queued, and
        # the host is about to wait for the device at the proposal read-back anyway: ~10 ms of slack, the 3-11 ms of index
        # building hide completely.  Built at the top of the step they left the device idle for ~2 ms (the previous
        # step's queue drained first); the reference builds them after the forward, on the critical path.
        att_pairs = None
        if (self.use_sup_att and self.training and self.vfeat_flag and not self.inference_mode
                and not bool(_opt(batch, "use_hard_negatives", False)) and _opt(batch, "att_pairs", None) is None):
            from .att_host import AttPairs, build_att_pairs, targets_on_device_ok
            on_dev = targets_on_device_ok(self, batch, NA)       # no host copy of the answers: their offset is added on the device
            pos, neg = build_att_pairs(self, batch, None, n_local_candidates=NA, placeholder_targets=on_dev)
            if pos is not None:
                Li_v, Lr_v = batch.vid.shape[1:3]
                att_pairs = AttPairs(pos, neg, (N, NA, Li_v, batch.qas_mask.shape[-1], Lr_v), batch.vid.device,
                                     getattr(self, "_att_stage", None), target_dev=batch.target if on_dev else None)
                self._att_stage = att_pairs.stage
        ctx_m = vid_mask if self.vfeat_flag else sub_mask       # the statement mask's context side (model/stage.py:386)
        factors = ((qas_mask != 0).any(-1), ctx_m.sum(-1) != 0)
        out, target, t_scores = self.classfier_head_multi_proposal(
            statement, statement_mask, batch.target, batch.ts_label, batch.ts_label_mask.float(),
            extra_span_length=self.extra_span_length, gt_scores_fn=gt_scores_fn, pool_mask_factors=factors, lay=lay, qa_mask=qas_mask)
        assert len(out) == len(target)
        other_outputs["temporal_scores"] = t_scores

        if self.inference_mode:
            from .att_host import get_att_prediction
            return {
                "answer": out,
                "t_scores": F.softmax(t_scores, dim=2),
                "att_predictions": get_att_prediction(
                    scores=other_outputs["vid_raw_s"], object_vocab=batch.eval_object_word_ids, words=batch.qas,
                    vid_names=batch.vid_name, qids=batch.qid, img_indices=batch.image_indices, boxes=batch.boxes,
                    start_indices=batch.anno_st_idx) if self.vfeat_flag else None,
            }

        att_loss = 0
        att_predictions = None
        if self.use_sup_att and self.training and self.vfeat_flag:
            from .att_host import get_att_loss
            att_loss, att_predictions = get_att_loss(self, other_outputs["vid_raw_s"], batch, pairs=att_pairs)
        temporal_loss = self.get_ts_loss(t_scores, batch.ts_label, batch.target, cand_offset)
        if self.training:
            return [out, target], att_loss, att_predictions, temporal_loss, t_scores, other_outputs
        return out, att_loss, att_predictions, temporal_loss, F.softmax(t_scores, dim=2), other_outputs
