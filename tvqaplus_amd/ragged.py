"""Ragged token rows: which rows of the ``(N, 5, Li, Lqa, D)`` tensors of STAGE are ever used, as index tables for the HIP kernels.

The reference computes every padded row of those tensors (model/stage.py:365-387 ``qa_ctx_attention``, :276-279 ``concat_fc``,
:484-505 the classifier head; no mask inside LayerNorm / Linear / the convolutions of model/encoder.py:35-52, model/cnn.py:42-47).
What reaches an output or a gradient is less (DESIGN.md "ragged token rows", csrc/ragged.hip):

* **dead frames** -- model/stage.py:503 takes ``max over words of (statement * mask + (1 - mask) * -1e10)``.  A frame whose statement
  mask (QA word mask x "the frame has a valid region / word", model/stage.py:386) is all zero pools to the constant -1e10 whatever the
  classifier encoder computed, and the gradient that comes back is ``dout * mask = 0``: nothing of that frame is used;
* **dead words** -- of a live frame the words ``w < Lv`` (``Lv`` = last valid word + 1) are used, and the classifier encoder's
  depthwise convolutions (``n_conv`` layers of width ``k``, unmasked, zero padding only at the ends of the ``Lqa`` axis) let the words
  up to ``Lv + halo - 1``, ``halo = n_blocks * n_conv * (k // 2)``, leak into them -- in the forward and, symmetrically, in the
  backward.  Words at and behind ``Lc = min(Lqa, Lv + halo)`` never reach a valid word.

``RaggedLayout`` holds the tables (one int32 upload per batch); layouts:

* compact rows ``[group g = (n, a)][live frame][word < Lc(g)]`` -- every row kernel behind the attention;
* frame-compact rows ``[sequence = first(n) + a * slots(n) + slot][word < Lqa]`` -- the attention output A and its gradient
  (``slots(n)`` = live frames of example n + one dump slot for its dead frames).

Exactness: rows in ``[Lv + halo - ..., Lc)`` see zero padding where the reference sees more padded words, so THEIR values differ
from the reference's -- but only in positions whose receptive field never contains a valid word and whose gradient is exactly zero;
everything that is returned (logits, span scores, attention maps, losses, every parameter gradient) is computed from the same
numbers as on the dense path (tests/test_ragged.py, tests/test_hip_ragged.py).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib

CAP_STEP = 16384      # arenas are sized for the live rows rounded up to this: a handful of distinct allocation sizes per run


def conv_halo(n_blocks: int, n_conv: int, kernel_size: int) -> int:
    """Words behind the last valid one that reach it through the classifier encoder's depthwise convolutions."""
    return int(n_blocks) * int(n_conv) * (int(kernel_size) // 2)


class RaggedTables:
    """Host side (numpy, no device): the tables of one batch.  ``qa_valid`` (N, NA, Lqa) bool, ``frame_live`` (N, Li) bool."""

    def __init__(self, qa_valid: np.ndarray, frame_live: np.ndarray, halo: int):
        qa_valid = np.ascontiguousarray(qa_valid, dtype=bool)
        frame_live = np.ascontiguousarray(frame_live, dtype=bool)
        N, NA, Lqa = qa_valid.shape
        Li = frame_live.shape[1]
        assert frame_live.shape[0] == N
        self.N, self.NA, self.Li, self.Lqa, self.halo = N, NA, Li, Lqa, int(halo)
        G = N * NA
        anyv = qa_valid.any(-1)
        last = Lqa - np.argmax(qa_valid[..., ::-1], axis=-1)              # last valid word + 1 where any
        Lv = np.where(anyv, last, 0).reshape(G)
        Lc = np.where(Lv > 0, np.minimum(Lqa, Lv + int(halo)), 0).astype(np.int64)
        nlive = frame_live.sum(1).astype(np.int64)                        # (N,)
        slots = nlive + 1
        first = NA * np.concatenate([[0], np.cumsum(slots)[:-1]])         # first frame-compact sequence of example n
        slot_of_frame = np.where(frame_live, np.cumsum(frame_live, axis=1) - 1, -1)
        frame_of_slot = np.argsort(~frame_live, axis=1, kind="stable")    # live frames first, in order
        n_of_g = np.repeat(np.arange(N), NA)
        a_of_g = np.tile(np.arange(NA), N)
        rows_g = nlive[n_of_g] * Lc
        rowbase = np.concatenate([[0], np.cumsum(rows_g)[:-1]])
        fcseq0 = first[n_of_g] + a_of_g * slots[n_of_g]
        cnt = np.where(Lc > 0, nlive[n_of_g], 0)
        S = int(cnt.sum())
        g_of_s = np.repeat(np.arange(G), cnt)
        li_of_s = np.arange(S) - np.repeat(np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
        self.U = int(rows_g.sum())
        self.S = S
        self.seqs_fc = int(NA * slots.sum())                              # frame-compact sequences incl. the dump slots
        self.Fc = self.seqs_fc * Lqa                                      # frame-compact rows
        self.Lc, self.Lv, self.nlive = Lc, Lv, nlive
        self.fmap = np.concatenate([slot_of_frame.reshape(-1), slots, first]).astype(np.int32)
        self.gdesc = np.stack([rowbase, Lc, slots[n_of_g], fcseq0], axis=1).astype(np.int32)
        self.seq = np.stack([rowbase[g_of_s] + li_of_s * Lc[g_of_s], Lc[g_of_s], g_of_s,
                             g_of_s * Li + frame_of_slot[n_of_g[g_of_s], li_of_s]], axis=1).astype(np.int32)
        self.seqfc = (fcseq0[g_of_s] + li_of_s).astype(np.int32)

    def work_table(self, n_wg: int) -> np.ndarray:
        """Balanced work table of the fused [a, b, a*b] backward (csrc/cat3_fused.hip: cf_bwd_kernel MODE 3 with ``wtab``) for
        ``n_wg`` persistent workgroups: int32 [first segment of workgroup w: n_wg + 1 entries, padded to a multiple of 4]
        [(first slab, slabs) of every group][segments (group, first frame, end frame, workgroup)].
        The kernel walks a group frame by frame (Lc <= 32: one 32-row tile per frame) or in quads of frames (Lc > 32: five tiles per
        four frames); an ATOM is one such step, its cost 4 / 20 quarter-tiles.  All atoms, in (group, frame) order, are cut into
        n_wg runs of equal cost; a run that crosses a group boundary becomes several segments.  Every atom belongs to the run its
        midpoint falls into, so each is computed exactly once and a run is within half an atom of the mean; segments are in (workgroup, group) = (group, workgroup) order:
        the slabs of one group are consecutive."""
        G = self.N * self.NA
        frames = np.where(self.Lc > 0, np.repeat(self.nlive, self.NA), 0).astype(np.int64)
        big = self.Lc > 32
        step = np.where(big, 4, 1).astype(np.int64)
        cost = np.where(big, 20, 4).astype(np.int64)
        atoms = (frames + step - 1) // step
        P = np.concatenate([[0], np.cumsum(atoms * cost)])
        W = int(P[-1])
        wg_first = np.zeros(n_wg + 1, dtype=np.int64)
        gseg = np.zeros((G, 2), dtype=np.int64)
        seg = np.zeros((0, 4), dtype=np.int64)
        if W > 0:
            T = (W + n_wg - 1) // n_wg
            cuts = np.unique(np.concatenate([P[:-1][atoms > 0], np.arange(n_wg, dtype=np.int64) * T]))
            cuts = cuts[cuts < W]
            ends = np.concatenate([cuts[1:], [W]])
            g = np.searchsorted(P, cuts, side="right") - 1
            w = cuts // T
            # atom k of group g covers the cost units [P_g + k c, P_g + (k + 1) c): it goes to the run that holds its MIDPOINT
            k0 = np.maximum(0, (2 * (cuts - P[g]) - cost[g] + 2 * cost[g] - 1) // (2 * cost[g]))
            k1 = np.minimum(atoms[g], np.maximum(0, (2 * (ends - P[g]) - cost[g] + 2 * cost[g] - 1) // (2 * cost[g])))
            keep = k1 > k0
            g, w, k0, k1 = g[keep], w[keep], k0[keep], k1[keep]
            seg = np.stack([g, k0 * step[g], np.minimum(frames[g], k1 * step[g]), w], axis=1)
            wg_first = np.searchsorted(w, np.arange(n_wg + 1), side="left")
            gseg[:, 0] = np.searchsorted(g, np.arange(G), side="left")
            gseg[:, 1] = np.searchsorted(g, np.arange(G), side="right") - gseg[:, 0]
        pad4 = lambda a: np.concatenate([a, np.zeros((-a.size) % 4, dtype=a.dtype)])
        return np.concatenate([pad4(wg_first), pad4(gseg.reshape(-1)), seg.reshape(-1)]).astype(np.int32)

    # ---- reference semantics of the layouts (tests; never on the product path) ------------------------------------
    def compact_index(self) -> np.ndarray:
        """(U, 4) int64: (n, a, i, w) of every compact row, in order."""
        out = np.empty((self.U, 4), dtype=np.int64)
        for s in range(self.S):
            start, ln, g, dense = (int(v) for v in self.seq[s])
            n, a, i = g // self.NA, g % self.NA, dense - g * self.Li
            out[start:start + ln, 0], out[start:start + ln, 1], out[start:start + ln, 2] = n, a, i
            out[start:start + ln, 3] = np.arange(ln)
        return out

    def rowinfo_host(self) -> np.ndarray:
        """What stage_rag_rowinfo computes on the device."""
        out = np.empty((self.U, 4), dtype=np.int32)
        for s in range(self.S):
            start, ln, g, dense = (int(v) for v in self.seq[s])
            w = np.arange(ln)
            out[start:start + ln] = np.stack([g * self.Lqa + w, int(self.seqfc[s]) * self.Lqa + w, np.full(ln, dense), w], axis=1)
        return out


def _align(n: int, to: int = 4) -> int:
    return (n + to - 1) // to * to


class RaggedLayout:
    """Device side of ``RaggedTables``: one upload, the row-info table expanded by ``stage_rag_rowinfo``, the host array of table
    pointers the ``*_rag`` group calls take.  Keep it alive until the backward of the step has run (the autograd nodes hold it)."""

    def __init__(self, tab: RaggedTables, device, stage=None):
        self.tab = tab
        self.N, self.NA, self.Li, self.Lqa = tab.N, tab.NA, tab.Li, tab.Lqa
        self.U, self.S, self.Fc = tab.U, tab.S, tab.Fc
        self.Ucap = max(CAP_STEP, _align(tab.U, CAP_STEP))
        self.n_wg = 0
        wtab = np.zeros(0, dtype=np.int32)
        if torch.device(device).type == "cuda":
            self.n_wg = int(_lib.load().stage_cat3_rag_work_groups())
            wtab = tab.work_table(self.n_wg)
        parts = [tab.fmap, tab.gdesc.reshape(-1), tab.seq.reshape(-1), tab.seqfc, wtab]
        offs, total = [], 0
        for p in parts:
            offs.append(total)
            total += _align(p.size, 4)                    # 16-byte aligned starts (int4 loads)
        host = np.zeros(total, dtype=np.int32)
        for o, p in zip(offs, parts):
            host[o:o + p.size] = p
        if torch.device(device).type == "cuda":
            if stage is None:
                from .att_host import PinnedStage
                stage = PinnedStage()
            self.tables = stage.upload(host, device)      # (numpy in: staged without torch's intra-op thread pool)
        else:                                             # host-logic tests
            self.tables = torch.from_numpy(host).to(device)
        self.stage = stage
        self.fmap = self.tables[offs[0]: offs[0] + tab.fmap.size]
        self.gdesc = self.tables[offs[1]: offs[1] + tab.gdesc.size]
        self.seq = self.tables[offs[2]: offs[2] + tab.seq.size]
        self.seqfc = self.tables[offs[3]: offs[3] + tab.seqfc.size]
        self.wtab = self.tables[offs[4]: offs[4] + wtab.size] if wtab.size else None
        self.rowinfo = torch.empty(max(self.U, 1) * 4, dtype=torch.int32, device=device)
        if self.tables.is_cuda and self.S > 0:
            from .ops import _stream
            with torch.cuda.device(device):
                _lib.check(_lib.load().stage_rag_rowinfo(self.seq.data_ptr(), self.seqfc.data_ptr(), self.S, self.Lqa,
                                                         self.rowinfo.data_ptr(), _stream()), "stage_rag_rowinfo")
        self.T = (ctypes.c_void_p * 4)(self.fmap.data_ptr(), self.gdesc.data_ptr(), self.seq.data_ptr(), self.rowinfo.data_ptr())

    def device_tensors(self) -> list:
        """The device allocations behind the table pointers (for ``record_stream`` when another stream reads them)."""
        return [self.tables, self.rowinfo]

    @property
    def out_rows(self) -> int:
        return self.N * self.NA * self.Li      # the pooled encoder group writes one row per (example, candidate, frame)

    def attention_tables(self, ctx: Optional["CtxLayout"]):
        """Table pointers of the attention group: fmap, gdesc, seq, rowinfo, the context stream's frame table (or NULL), the balanced
        work table of the fused [a, b, a*b] backward (or NULL)."""
        return (ctypes.c_void_p * 6)(self.fmap.data_ptr(), self.gdesc.data_ptr(), self.seq.data_ptr(), self.rowinfo.data_ptr(),
                                     None if ctx is None else ctx.cq.data_ptr(), None if self.wtab is None else self.wtab.data_ptr())

    @property
    def live_fraction(self) -> float:
        return self.U / float(max(1, self.N * self.NA * self.Li * self.Lqa))


class CtxTables:
    """Host side of a ragged CONTEXT stream (subtitle words / video regions in front of the attention, model/stage.py:235-270).
    ``lens`` (N, Li): last valid word / region + 1 of every frame (0: none).  A frame keeps ``min(L, len + halo)`` rows --
    ``halo = n_blocks * n_conv * (k // 2)`` of the INPUT encoder, whose unmasked convolutions let that many padded positions leak into
    the valid ones; what lies behind never reaches a valid position, and the attention masks every padded position itself
    (model/context_query_attention.py:58-61, 100), so nothing else reads it.  A frame without a valid position keeps no row."""

    def __init__(self, lens: np.ndarray, L: int, halo: int):
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        self.N, self.Li = lens.shape
        self.L, self.halo = int(L), int(halo)
        lv = lens.reshape(-1)
        qlen = np.where(lv > 0, np.minimum(L, lv + int(halo)), 0)
        qstart = np.concatenate([[0], np.cumsum(qlen)[:-1]])
        self.U = int(qlen.sum())
        live = qlen > 0
        self.S = int(live.sum())
        self.live_frames = np.nonzero(live)[0].astype(np.int32)                                   # frames that keep rows, in order
        self.regular = bool(self.S == 0 or (qlen[live] == self.L).all())                        # every live frame keeps all L rows
        self.cq = np.stack([np.where(live, qstart, 0), qlen], axis=1).astype(np.int32)        # (frames, 2)
        self.seq = np.stack([qstart[live], qlen[live], np.zeros(self.S, np.int64), np.zeros(self.S, np.int64)], axis=1).astype(np.int32)

    def src_rows_host(self) -> np.ndarray:
        """What stage_rag_ctx_rows computes on the device (tests)."""
        out = np.empty(self.U, dtype=np.int32)
        for f in range(self.cq.shape[0]):
            s, n = int(self.cq[f, 0]), int(self.cq[f, 1])
            out[s:s + n] = f * self.L + np.arange(n)
        return out


class CtxLayout:
    """Device side of ``CtxTables``: the tables, the source-row table, and the table-pointer array the ragged encoder group takes."""
    out_rows = 0          # (the encoder group does not pool over these sequences)

    def __init__(self, tab: CtxTables, device, stage=None):
        self.tab = tab
        self.N, self.Li, self.Lqa = tab.N, tab.Li, tab.L       # Lqa: the longest sequence, under the name the encoder group uses
        self.U, self.S = tab.U, tab.S
        self.Ucap = max(CAP_STEP, _align(tab.U, CAP_STEP))
        parts = [tab.cq.reshape(-1), tab.seq.reshape(-1), tab.live_frames]
        offs, total = [], 0
        for p in parts:
            offs.append(total)
            total += _align(p.size, 4)
        host = np.zeros(max(total, 4), dtype=np.int32)
        for o, p in zip(offs, parts):
            host[o:o + p.size] = p
        if torch.device(device).type == "cuda":
            if stage is None:
                from .att_host import PinnedStage
                stage = PinnedStage()
            self.tables = stage.upload(host, device)
        else:
            self.tables = torch.from_numpy(host).to(device)
        self.stage = stage
        self.cq = self.tables[offs[0]: offs[0] + tab.cq.size]
        self.seq = self.tables[offs[1]: offs[1] + tab.seq.size]
        self.live_frames = self.tables[offs[2]: offs[2] + tab.live_frames.size]     # (S,) frame index of every sequence
        self.regular = tab.regular
        self.src_rows = torch.empty(max(self.U, 1), dtype=torch.int32, device=device)
        if self.tables.is_cuda and self.U > 0:
            from .ops import _stream
            with torch.cuda.device(device):
                _lib.check(_lib.load().stage_rag_ctx_rows(self.cq.data_ptr(), tab.N * tab.Li, tab.L, self.src_rows.data_ptr(), _stream()),
                           "stage_rag_ctx_rows")
        self.T = (ctypes.c_void_p * 4)(None, None, self.seq.data_ptr(), None)

    def device_tensors(self) -> list:
        return [self.tables, self.src_rows]


def mask_lens(mask: np.ndarray) -> np.ndarray:
    """(..., L) 0/1 mask -> last non-zero position + 1 along the last axis (0: none); holes inside stay live."""
    v = np.asarray(mask) != 0
    L = v.shape[-1]
    return np.where(v.any(-1), L - np.argmax(v[..., ::-1], axis=-1), 0).astype(np.int32)


def host_info(batch) -> Optional[dict]:
    """What the loader knows about the masks, as attached to the batch (``batch.mask_host``: ``qas`` (N, NA, Lqa) bool and the per-frame
    lengths ``sub_len`` / ``vid_len`` (N, Li) = last valid word / region + 1: tvqaplus_amd.synth.make_batch; the collate function builds
    the masks from exactly these numbers, tvqa_dataset.py:515-590), or None."""
    mh = batch.get("mask_host") if isinstance(batch, dict) else getattr(batch, "mask_host", None)
    return mh if mh and mh.get("qas") is not None else None


def host_masks(batch, frame_stream: str) -> Optional[tuple]:
    """(qa_valid, frame_live) from ``batch.mask_host`` (see host_info), or None."""
    mh = host_info(batch)
    if mh is None or mh.get(frame_stream + "_len") is None:
        return None
    return np.asarray(mh["qas"], dtype=bool), np.asarray(mh[frame_stream + "_len"]) > 0


def info_from_device(qas_mask: torch.Tensor, ctx_masks: dict) -> dict:
    """The same from the device tensors: ONE small device -> host copy (it waits for the queue: batches that come with ``mask_host``
    avoid it).  qas_mask (N, NA, Lqa); ctx_masks: {"sub" / "vid": (N, Li, L)}."""
    N, NA, Lqa = qas_mask.shape
    parts = [(qas_mask != 0).reshape(-1).to(torch.int32)]
    keys = sorted(ctx_masks)
    for k in keys:
        m = ctx_masks[k]
        pos = torch.arange(1, m.shape[-1] + 1, device=m.device, dtype=torch.int32)
        parts.append(((m != 0).to(torch.int32) * pos).amax(-1).reshape(-1))
    flat = torch.cat(parts).cpu().numpy()
    out = {"qas": flat[:N * NA * Lqa].reshape(N, NA, Lqa).astype(bool)}
    o = N * NA * Lqa
    for k in keys:
        n = ctx_masks[k].shape[0] * ctx_masks[k].shape[1]
        out[k + "_len"] = flat[o:o + n].reshape(ctx_masks[k].shape[0], ctx_masks[k].shape[1]).astype(np.int32)
        o += n
    return out


def masks_from_device(qas_mask: torch.Tensor, ctx_mask: torch.Tensor) -> tuple:
    """(qa_valid, frame_live) from the device masks (one read-back)."""
    info = info_from_device(qas_mask, {"ctx": ctx_mask})
    return info["qas"], info["ctx_len"] > 0


def bucket_plan(lens: np.ndarray, L: int, halo: int, step: int) -> list:
    """Length buckets of a context stream for the paths that keep dense (frames, L, .) tensors (long rows, bf16 storage: everything
    the ``*_rag`` group entry points decline).  ``lens`` (frames,) = last valid position + 1 (0: none).  A frame needs the positions
    below ``len + halo`` (the input encoder's convolutions reach no further: module docstring); it goes into the bucket of the
    smallest multiple of ``step`` that holds them (capped at L), frames without a valid position into none.  Returns
    ``[(frame indices int64, Lb), ...]`` by ascending Lb -- each bucket is an ordinary dense ``(len(idx), Lb, .)`` batch whose rows
    behind ``Lb`` are never computed."""
    lens = np.asarray(lens).reshape(-1).astype(np.int64)
    need = np.minimum(L, lens + int(halo))
    Lb = np.minimum(L, -(-need // int(step)) * int(step))
    Lb = np.where(lens > 0, Lb, 0)
    return [(np.nonzero(Lb == v)[0].astype(np.int64), int(v)) for v in np.unique(Lb) if v > 0]
