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
        parts = [tab.fmap, tab.gdesc.reshape(-1), tab.seq.reshape(-1), tab.seqfc]
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
        self.rowinfo = torch.empty(max(self.U, 1) * 4, dtype=torch.int32, device=device)
        if self.tables.is_cuda and self.S > 0:
            from .ops import _stream
            with torch.cuda.device(device):
                _lib.check(_lib.load().stage_rag_rowinfo(self.seq.data_ptr(), self.seqfc.data_ptr(), self.S, self.Lqa,
                                                         self.rowinfo.data_ptr(), _stream()), "stage_rag_rowinfo")
        self.T = (ctypes.c_void_p * 4)(self.fmap.data_ptr(), self.gdesc.data_ptr(), self.seq.data_ptr(), self.rowinfo.data_ptr())

    @property
    def live_fraction(self) -> float:
        return self.U / float(max(1, self.N * self.NA * self.Li * self.Lqa))


def host_masks(batch, frame_stream: str) -> Optional[tuple]:
    """(qa_valid, frame_live) as numpy bools from the host copies a loader attached to the batch (``batch.mask_host``: dict with
    ``qas`` (N, NA, Lqa) and ``sub_frames`` / ``vid_frames`` (N, Li): tvqaplus_amd.synth.make_batch, tvqaplus_amd.prefetch), or None."""
    mh = batch.get("mask_host") if isinstance(batch, dict) else getattr(batch, "mask_host", None)
    if not mh:
        return None
    fl = mh.get(frame_stream + "_frames")
    qa = mh.get("qas")
    if fl is None or qa is None:
        return None
    return np.asarray(qa, dtype=bool), np.asarray(fl, dtype=bool)


def masks_from_device(qas_mask: torch.Tensor, ctx_mask: torch.Tensor) -> tuple:
    """The same from the device tensors: ONE small device -> host copy (it waits for the queue: batches that come with
    ``mask_host`` avoid it).  qas_mask (N, NA, Lqa), ctx_mask (N, Li, Lr)."""
    N, NA, Lqa = qas_mask.shape
    Li = ctx_mask.shape[1]
    flat = torch.cat([(qas_mask != 0).reshape(-1), (ctx_mask.sum(-1) != 0).reshape(-1)]).to(torch.uint8).cpu().numpy().astype(bool)
    return flat[:N * NA * Lqa].reshape(N, NA, Lqa), flat[N * NA * Lqa:].reshape(N, Li)
