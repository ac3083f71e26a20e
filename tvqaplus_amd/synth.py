"""Synthetic TVQA+-shaped batches (SURVEY.md section 8d / BASELINE.md section 3).

Produces the ``batch`` object that ``STAGE.forward`` consumes, with the schema that the reference's
``pad_collate`` + ``prepare_inputs`` emit (tvqa_dataset.py:592-688): zero-padded fp32 features, fp32 0/1
masks, int64 labels.  Used by bench.py, the tests and the golden-vector generator.
"""
from __future__ import annotations

from argparse import Namespace
from typing import Optional

import torch


class Batch(dict):
    """Attribute + item access, like the ``EasyDict`` the reference drivers pass (inference.py:67)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to(self, device, non_blocking: bool = False) -> "Batch":
        out = Batch()
        for k, v in self.items():
            if torch.is_tensor(v):
                out[k] = v.to(device, non_blocking=non_blocking)
            elif isinstance(v, dict):
                out[k] = {kk: (vv.to(device, non_blocking=non_blocking) if torch.is_tensor(vv) else vv)
                          for kk, vv in v.items()}
            else:
                out[k] = v
        return out


def make_opt(**kw) -> Namespace:
    """The attributes ``STAGE.__init__`` reads (model/stage.py:58-76,82,122-145) with config.py defaults."""
    d = dict(sub_flag=True, vfeat_flag=True, vfeat_size=300, t_iter=0, extra_span_length=3, add_local=False,
             use_sup_att=False, num_negatives=2, negative_pool_size=0, num_hard=2, drop_topk=0, margin=0.1,
             att_loss_type="lse", scale=10.0, alpha=20.0, dropout=0.1, hsz=128, embedding_size=768,
             input_encoder_n_blocks=1, input_encoder_n_conv=2, input_encoder_kernel_size=7,
             input_encoder_n_heads=0, cls_encoder_n_blocks=1, cls_encoder_n_conv=2, cls_encoder_kernel_size=5,
             cls_encoder_n_heads=0, add_non_visual=False)
    d.update(kw)
    return Namespace(**d)


def _lengths(gen: torch.Generator, shape, lo: int, hi: int) -> torch.Tensor:
    lo = max(1, min(lo, hi))
    return torch.randint(lo, hi + 1, shape, generator=gen)


def _last_valid(mask: torch.Tensor):
    """(..., L) 0/1 mask -> numpy int32 (...): last valid position + 1 (0: none) -- the lengths the collate function pads to."""
    v = (mask != 0).numpy()
    L = v.shape[-1]
    import numpy as np
    return np.where(v.any(-1), L - np.argmax(v[..., ::-1], axis=-1), 0).astype(np.int32)


def _len_mask(lengths: torch.Tensor, L: int) -> torch.Tensor:
    return (torch.arange(L).view(*([1] * lengths.dim()), L) < lengths.unsqueeze(-1)).float()


def make_batch(N: int = 16, Li: int = 300, Lr: int = 20, Lw: int = 50, Lqa: int = 40, wd_size: int = 768,
               vfeat_size: int = 300, seed: int = 2018, ragged: bool = True, device: Optional[str] = None,
               empty_frames: bool = False, att_imgs: int = 0, att_words: int = 0) -> Batch:
    """ragged=True: Lqa_{n,a}~U[0.3Lqa,Lqa], Lw_{n,i}~U[0.1Lw,Lw], Lr_{n,i}~U[0.4Lr,Lr], frames
    Li_n~U[2Li/3,Li] with trailing frames fully masked (ts_label_mask = frame mask); item 0 keeps full
    lengths so padded shapes equal the requested ones.  ragged=False: all-ones masks (dense upper bound).
    empty_frames=True additionally blanks the regions of one *valid* frame per item (edge case).
    att_imgs > 0: region-level attention labels for the first att_imgs frames of every item (tvqa_dataset.py's
    att_labels: per item a list of (Lqa, Lr) 0/1 tensors, ~10 % positives, region 0 always negative).
    att_words > 0 (the bench): TVQA+-like sparsity instead -- per annotated frame `att_words` object words of the
    ground-truth answer, each with one or two positive regions, all inside the valid words / regions of the item."""
    gen = torch.Generator().manual_seed(seed)
    f32 = dict(generator=gen, dtype=torch.float32)
    qas_bert = torch.randn(N, 5, Lqa, wd_size, **f32)
    sub_bert = torch.randn(N, Li, Lw, wd_size, **f32)
    vid = torch.randn(N, Li, Lr, vfeat_size, **f32)
    if ragged:
        n_frames = _lengths(gen, (N,), (2 * Li + 2) // 3, Li)
        n_frames[0] = Li
        qa_len = _lengths(gen, (N, 5), max(1, (3 * Lqa) // 10), Lqa)
        qa_len[0, 0] = Lqa
        w_len = _lengths(gen, (N, Li), max(1, Lw // 10), Lw)
        w_len[0, 0] = Lw
        r_len = _lengths(gen, (N, Li), max(1, (2 * Lr) // 5), Lr)
        r_len[0, 0] = Lr
    else:
        n_frames = torch.full((N,), Li)
        qa_len = torch.full((N, 5), Lqa)
        w_len = torch.full((N, Li), Lw)
        r_len = torch.full((N, Li), Lr)
    frame_mask = _len_mask(n_frames, Li)                      # (N, Li)
    qas_mask = _len_mask(qa_len, Lqa)                         # (N, 5, Lqa)
    sub_mask = _len_mask(w_len, Lw) * frame_mask.unsqueeze(-1)
    vid_mask = _len_mask(r_len, Lr) * frame_mask.unsqueeze(-1)
    if empty_frames and Li > 1:
        for n in range(N):
            vid_mask[n, int(n_frames[n]) // 2] = 0.0
    qas_bert = qas_bert * qas_mask.unsqueeze(-1)
    sub_bert = sub_bert * sub_mask.unsqueeze(-1)
    vid = vid * vid_mask.unsqueeze(-1)
    target = torch.randint(0, 5, (N,), generator=gen)
    st = (torch.rand(N, generator=gen) * n_frames.float()).long().clamp(max=Li - 1)
    ed = st + (torch.rand(N, generator=gen) * (n_frames - st).float()).long()
    ed = torch.minimum(ed, n_frames - 1)
    b = Batch(qas_bert=qas_bert, qas_mask=qas_mask, sub_bert=sub_bert, sub_mask=sub_mask, vid=vid, vid_mask=vid_mask,
              target=target, ts_label=dict(st=st, ed=ed), ts_label_mask=frame_mask,
              target_list=target.tolist(),     # host copy kept by the input pipeline (att_host.build_att_pairs)
              # host copies of what the masks say about whole words / frames (the collate function builds the masks from
              # lengths on the host, tvqa_dataset.py:515-590): lets STAGE lay out its ragged token rows without reading the
              # device masks back (tvqaplus_amd/ragged.py: host_masks)
              mask_host=dict(qas=(qas_mask != 0).numpy(), sub_len=_last_valid(sub_mask), vid_len=_last_valid(vid_mask)),
              qid=list(range(N)), vid_name=["synthetic_%d" % i for i in range(N)],
              qas=torch.zeros(N, 5, Lqa, dtype=torch.long), att_labels=None, anno_st_idx=[0] * N, q_l=[1] * N,
              image_indices=[list(range(Li)) for _ in range(N)], boxes=[[] for _ in range(N)],
              use_hard_negatives=False, eval_object_word_ids=[])
    if att_imgs > 0:
        b.att_labels = []
        for n in range(N):
            per = []
            for i in range(att_imgs):
                if att_words > 0:
                    lab = torch.zeros(Lqa, Lr)
                    nw, nr = int(qa_len[n, int(target[n])]), int(r_len[n, i])
                    words = torch.randperm(nw, generator=gen)[:min(att_words, nw)]
                    for w in words.tolist():
                        k = 1 + int(torch.randint(0, 2, (1,), generator=gen))
                        regs = 1 + torch.randperm(max(nr - 1, 1), generator=gen)[:k]      # region 0 stays negative
                        lab[w, regs.clamp(max=nr - 1)] = 1.0
                else:
                    lab = (torch.rand(Lqa, Lr, generator=gen) < 0.1).float()
                    lab[:, 0] = 0
                per.append(lab)
            b.att_labels.append(per)
    return b.to(device) if device is not None else b
