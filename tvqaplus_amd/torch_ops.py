"""``torch.ops.stage_hip.*``: the per-kernel wrappers of ``tvqaplus_amd.ops`` registered as torch custom operators through
``torch.library`` (CompositeImplicitAutograd: the wrappers' ``autograd.Function``s carry the hand-written backward kernels).
The kernels stay behind the C ABI of ``libstage_hip.so``; the model itself calls the K-group entry points (``groups.py``).
"""
from __future__ import annotations

import torch

from . import ops

_LIB = None

# name -> (schema, implementation)
_OPS = {
    "layernorm": ("(Tensor x, Tensor gamma, Tensor beta, float p=0.0, int seed=0, Tensor? res=None, int res_period=0) -> (Tensor, Tensor?)",
                  lambda x, gamma, beta, p=0.0, seed=0, res=None, res_period=0: ops.layernorm(x, gamma, beta, p, seed, res, res_period)),
    "cat3_layernorm": ("(Tensor a, Tensor b, Tensor gamma, Tensor beta, int rep=1, int inner=1, float p=0.0, int seed=0) -> Tensor",
                       lambda a, b, gamma, beta, rep=1, inner=1, p=0.0, seed=0: ops.cat3_layernorm(a, b, gamma, beta, rep, inner, p, seed)),
    "linear": ("(Tensor x, Tensor w, Tensor? bias=None, bool relu=False) -> Tensor",
               lambda x, w, bias=None, relu=False: ops.linear(x, w, bias, relu)),
    "dwconv": ("(Tensor x, Tensor w, Tensor bias) -> Tensor", lambda x, w, bias: ops.dwconv(x, w, bias)),
    "ln_dwconv": ("(Tensor x, Tensor gamma, Tensor beta, Tensor w, Tensor bias, float p=0.0, int seed=0, Tensor? res=None, int res_period=0) -> (Tensor, Tensor?)",
                  lambda x, gamma, beta, w, bias, p=0.0, seed=0, res=None, res_period=0:
                  ops.ln_dwconv(x, gamma, beta, w, bias, p, seed, res, res_period)),
    "l2norm": ("(Tensor x, float p=0.0, int seed=0) -> Tensor", lambda x, p=0.0, seed=0: ops.l2norm(x, p, seed)),
    "structured_attention": ("(Tensor C, Tensor Q, Tensor c_mask, Tensor q_mask, float scale, float p=0.0, int seed_c=0, int seed_q=0) -> (Tensor, Tensor, Tensor)",
                             lambda C, Q, c_mask, q_mask, scale, p=0.0, seed_c=0, seed_q=0:
                             ops.structured_attention(C, Q, c_mask, q_mask, scale, p, seed_c, seed_q)),
    "masked_max": ("(Tensor x, Tensor mask, Tensor? window=None) -> Tensor", lambda x, mask, window=None: ops.masked_max(x, mask, window)),
    "ln_masked_max": ("(Tensor x, Tensor? res, Tensor gamma, Tensor beta, Tensor mask) -> Tensor",
                      lambda x, res, gamma, beta, mask: ops.ln_masked_max(x, res, gamma, beta, mask)),
    "mha_core": ("(Tensor q, Tensor k, Tensor v, Tensor mask, int nh, float p=0.0, int seed=0) -> Tensor",
                 lambda q, k, v, mask, nh, p=0.0, seed=0: ops.mha_core(q, k, v, mask, nh, p, seed)),
}


def register() -> None:
    global _LIB
    if _LIB is not None:
        return
    lib = torch.library.Library("stage_hip", "DEF")
    for name, (schema, fn) in _OPS.items():
        lib.define(name + schema)
        lib.impl(name, fn, "CompositeImplicitAutograd")
    _LIB = lib


def names():
    return sorted(_OPS)
