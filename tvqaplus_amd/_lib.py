"""ctypes binding of the C-ABI HIP library (include/stage_hip.h -> tvqaplus_amd/libstage_hip.so).

There is NO fallback: if the library is missing or a symbol is absent this raises, and every op in
``tvqaplus_amd.ops`` refuses CPU tensors.  Build it with ``make`` or ``python -c 'import __graft_entry__ as g; g.build()'``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# STAGE_HIP_LIB: experiment builds (tools/build_variant.sh).
LIB_PATH = os.environ.get("STAGE_HIP_LIB") or os.path.join(_HERE, "libstage_hip.so")

P = c_void_p  # every device pointer / stream travels as void*
I, LL, F, U64, SZ = c_int, c_longlong, c_float, c_ulonglong, c_size_t

# name -> (restype, argtypes); mirrors include/stage_hip.h one to one (tests/test_abi.py checks the two agree)
SIGNATURES = {
    "stage_hip_abi_version": (I, []),
    "stage_hip_error_string": (c_char_p, [I]),
    "stage_timer_create": (P, []),
    "stage_timer_destroy": (None, [P]),
    "stage_timer_elapsed_ms": (F, [P, P]),
    "stage_k1_fwd_timer": (None, [P, P, I]),
    "stage_str_attn_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, U64, P]),
    "stage_str_attn_bwd_ws_bytes": (SZ, [I, I, I, I]),
    "stage_str_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, SZ, P]),
    "stage_str_attn_bwd_fused_ws_bytes": (SZ, [I, I, I, I, I]),
    "stage_str_attn_bwd_fused": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, SZ, P]),
    "stage_str_attn_long_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, I, P]),
    "stage_str_attn_long_bwd_ws_bytes": (SZ, [I, I, I, I]),
    "stage_str_attn_long_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, I, P, SZ, P]),
    "stage_str_attn_long_bwd_qm_ws_bytes": (SZ, [I, I, I, I, I]),
    "stage_str_attn_long_bwd_qm": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, I, P, SZ, P]),
    "stage_l2norm_fwd": (I, [P, P, P, LL, I, F, F, U64, P]),
    "stage_l2norm_bwd": (I, [P, P, P, LL, I, F, F, U64, I, P]),
    "stage_layernorm_fwd": (I, [P, P, I, P, P, P, P, P, P, LL, I, F, F, U64, P]),
    "stage_ln_bwd_ws_bytes": (SZ, [I]),
    "stage_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, LL, I, F, U64, P, SZ, P]),
    "stage_cat3_layernorm_fwd": (I, [P, P, P, P, P, P, P, LL, I, I, I, F, F, U64, P]),
    "stage_cat3_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_cat3_layernorm_bwd_reduced_ws_bytes": (SZ, [LL, I, I, I]),
    "stage_cat3_layernorm_bwd_reduced": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_reduce_rep": (I, [P, P, LL, I, LL, P]),
    "stage_ln_dwconv_fwd": (I, [P, P, I, P, P, P, P, P, P, P, P, LL, I, I, I, F, F, U64, P]),
    "stage_ln_dwconv_bwd_ws_bytes": (SZ, [I, I]),
    "stage_ln_dwconv_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_gemm_nt": (I, [P, P, P, P, P, P, LL, I, I, I, P]),
    "stage_gemm_tn_ws_bytes": (SZ, [LL, I, I]),
    "stage_gemm_tn": (I, [P, P, P, P, P, LL, I, I, P, SZ, P]),
    "stage_gemm_mask_supported": (I, [LL, I, I]),
    "stage_gemm_nt_mask": (I, [P, P, P, P, P, P, LL, I, I, I, P]),
    "stage_gemm_tn_mask": (I, [P, P, P, P, P, LL, I, I, P, SZ, P]),
    "stage_add_pe": (I, [P, P, P, LL, I, I, P]),
    "stage_dwconv_fwd": (I, [P, P, P, P, LL, I, I, I, P]),
    "stage_dwconv_bwd_ws_bytes": (SZ, [I, I]),
    "stage_dwconv_bwd": (I, [P, P, P, P, P, P, LL, I, I, I, P, SZ, P]),
    "stage_mha_core_recomputes": (I, [I, I, I]),
    "stage_mha_core_fwd": (I, [P, P, P, P, P, P, LL, I, I, I, F, U64, P]),
    "stage_mha_core_bwd": (I, [P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P]),
    "stage_masked_max_fwd": (I, [P, P, P, P, P, LL, I, I, P]),
    "stage_masked_max_bwd": (I, [P, P, P, P, LL, I, I, I, P]),
    "stage_ln_masked_max_supported": (I, [I, I]),
    "stage_gemm_nt_lnparam_supported": (I, [LL, I, I]),
    "stage_gemm_nt_lnparam_ws_bytes": (SZ, [LL, I]),
    "stage_gemm_nt_lnparam": (I, [P, P, P, P, P, P, P, F, P, P, LL, I, I, P, SZ, P]),
    "stage_dropout_keepmask": (I, [F, U64, P, LL, I, P]),
    "stage_ln_masked_max_fwd": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, F, P]),
    "stage_ln_masked_max_bwd": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, P, SZ, P]),
    # bf16 storage mode: same argument lists as the fp32 entry points of the same name
    "stage_str_attn_fwd_bf16": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, U64, P]),
    "stage_str_attn_bwd_fused_bf16": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, SZ, P]),
    "stage_layernorm_fwd_bf16": (I, [P, P, I, P, P, P, P, P, P, LL, I, F, F, U64, P]),
    "stage_layernorm_bwd_bf16": (I, [P, P, P, P, P, P, P, P, P, LL, I, F, U64, P, SZ, P]),
    "stage_cat3_layernorm_fwd_bf16": (I, [P, P, P, P, P, P, P, LL, I, I, I, F, F, U64, P]),
    "stage_cat3_layernorm_bwd_bf16": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_cat3_layernorm_bwd_reduced_bf16": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_l2norm_fwd_bf16": (I, [P, P, P, LL, I, F, F, U64, P]),
    "stage_l2norm_bwd_bf16": (I, [P, P, P, LL, I, F, F, U64, I, P]),
    "stage_l2norm_bwd_mixed_bf16": (I, [P, P, P, P, LL, I, F, F, U64, P]),
    "stage_gemm_nt_bf16": (I, [P, P, P, P, P, P, LL, I, I, I, P]),
    "stage_gemm_tn_bf16_ws_bytes": (SZ, [LL, I, I]),
    "stage_gemm_tn_bf16": (I, [P, P, P, P, P, LL, I, I, P, SZ, P]),
    "stage_dwconv_fwd_bf16": (I, [P, P, P, P, LL, I, I, I, P]),
    "stage_dwconv_bwd_bf16": (I, [P, P, P, P, P, P, LL, I, I, I, P, SZ, P]),
    "stage_ln_dwconv_fwd_bf16": (I, [P, P, I, P, P, P, P, P, P, P, P, LL, I, I, I, F, F, U64, P]),
    "stage_ln_dwconv_bwd_bf16": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_mha_core_fwd_bf16": (I, [P, P, P, P, P, LL, I, I, I, F, U64, P]),
    "stage_mha_core_bwd_bf16": (I, [P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P]),
    "stage_mha_core_qkv_fwd": (I, [P, P, P, LL, I, I, I, F, U64, I, P]),
    "stage_mha_core_qkv_bwd": (I, [P, P, P, P, LL, I, I, I, F, U64, I, P]),
    "stage_masked_max_fwd_bf16": (I, [P, P, P, P, P, LL, I, I, P]),
    "stage_masked_max_bwd_bf16": (I, [P, P, P, P, LL, I, I, I, P]),
    "stage_ln_masked_max_fwd_bf16": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, F, P]),
    "stage_ln_masked_max_bwd_bf16": (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, P, SZ, P]),
    "stage_cat3_dx_ln_bwd_supported": (I, [LL, I, I, I]),
    "stage_cat3_dx_ln_bwd_ws_bytes": (SZ, [LL, I, I, I]),
    "stage_cat3_dx_ln_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_cat3_bwd_dw_supported": (I, [LL, I, I, I]),
    "stage_cat3_bwd_dw_ws_bytes": (SZ, [LL, I, I, I]),
    "stage_cat3_bwd_dw": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_cat3_ln_gemm_fwd_supported": (I, [LL, I, I, I]),
    "stage_cat3_ln_gemm_fwd_ws_bytes": (SZ, []),
    "stage_cat3_ln_gemm_fwd": (I, [P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, F, U64, P, SZ, P]),
    # K-groups (csrc/groups.hip): one forward and one backward symbol per fused-op group; params / grads / seeds / flags are
    # HOST arrays (pointers to them travel as void*)
    "stage_grp_input_mlp_arena_bytes": (SZ, [LL, I, I, I, I]),
    "stage_grp_input_mlp_fwd": (I, [P, P, P, P, SZ, P, LL, I, I, I, I, F, P, P]),
    "stage_grp_input_mlp_bwd_tmp_bytes": (SZ, [LL, I, I, I]),
    "stage_grp_input_mlp_bwd": (I, [P, P, P, P, P, SZ, P, P, SZ, LL, I, I, I, I, F, P, P]),
    "stage_grp_encoder_arena_bytes": (SZ, [LL, I, I, I, I]),
    "stage_grp_encoder_fwd": (I, [P, P, P, P, P, P, SZ, P, LL, I, I, I, I, F, P, P]),
    "stage_grp_encoder_bwd_tmp_bytes": (SZ, [LL, I, I, I, I]),
    "stage_grp_encoder_bwd": (I, [P, P, P, P, P, P, P, SZ, P, P, SZ, LL, I, I, I, I, F, P, P]),
    "stage_grp_qa_ctx_arena_bytes": (SZ, [I, I, I, I, I]),
    "stage_grp_qa_ctx_fwd": (I, [P, P, P, P, P, P, P, P, P, SZ, P, I, I, I, I, I, I, F, F, P, P]),
    "stage_grp_qa_ctx_bwd_tmp_bytes": (SZ, [I, I, I, I, I, I]),
    "stage_grp_qa_ctx_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, SZ, P, P, SZ, I, I, I, I, I, I, F, F, P, P]),
    "stage_grp_concat_fc_arena_bytes": (SZ, [LL, I]),
    "stage_grp_concat_fc_fwd": (I, [P, P, P, P, P, SZ, P, LL, I, F, P, P]),
    "stage_grp_concat_fc_bwd_tmp_bytes": (SZ, [LL, I]),
    "stage_grp_concat_fc_bwd": (I, [P, P, P, P, P, P, P, P, SZ, P, P, SZ, LL, I, F, P, P]),
    "stage_grp_temporal_head_arena_bytes": (SZ, [LL, I]),
    "stage_grp_temporal_head_fwd": (I, [P, P, P, P, P, P, SZ, P, LL, I, F, P, P]),
    "stage_grp_temporal_head_bwd_tmp_bytes": (SZ, [LL, I]),
    "stage_grp_temporal_head_bwd": (I, [P, P, P, P, P, P, P, P, P, SZ, P, P, SZ, LL, I, F, P, P]),
    # head glue: temporal scores, span proposal, pooling + classifier, auxiliary losses
    "stage_tscores_fwd": (I, [P, P, P, P, I, I, I, P]),
    "stage_tscores_bwd": (I, [P, P, P, P, I, I, I, P]),
    "stage_gt_spans": (I, [P, P, P, P, P, I, I, I, P]),
    "stage_ts_loss": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "stage_train_loss": (I, [P, P, P, P, P, F, F, F, P, P, I, I, P]),
    "stage_att_loss_fwd": (I, [P, P, LL, I, F, F, P, P, P]),
    "stage_att_loss_bwd": (I, [P, P, P, LL, P, LL, P]),
    "stage_grp_pool_cls_arena_bytes": (SZ, [LL, I, I]),
    "stage_grp_pool_cls_fwd": (I, [P, P, P, P, P, P, P, SZ, I, I, I, I, LL, F, P, P]),
    "stage_grp_pool_cls_bwd_tmp_bytes": (SZ, [LL, I, I]),
    "stage_grp_pool_cls_bwd": (I, [P, P, P, P, P, P, P, P, SZ, P, SZ, I, I, I, I, LL, F, P, P]),
    # ragged token rows (csrc/ragged.hip and the *_rag / *_fc variants; include/stage_hip.h "Ragged token rows")
    "stage_rag_rowinfo": (I, [P, P, LL, I, P, P]),
    "stage_rag_fill_pooled": (I, [P, P, LL, I, P]),
    "stage_rag_zero_dump": (I, [P, P, I, I, I, I, I, P]),
    "stage_str_attn_fwd_fc": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, U64, P]),
    "stage_str_attn_bwd_fused_fc": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, SZ, P]),
    "stage_rag_ctx_rows": (I, [P, LL, I, P, P]),
    "stage_layernorm_gather_fwd": (I, [P, P, P, P, P, P, P, LL, I, F, F, U64, P]),
    "stage_layernorm_gather_bwd": (I, [P, P, P, P, P, P, P, P, LL, I, F, U64, P, SZ, P]),
    "stage_l2norm_gather_fwd": (I, [P, P, P, LL, I, F, P]),
    "stage_grp_input_mlp_rag_fwd": (I, [P, P, P, P, P, SZ, P, LL, I, I, I, I, F, P, P]),
    "stage_grp_input_mlp_rag_bwd": (I, [P, P, P, P, P, P, SZ, P, P, SZ, LL, I, I, I, I, F, P, P]),
    "stage_cat3_ln_gemm_fwd_rag_supported": (I, [LL, LL, LL, I]),
    "stage_cat3_ln_gemm_fwd_rag": (I, [P, P, P, P, P, P, P, P, P, P, P, P, LL, LL, LL, I, F, F, U64, P, SZ, P]),
    "stage_cat3_dx_ln_bwd_rag_supported": (I, [LL, LL, I, I, I, I]),
    "stage_cat3_dx_ln_bwd_rag_ws_bytes": (SZ, [I, I, I]),
    "stage_cat3_rag_work_groups": (I, []),
    "stage_cat3_bwd_dw_rag_supported": (I, [LL, LL, I, I, I, I]),
    "stage_cat3_bwd_dw_rag_work_groups": (I, []),
    "stage_cat3_bwd_dw_rag_ws_bytes": (SZ, [I, I]),
    "stage_cat3_bwd_dw_rag": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, LL, LL, I, I, I, I, F, U64, P, SZ, P]),
    "stage_cat3_dx_ln_bwd_rag": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, LL, LL, I, I, I, I, F, U64, P, SZ, P]),
    "stage_ln_dwconv_rag_fwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, F, U64, P]),
    "stage_ln_dwconv_rag_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, U64, P, SZ, P]),
    "stage_ln_masked_max_rag_fwd": (I, [P, P, P, P, P, P, P, P, P, P, P, LL, I, I, F, P]),
    "stage_ln_masked_max_rag_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, LL, I, P, SZ, P]),
    "stage_grp_qa_ctx_rag_supported": (I, [I, I, I, I, I, I, LL, LL]),
    "stage_grp_qa_ctx_rag_arena_bytes": (SZ, [I, I, I, I, LL, LL]),
    "stage_grp_qa_ctx_rag_bwd_tmp_bytes": (SZ, [I, I, I, I, I, I, LL, LL]),
    "stage_grp_qa_ctx_rag_fwd": (I, [P, P, P, P, P, P, P, P, P, P, SZ, P, I, I, I, I, I, I, LL, LL, LL, LL, F, F, P, P]),
    "stage_grp_qa_ctx_rag_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, SZ, P, P, SZ, I, I, I, I, I, I, LL, LL, LL, LL, F, F, P, P]),
    "stage_grp_encoder_rag_arena_bytes": (SZ, [LL, LL, I, I]),
    "stage_grp_encoder_rag_bwd_tmp_bytes": (SZ, [LL, I, I]),
    "stage_grp_encoder_rag_fwd": (I, [P, P, P, P, P, P, P, SZ, P, LL, LL, LL, LL, I, I, I, I, F, P, P]),
    "stage_grp_encoder_rag_bwd": (I, [P, P, P, P, P, P, P, SZ, P, P, SZ, LL, LL, LL, LL, I, I, I, I, F, P, P]),
}

ABI_VERSION = 5    # include/stage_hip.h: STAGE_HIP_ABI_VERSION (tests/test_abi.py holds the two together)

_lib = None


class StageHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libstage_hip.so once; raise loudly if it (or any declared symbol) is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StageHipError(
            "tvqaplus_amd: %s not found -- the HIP extension is required (no CPU / eager fallback exists). "
            "Build it with `make` at the repo root." % LIB_PATH)
    # torch ships its own libamdhip64: it has to be in the process BEFORE this library so that both resolve to the same
    # HIP runtime (loaded the other way round, the library binds /opt/rocm's copy and sees no device once torch
    # initialises its own -- "no ROCm-capable device is detected")
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise StageHipError("tvqaplus_amd: symbol %s missing from %s" % (name, LIB_PATH)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.stage_hip_abi_version() != ABI_VERSION:
        raise StageHipError("tvqaplus_amd: %s has ABI version %d, this package binds version %d -- rebuild with `make`"
                            % (LIB_PATH, lib.stage_hip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


STAGE_ERR_SHAPE = -1      # include/stage_hip.h
STAGE_ERR_WORKSPACE = -2


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().stage_hip_error_string(code)
        msg = msg.decode() if msg else "?"
        if "memory" in msg.lower():
            msg += " (out of memory)"  # main.py:75-77 greps for this substring
        raise StageHipError("%s failed: %s (code %d)" % (what, msg, code))
