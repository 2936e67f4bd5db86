"""ctypes binding of libclibd_hip.so (C ABI declared in include/clibd_hip.h).

The product path has NO CPU fallback: if the shared object is missing or a symbol cannot be bound the
import of any compute entry point raises.  (`oracle/` is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CLIBD_HIP_LIB", _HERE / "libclibd_hip.so"))

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class GemmEpilogue(C.Structure):
    """Mirror of `struct clibd_gemm_epilogue` (include/clibd_hip.h)."""

    _fields_ = [
        ("bias", c_void_p),
        ("rank_u", c_void_p),
        ("rank_v", c_void_p),
        ("aux_bf16", c_void_p),
        ("residual_f32", c_void_p),
        ("out_pre_bf16", c_void_p),
        ("out_bf16", c_void_p),
        ("out_f32", c_void_p),
        ("act", C.c_int32),
        ("ld_rank_u", C.c_int32),
        ("ld_aux", C.c_int32),
        ("ld_res", C.c_int32),
        ("ld_pre", C.c_int32),
        ("ld_out_bf16", C.c_int32),
        ("ld_out_f32", C.c_int32),
        ("split_k", C.c_int32),
        ("drop_seed", C.c_uint32),
        ("drop_thr16", C.c_int32),
        ("drop_scale", C.c_float),
        ("drop_ld", C.c_int32),
        ("row_sums", c_void_p),
        ("row_stats", c_void_p),
        ("col_sum_w", c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/clibd_hip.h declares must appear here
SIGNATURES = {
    "clibd_last_error": (C.c_char_p, []),
    "clibd_abi_version": (c_int, []),
    "clibd_build_hash": (C.c_char_p, []),
    "clibd_gemm_bf16_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, C.POINTER(GemmEpilogue), c_void_p]),
    "clibd_gemm_tail_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "clibd_gemm_bf16_nt_ws": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, C.POINTER(GemmEpilogue), c_void_p, c_size_t, c_void_p]),
    "clibd_gemm_bf16_nt_khole": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, C.POINTER(GemmEpilogue), c_void_p]),
    "clibd_gemm_fp8_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, C.POINTER(GemmEpilogue), c_void_p]),
    "clibd_quantize_rows_fp8": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "clibd_gemm_fp8_dgrad_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, C.POINTER(GemmEpilogue), c_void_p]),
    "clibd_quantize_rows_fp8_bf16": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clibd_transpose_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "clibd_transpose_colsum_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "clibd_transpose_colsum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "clibd_transpose_colsum_bf16_ws": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_cast_transpose_f32_to_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "clibd_layernorm_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clibd_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clibd_layernorm_fwd_drop": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_layernorm_bwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_layernorm_fwd_fp8": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_uint32, c_int, c_float, c_void_p, c_float, c_void_p]),
    "clibd_attention_fwd_fp8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, C.c_uint32, c_int, c_float, c_float, c_void_p]),
    "clibd_layernorm_bwd_res16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_layernorm_bwd_any": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        C.c_uint32, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "clibd_layernorm_bwd_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        C.c_uint32, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "clibd_layernorm_bwd_fp8_pg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           C.c_uint32, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clibd_layernorm_bwd_pg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, C.c_uint32, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "clibd_attention_fwd_drop": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_attention_bwd_drop": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_attention_fwd_save": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, C.c_uint32, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "clibd_attention_bwd_sp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_attention_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "clibd_attention_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "clibd_lora_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clibd_lora_down_proj": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "clibd_lora_workspace_bytes": (c_size_t, [c_int, c_int]),
    "clibd_lora_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_lora_backward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_patchify": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "clibd_patchify_u8": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "clibd_rowsum_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "clibd_ln_fold_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clibd_vit_assemble_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_gelu_bwd_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "clibd_bert_embed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clibd_softmax_mean_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_softmax_mean_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_token_mean_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_token_mean_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_colsum_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_gather_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_scatter_rows_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "clibd_l2norm_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "clibd_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "clibd_softce_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "clibd_softce_rows_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_softce_rows_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_topk_ip_workspace_bytes": (c_size_t, [c_int, c_int]),
    "clibd_topk_prepare_keys": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "clibd_topk_ip_fast_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "clibd_topk_ip_fast": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_topk_ip": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_kmer_tokenize": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_layernorm_param_grads": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_gemm_splitk_workspace_bytes": (c_size_t, [c_int, c_int]),
    "clibd_gemm_bf16_tn_splitk": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "clibd_gemm_bf16_nt_splitk": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "clibd_dropout_apply_f32": (c_int, [c_void_p, c_size_t, c_void_p, C.c_uint32, c_int, c_float, c_void_p]),
    "clibd_batch_sum_f32": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p]),
    "clibd_bert_embed_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "clibd_slice_rows_cast_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clibd_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p]),
}

_lib = None
ABI_VERSION = 5   # what this binding was written against (clibd_abi_version(), csrc/capi.hip); load() refuses any other library


class ClibdHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library and bind every declared symbol. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's): it must be the one HIP runtime of the
    # process, or device pointers / streams handed over from torch would belong to a different runtime instance.
    import torch  # noqa: F401  (loads libtorch_hip -> the bundled HIP runtime first)

    if not LIB_PATH.exists():
        raise ClibdHipError(
            f"{LIB_PATH} not found: build it with `python -m clibd_amd.build` (hipcc, gfx950). "
            "clibd_amd has no CPU fallback for its compute path."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ClibdHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.clibd_abi_version() != ABI_VERSION:
        raise ClibdHipError(f"{LIB_PATH} speaks C-ABI version {lib.clibd_abi_version()}, this binding {ABI_VERSION}: rebuild with `python -m clibd_amd.build`")
    _check_build_hash(lib)
    _lib = lib
    return lib


def _check_build_hash(lib) -> None:
    """The library must have been built from the kernel sources that sit beside it (mtime-based rebuilds can be fooled by a
    checkout or a copied tree).  CLIBD_HIP_LIB (an explicitly chosen library) and a source-less installation skip the check."""
    if "CLIBD_HIP_LIB" in os.environ or not (_HERE / "csrc").is_dir():
        return
    from .build import csrc_hash

    built, cur = (lib.clibd_build_hash() or b"").decode(), csrc_hash()
    if built != cur:
        raise ClibdHipError(f"{LIB_PATH} was built from other kernel sources (library {built}, clibd_amd/csrc {cur}): "
                            "rebuild it with `python -m clibd_amd.build`")


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().clibd_last_error()
        raise ClibdHipError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")
