"""Tensor-level wrappers over the C ABI (include/clibd_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function below only checks the
operands on the host, takes raw pointers and enqueues hand-written gfx950 kernels on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import GemmEpilogue, check

ACT_NONE, ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_GRAD, ACT_MUL_AUX, ACT_ADD_AUX, ACT_GELU_SAVE_GRAD_U8, ACT_MUL_AUX_U8 = 0, 1, 2, 3, 4, 5, 6, 7
ACT_GELU_SAVE_GRAD_E12, ACT_MUL_AUX_E12 = 8, 9   # gelu' as the 12-bit e4m7 form of its bf16 value: uint8 buffers [M, 3N/2] (include/clibd_hip.h)
BF16, F32, I64, I32 = torch.bfloat16, torch.float32, torch.int64, torch.int32
FP8 = torch.float8_e4m3fn  # OCP e4m3 (gfx950's fp8 MFMA operand format); max finite 448


def _stream() -> int:
    """Raw handle of torch's current HIP stream.  Two direct C calls: `torch.cuda.current_stream().cuda_stream` builds a Stream object
    through four Python layers (8 us per call, 1.8 ms per step at ~1000 launches: tools/host_profile.py, round 4)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str, contiguous: bool = True) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor (clibd_amd has no CPU compute path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


def _rowmajor(t: torch.Tensor, name: str) -> int:
    """2-D row-major view with unit inner stride; returns the leading dimension in elements."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with unit inner stride")
    return t.stride(0)


class Drop:
    """Dropout site: probability p and a 32-bit seed; the mask is hash(seed, element index), see include/clibd_hip.h."""
    __slots__ = ("seed", "thr16", "scale")

    def __init__(self, p: float, seed: int):
        self.thr16 = int(round(p * 65536.0))
        if not 0 <= self.thr16 < 65536:
            raise ValueError("dropout p must be in [0, 1)")
        self.seed = seed & 0xFFFFFFFF
        self.scale = 1.0 / (1.0 - self.thr16 / 65536.0)


def derive_seed(base: int, layer: int, site: int) -> int:
    """site seeds of one tower call (site: 0 attention probs, 1 attention-output dropout, 2 output dropout, 3 embeddings)"""
    x = (base ^ ((layer * 8 + site + 1) * 0x9E3779B1)) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


# Measured in-step (profiles/r06_exp_gemm_stream_k_tail.log): 1.8-2.4 % SLOWER at per-GPU batch 256, 0.4-0.7 % slower at 2048 -> off by default (CLIBD_GEMM_STREAMK=1 turns it on)
_STREAMK = os.environ.get("CLIBD_GEMM_STREAMK", "0") == "1"
_TAIL_WS_BYTES = 48 * 1024 * 1024 + 1024
_tail_ws: dict = {}


def gemm_nt(
    a: torch.Tensor,
    w: torch.Tensor,
    *,
    bias: Optional[torch.Tensor] = None,
    rank_u: Optional[torch.Tensor] = None,
    rank_v: Optional[torch.Tensor] = None,
    act: int = ACT_NONE,
    aux: Optional[torch.Tensor] = None,
    residual: Optional[torch.Tensor] = None,
    out_pre: Optional[torch.Tensor] = None,
    out_bf16: Optional[torch.Tensor] = None,
    out_f32: Optional[torch.Tensor] = None,
    split_k: int = 1,
    drop: Optional["Drop"] = None,
    k_hole: Optional[tuple] = None,
    row_sums: Optional[torch.Tensor] = None,
    row_stats: Optional[torch.Tensor] = None,
    col_sum_w: Optional[torch.Tensor] = None,
) -> None:
    """out = epilogue(a[M,K] @ w[N,K]^T); see clibd_gemm_bf16_nt in include/clibd_hip.h.
    k_hole = (k0, length): the K range [k0, k0+length) of both operands is skipped (multiples of 64).
    row_sums (fp32 [N/128, M, 2]) / row_stats (fp32 [M,2]) + col_sum_w (fp32 [N]): the producer / consumer epilogues of the
    LayerNorm -> Linear fold (clibd_gemm_epilogue.row_sums, .row_stats)."""
    _chk(a, BF16, "a", contiguous=False)
    _chk(w, BF16, "w", contiguous=False)
    lda, ldw = _rowmajor(a, "a"), _rowmajor(w, "w")
    M, K = a.shape
    N, K2 = w.shape
    if K != K2:
        raise ValueError(f"gemm_nt: K mismatch {K} vs {K2}")
    ep = GemmEpilogue()
    ep.act = act
    ep.split_k = split_k
    if drop is not None and drop.thr16 > 0:
        ep.drop_seed, ep.drop_thr16, ep.drop_scale, ep.drop_ld = drop.seed, drop.thr16, drop.scale, N
    if bias is not None:
        _chk(bias, F32, "bias")
        if bias.numel() != N:
            raise ValueError("gemm_nt: bias must have N elements")
        ep.bias = bias.data_ptr()
    if rank_u is not None or rank_v is not None:
        _chk(rank_u, BF16, "rank_u", contiguous=False)
        _chk(rank_v, BF16, "rank_v")
        if rank_u.shape[0] != M or rank_u.shape[1] < 8 or tuple(rank_v.shape) != (N, 8):
            raise ValueError("gemm_nt: rank_u must be [M,>=8], rank_v [N,8]")
        ep.rank_u, ep.rank_v, ep.ld_rank_u = rank_u.data_ptr(), rank_v.data_ptr(), _rowmajor(rank_u, "rank_u")
    if aux is not None:
        _chk(aux, torch.uint8 if act in (ACT_MUL_AUX_U8, ACT_MUL_AUX_E12) else BF16, "aux", contiguous=False)   # _U8: gelu' codes, one byte per element; _E12: 1.5 bytes
        if tuple(aux.shape) != (M, 3 * N // 2 if act == ACT_MUL_AUX_E12 else N):
            raise ValueError("gemm_nt: aux must be [M,N] ([M,3N/2] bytes for the e4m7 form)")
        ep.aux_bf16, ep.ld_aux = aux.data_ptr(), _rowmajor(aux, "aux")
    if residual is not None:
        _chk(residual, F32, "residual", contiguous=False)
        if tuple(residual.shape) != (M, N):
            raise ValueError("gemm_nt: residual must be [M,N]")
        ep.residual_f32, ep.ld_res = residual.data_ptr(), _rowmajor(residual, "residual")
    for name, t, dt in (("out_pre", out_pre, torch.uint8 if act in (ACT_GELU_SAVE_GRAD_U8, ACT_GELU_SAVE_GRAD_E12) else BF16), ("out_bf16", out_bf16, BF16), ("out_f32", out_f32, F32)):
        if t is not None:
            _chk(t, dt, name, contiguous=False)
            if tuple(t.shape) != (M, 3 * N // 2 if (name == "out_pre" and act == ACT_GELU_SAVE_GRAD_E12) else N):
                raise ValueError(f"gemm_nt: {name} must be [M,N] ([M,3N/2] bytes for the e4m7 form)")
    if out_pre is not None:
        ep.out_pre_bf16, ep.ld_pre = out_pre.data_ptr(), _rowmajor(out_pre, "out_pre")
    if out_bf16 is not None:
        ep.out_bf16, ep.ld_out_bf16 = out_bf16.data_ptr(), _rowmajor(out_bf16, "out_bf16")
    if out_f32 is not None:
        ep.out_f32, ep.ld_out_f32 = out_f32.data_ptr(), _rowmajor(out_f32, "out_f32")
    if row_sums is not None:
        _chk(row_sums, F32, "row_sums")
        if N % 128 or tuple(row_sums.shape) != (N // 128, M, 2):
            raise ValueError("gemm_nt: row_sums must be [N/128, M, 2]")
        ep.row_sums = row_sums.data_ptr()
    if row_stats is not None or col_sum_w is not None:
        _chk(row_stats, F32, "row_stats")
        _chk(col_sum_w, F32, "col_sum_w")
        if tuple(row_stats.shape) != (M, 2) or col_sum_w.numel() != N:
            raise ValueError("gemm_nt: row_stats must be [M,2], col_sum_w [N]")
        ep.row_stats, ep.col_sum_w = row_stats.data_ptr(), col_sum_w.data_ptr()
    if k_hole is not None:
        check(_lib.load().clibd_gemm_bf16_nt_khole(a.data_ptr(), lda, w.data_ptr(), ldw, M, N, K, int(k_hole[0]), int(k_hole[1]), C.byref(ep),
                                                   _stream()), "gemm_bf16_nt_khole")
        return
    lib = _lib.load()
    # Stream-K tail (round 6): launches whose last tile round is at most half full and whose contraction is long take a per-(device, stream) workspace
    # (zeroed once; 48 MiB + 1 KiB covers every shape) and cut that round's tiles into K-slices over the idle CUs.  Opt-in: see _STREAMK.
    if _STREAMK and split_k == 1 and M >= 1024:
        need = int(lib.clibd_gemm_tail_workspace_bytes(M, N, K))
        if need > 0:
            key = (a.device, torch.cuda.current_stream(a.device).cuda_stream)
            ws = _tail_ws.get(key)
            if ws is None or ws.numel() < need:
                ws = torch.zeros((max(need, _TAIL_WS_BYTES),), dtype=torch.uint8, device=a.device)
                _tail_ws[key] = ws
            check(lib.clibd_gemm_bf16_nt_ws(a.data_ptr(), lda, w.data_ptr(), ldw, M, N, K, C.byref(ep), ws.data_ptr(), ws.numel(), _stream()), "gemm_bf16_nt_ws")
            return
    check(lib.clibd_gemm_bf16_nt(a.data_ptr(), lda, w.data_ptr(), ldw, M, N, K, C.byref(ep), _stream()), "gemm_bf16_nt")


def rowsum_finalize(row_sums: torch.Tensor, eps: float, stats: torch.Tensor) -> None:
    """stats[m] = (mean, rstd) from the fold producer's per-slice sums [S, M, 2] (H = 128 S)."""
    _chk(row_sums, F32, "row_sums")
    _chk(stats, F32, "stats")
    S, M, _ = row_sums.shape
    if tuple(stats.shape) != (M, 2):
        raise ValueError("rowsum_finalize: stats must be [M,2]")
    check(_lib.load().clibd_rowsum_finalize(row_sums.data_ptr(), S, M, 128 * S, float(eps), stats.data_ptr(), _stream()), "rowsum_finalize")


def ln_fold_weights(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor]):
    """(wg bf16 [N,K], col_sum_w fp32 [N], bias_folded fp32 [N]) of a frozen Linear behind LayerNorm(gamma, beta): clibd_ln_fold_weights."""
    _chk(w, F32, "w")
    _chk(gamma, F32, "gamma")
    _chk(beta, F32, "beta")
    N, K = w.shape
    wg = torch.empty((N, K), dtype=BF16, device=w.device)
    s = torch.empty((N,), dtype=F32, device=w.device)
    bp = torch.empty((N,), dtype=F32, device=w.device)
    check(_lib.load().clibd_ln_fold_weights(w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr() if bias is not None else None, N, K,
                                            wg.data_ptr(), s.data_ptr(), bp.data_ptr(), _stream()), "ln_fold_weights")
    return wg, s, bp


def quantize_rows_fp8(w: torch.Tensor, act_scale: float):
    """(w_fp8 [N,K] float8_e4m3fn, col_scale [N] fp32) for gemm_fp8_nt; see clibd_quantize_rows_fp8."""
    _chk(w, F32, "w")
    N, K = w.shape
    w8 = torch.empty((N, K), dtype=torch.uint8, device=w.device).view(FP8)
    cs = torch.empty((N,), dtype=F32, device=w.device)
    check(_lib.load().clibd_quantize_rows_fp8(w.data_ptr(), N, K, float(act_scale), w8.data_ptr(), cs.data_ptr(), _stream()), "quantize_rows_fp8")
    return w8, cs


def quantize_rows_fp8_bf16(w: torch.Tensor, act_scale: float = 1.0, l1max: Optional[torch.Tensor] = None):
    """(w_fp8 [N,K] float8_e4m3fn, col_scale [N] fp32) from a bf16 matrix (a transposed weight shadow): the W operand of gemm_fp8_dgrad_nt.
    l1max (fp32 scalar tensor, zeroed by the caller): raised to the largest row l1 norm of the de-quantised image."""
    _chk(w, BF16, "w")
    N, K = w.shape
    w8 = torch.empty((N, K), dtype=torch.uint8, device=w.device).view(FP8)
    cs = torch.empty((N,), dtype=F32, device=w.device)
    if l1max is not None:
        _chk(l1max, F32, "l1max")
    check(_lib.load().clibd_quantize_rows_fp8_bf16(w.data_ptr(), N, K, float(act_scale), w8.data_ptr(), cs.data_ptr(), _p(l1max), _stream()),
          "quantize_rows_fp8_bf16")
    return w8, cs


def gemm_fp8_dgrad_nt(a: torch.Tensor, w: torch.Tensor, col_scale: torch.Tensor, *, a_row_dequant: Optional[torch.Tensor] = None,
                      aux: Optional[torch.Tensor] = None, act: int = ACT_NONE, out_bf16: Optional[torch.Tensor] = None,
                      out_fp8: Optional[torch.Tensor] = None, out_fp8_scale: float = 0.0, out_bf16_dual: Optional[torch.Tensor] = None) -> None:
    """8-bit dgrad (clibd_gemm_fp8_dgrad_nt): a [M,K] e4m3 gradient rows with per-row scales (a_row_dequant [M] = 1 / scale), w [N,K] the
    transposed weight from quantize_rows_fp8_bf16.  Forms: act NONE -> out_bf16 | ACT_ADD_AUX + aux -> out_bf16 (both need a_row_dequant) |
    ACT_MUL_AUX (bf16 aux) / ACT_MUL_AUX_U8 (one-byte gelu' codes) -> out_fp8 = e4m3(value * out_fp8_scale), which keeps a's row scales;
    out_bf16_dual (full fine-tune; MUL_AUX forms, needs a_row_dequant): also the de-scaled value as bf16, the weight gradient's operand."""
    _chk(a, FP8, "a", contiguous=False)
    _chk(w, FP8, "w", contiguous=False)
    _chk(col_scale, F32, "col_scale")
    lda, ldw = _rowmajor(a, "a"), _rowmajor(w, "w")
    M, K = a.shape
    N, K2 = w.shape
    if K != K2 or col_scale.numel() != N:
        raise ValueError("gemm_fp8_dgrad_nt: shape mismatch")
    if act not in (ACT_NONE, ACT_ADD_AUX, ACT_MUL_AUX, ACT_MUL_AUX_U8, ACT_MUL_AUX_E12):
        raise ValueError("gemm_fp8_dgrad_nt: act must be NONE, ADD_AUX, MUL_AUX, MUL_AUX_U8 or MUL_AUX_E12")
    if (act == ACT_NONE) != (aux is None):
        raise ValueError("gemm_fp8_dgrad_nt: aux comes with ADD_AUX / MUL_AUX only")
    ep = GemmEpilogue()
    ep.split_k = 1
    ep.act = act
    if aux is not None:
        _chk(aux, torch.uint8 if act in (ACT_MUL_AUX_U8, ACT_MUL_AUX_E12) else BF16, "aux", contiguous=False)
        if tuple(aux.shape) != (M, 3 * N // 2 if act == ACT_MUL_AUX_E12 else N):
            raise ValueError("gemm_fp8_dgrad_nt: aux must be [M,N] ([M,3N/2] bytes for the e4m7 form)")
        ep.aux_bf16, ep.ld_aux = aux.data_ptr(), _rowmajor(aux, "aux")
    if act in (ACT_MUL_AUX, ACT_MUL_AUX_U8, ACT_MUL_AUX_E12):
        if out_fp8 is None or out_bf16 is not None or not out_fp8_scale > 0:
            raise ValueError("gemm_fp8_dgrad_nt: the MUL_AUX form writes out_fp8 with a positive out_fp8_scale")
        _chk(out_fp8, FP8, "out_fp8", contiguous=False)
        if tuple(out_fp8.shape) != (M, N):
            raise ValueError("gemm_fp8_dgrad_nt: out_fp8 must be [M,N]")
        ep.out_bf16, ep.ld_out_bf16 = out_fp8.data_ptr(), _rowmajor(out_fp8, "out_fp8")
        if out_bf16_dual is not None:
            if a_row_dequant is None:
                raise ValueError("gemm_fp8_dgrad_nt: out_bf16_dual needs a_row_dequant")
            _chk(out_bf16_dual, BF16, "out_bf16_dual", contiguous=False)
            if tuple(out_bf16_dual.shape) != (M, N):
                raise ValueError("gemm_fp8_dgrad_nt: out_bf16_dual must be [M,N]")
            ep.out_pre_bf16, ep.ld_pre = out_bf16_dual.data_ptr(), _rowmajor(out_bf16_dual, "out_bf16_dual")
    else:
        if out_bf16_dual is not None:
            raise ValueError("gemm_fp8_dgrad_nt: out_bf16_dual comes with the MUL_AUX forms only")
        if out_bf16 is None or out_fp8 is not None or a_row_dequant is None:
            raise ValueError("gemm_fp8_dgrad_nt: the bf16-output forms need out_bf16 and a_row_dequant")
        _chk(out_bf16, BF16, "out_bf16", contiguous=False)
        if tuple(out_bf16.shape) != (M, N):
            raise ValueError("gemm_fp8_dgrad_nt: out_bf16 must be [M,N]")
        ep.out_bf16, ep.ld_out_bf16 = out_bf16.data_ptr(), _rowmajor(out_bf16, "out_bf16")
        out_fp8_scale = 0.0
    if a_row_dequant is not None:
        _chk(a_row_dequant, F32, "a_row_dequant")
        if a_row_dequant.numel() != M:
            raise ValueError("gemm_fp8_dgrad_nt: a_row_dequant must have M elements")
    check(_lib.load().clibd_gemm_fp8_dgrad_nt(a.data_ptr(), lda, w.data_ptr(), ldw, M, N, K, col_scale.data_ptr(), _p(a_row_dequant),
                                              float(out_fp8_scale), C.byref(ep), _stream()), "gemm_fp8_dgrad_nt")


def gemm_fp8_nt(
    a: torch.Tensor,
    w: torch.Tensor,
    col_scale: torch.Tensor,
    *,
    bias: torch.Tensor,
    rank_u: Optional[torch.Tensor] = None,
    rank_v: Optional[torch.Tensor] = None,
    gelu_out_fp8: Optional[torch.Tensor] = None,
    gelu_out_scale: float = 0.0,
    out_pre: Optional[torch.Tensor] = None,
    residual: Optional[torch.Tensor] = None,
    out_bf16: Optional[torch.Tensor] = None,
    out_f32: Optional[torch.Tensor] = None,
    drop: Optional["Drop"] = None,
) -> None:
    """fp8-forward GEMM: epilogue((a[M,K] @ w[N,K]^T) * col_scale[n]) with OCP e4m3 operands; see clibd_gemm_fp8_nt.
    Forms: -> out_bf16 (optionally + rank update) | gelu -> gelu_out_fp8 (= fp8(gelu * gelu_out_scale)) + out_pre (gelu') |
    [dropout] + residual -> out_f32."""
    _chk(a, FP8, "a", contiguous=False)
    _chk(w, FP8, "w", contiguous=False)
    _chk(col_scale, F32, "col_scale")
    _chk(bias, F32, "bias")
    lda, ldw = _rowmajor(a, "a"), _rowmajor(w, "w")
    M, K = a.shape
    N, K2 = w.shape
    if K != K2 or col_scale.numel() != N or bias.numel() != N:
        raise ValueError("gemm_fp8_nt: shape mismatch")
    ep = GemmEpilogue()
    ep.split_k = 1
    ep.bias = bias.data_ptr()
    if drop is not None and drop.thr16 > 0:
        ep.drop_seed, ep.drop_thr16, ep.drop_scale, ep.drop_ld = drop.seed, drop.thr16, drop.scale, N
    if rank_u is not None or rank_v is not None:
        _chk(rank_u, BF16, "rank_u", contiguous=False)
        _chk(rank_v, BF16, "rank_v")
        if rank_u.shape[0] != M or rank_u.shape[1] < 8 or tuple(rank_v.shape) != (N, 8):
            raise ValueError("gemm_fp8_nt: rank_u must be [M,>=8], rank_v [N,8]")
        ep.rank_u, ep.rank_v, ep.ld_rank_u = rank_u.data_ptr(), rank_v.data_ptr(), _rowmajor(rank_u, "rank_u")
    for name, t, dt in (("out_pre", out_pre, BF16), ("out_bf16", out_bf16, BF16), ("out_f32", out_f32, F32), ("residual", residual, F32),
                        ("gelu_out_fp8", gelu_out_fp8, FP8)):
        if t is not None:
            _chk(t, dt, name, contiguous=False)
            if tuple(t.shape) != (M, N):
                raise ValueError(f"gemm_fp8_nt: {name} must be [M,N]")
    out_scale = 0.0
    if gelu_out_fp8 is not None:
        if out_pre is None or out_bf16 is not None or gelu_out_scale <= 0:
            raise ValueError("gemm_fp8_nt: the gelu form needs out_pre and a positive gelu_out_scale, and no out_bf16")
        ep.act = ACT_GELU_SAVE_GRAD
        ep.out_pre_bf16, ep.ld_pre = out_pre.data_ptr(), _rowmajor(out_pre, "out_pre")
        ep.out_bf16, ep.ld_out_bf16 = gelu_out_fp8.data_ptr(), _rowmajor(gelu_out_fp8, "gelu_out_fp8")
        out_scale = float(gelu_out_scale)
    elif out_bf16 is not None:
        ep.out_bf16, ep.ld_out_bf16 = out_bf16.data_ptr(), _rowmajor(out_bf16, "out_bf16")
    if residual is not None:
        ep.residual_f32, ep.ld_res = residual.data_ptr(), _rowmajor(residual, "residual")
    if out_f32 is not None:
        ep.out_f32, ep.ld_out_f32 = out_f32.data_ptr(), _rowmajor(out_f32, "out_f32")
    check(_lib.load().clibd_gemm_fp8_nt(a.data_ptr(), lda, w.data_ptr(), ldw, M, N, K, col_scale.data_ptr(), out_scale, C.byref(ep), _stream()),
          "gemm_fp8_nt")


def transpose_bf16(x: torch.Tensor, pad_to: int = 64, colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R,C] bf16 -> [C, R_pad] bf16 (zero padded along R to a multiple of `pad_to`).
    colsum (fp32 [C], accumulates): column sums of x in the same pass (bias gradient beside the weight gradient's dy^T), through a
    partials workspace and a fixed-order second kernel (clibd_transpose_colsum_bf16_ws: no float atomics, bit-reproducible)."""
    _chk(x, BF16, "x", contiguous=False)
    ld = _rowmajor(x, "x")
    R, Cc = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty((Cc, Rp), dtype=BF16, device=x.device)
    if colsum is not None:
        _chk(colsum, F32, "colsum")
        if colsum.numel() != Cc:
            raise ValueError("transpose_bf16: colsum must have C elements")
        lib = _lib.load()
        need = int(lib.clibd_transpose_colsum_workspace_bytes(Rp, Cc))
        ws = torch.empty(((need + 3) // 4,), dtype=F32, device=x.device)
        check(lib.clibd_transpose_colsum_bf16_ws(x.data_ptr(), ld, R, Cc, out.data_ptr(), Rp, colsum.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()),
              "transpose_colsum_bf16_ws")
    else:
        check(_lib.load().clibd_transpose_bf16(x.data_ptr(), ld, R, Cc, out.data_ptr(), Rp, _stream()), "transpose_bf16")
    return out


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    _chk(x, F32, "x")
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(_lib.load().clibd_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "cast_f32_to_bf16")
    return out


def cast_transpose_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [R,C] -> bf16 [C,R]."""
    _chk(x, F32, "x")
    R, Cc = x.shape
    out = torch.empty((Cc, R), dtype=BF16, device=x.device)
    check(_lib.load().clibd_cast_transpose_f32_to_bf16(x.data_ptr(), R, Cc, out.data_ptr(), _stream()), "cast_transpose")
    return out


def layernorm_fwd(x, gamma, beta, eps, *, y_bf16=None, y_f32=None, stats=None, lora_a=None, t_out=None, drop=None,
                  y_fp8=None, fp8_scale: float = 0.0) -> None:
    """y_fp8 (fp8-forward mode): also / only emit e4m3(y * fp8_scale), the operand of the next gemm_fp8_nt."""
    _chk(x, F32, "x")
    M, H = x.shape
    _chk(gamma, F32, "gamma")
    _chk(beta, F32, "beta")
    for nm, t, dt, shape in (("y_bf16", y_bf16, BF16, (M, H)), ("y_f32", y_f32, F32, (M, H)), ("stats", stats, F32, (M, 2)),
                             ("lora_a", lora_a, BF16, (8, H)), ("t_out", t_out, BF16, (M, 8)), ("y_fp8", y_fp8, FP8, (M, H))):
        if t is not None:
            _chk(t, dt, nm)
            if tuple(t.shape) != shape:
                raise ValueError(f"layernorm_fwd: {nm} must be {shape}, got {tuple(t.shape)}")
    if y_fp8 is not None:
        d = drop if (drop is not None and drop.thr16 > 0) else Drop(0.0, 0)
        check(_lib.load().clibd_layernorm_fwd_fp8(x.data_ptr(), M, H, gamma.data_ptr(), beta.data_ptr(), float(eps), _p(y_bf16), _p(y_f32),
                                                  _p(stats), _p(lora_a), _p(t_out), d.seed, d.thr16, d.scale, y_fp8.data_ptr(), float(fp8_scale),
                                                  _stream()), "layernorm_fwd_fp8")
        return
    if drop is not None and drop.thr16 > 0:
        check(_lib.load().clibd_layernorm_fwd_drop(x.data_ptr(), M, H, gamma.data_ptr(), beta.data_ptr(), float(eps), _p(y_bf16), _p(y_f32),
                                                   _p(stats), _p(lora_a), _p(t_out), drop.seed, drop.thr16, drop.scale, _stream()),
              "layernorm_fwd_drop")
        return
    check(_lib.load().clibd_layernorm_fwd(x.data_ptr(), M, H, gamma.data_ptr(), beta.data_ptr(), float(eps), _p(y_bf16), _p(y_f32),
                                          _p(stats), _p(lora_a), _p(t_out), _stream()), "layernorm_fwd")


def layernorm_bwd(dy, x, stats, gamma, *, dres=None, dx_f32=None, dx_bf16=None, drop=None, dgamma=None, dbeta=None,
                  dres_bf16=None, dx_res_bf16=None, dx_fp8=None, row_dequant=None) -> None:
    """dgamma / dbeta (fp32 [H], accumulate): the LayerNorm parameter gradients in the same pass (full fine-tune mode).
    dres_bf16 / dx_res_bf16: the residual gradient travels as bf16 (clibd_layernorm_bwd_res16): dx = LN'(dy) + dres_bf16,
    dx_res_bf16 = bf16(dx) without the dropout mask that dx_bf16 carries; no fp32 input / output stream then.
    dx_fp8 [M,H] e4m3 + row_dequant [M] fp32 (8-bit dgrad, clibd_layernorm_bwd_fp8): the dx_bf16 values once more as the A operand of
    gemm_fp8_dgrad_nt, one power-of-two scale per row; dx_bf16 itself becomes optional."""
    _chk(x, F32, "x")
    M, H = x.shape
    if dx_fp8 is not None or row_dequant is not None:
        if dx_fp8 is None or row_dequant is None or (dgamma is None) != (dbeta is None):
            raise ValueError("layernorm_bwd: dx_fp8 and row_dequant come together (and dgamma with dbeta)")
        _chk(dx_fp8, FP8, "dx_fp8")
        _chk(row_dequant, F32, "row_dequant")
        if tuple(dx_fp8.shape) != (M, H) or row_dequant.numel() != M:
            raise ValueError("layernorm_bwd: dx_fp8 must be [M,H], row_dequant [M]")
        for nm, t, dt in (("dres", dres, F32), ("dres_bf16", dres_bf16, BF16), ("dx_f32", dx_f32, F32), ("dx_res_bf16", dx_res_bf16, BF16), ("dx_bf16", dx_bf16, BF16)):
            if t is not None:
                _chk(t, dt, nm)
                if tuple(t.shape) != (M, H):
                    raise ValueError(f"layernorm_bwd: {nm} shape")
        if dy.dtype not in (BF16, F32) or tuple(dy.shape) != (M, H):
            raise ValueError("layernorm_bwd: dy must be bf16 or fp32 [M,H]")
        _chk(dy, dy.dtype, "dy")
        _chk(stats, F32, "stats")
        _chk(gamma, F32, "gamma")
        d = drop if (drop is not None and drop.thr16 > 0) else Drop(0.0, 0)
        dyb, dyf = (dy.data_ptr(), None) if dy.dtype == BF16 else (None, dy.data_ptr())
        if dgamma is not None:   # full fine-tune under the 8-bit dgrad: the parameter gradients ride along (clibd_layernorm_bwd_fp8_pg)
            _chk(dgamma, F32, "dgamma"); _chk(dbeta, F32, "dbeta")
            if dgamma.numel() != H or dbeta.numel() != H:
                raise ValueError("layernorm_bwd: dgamma / dbeta must have H elements")
            check(_lib.load().clibd_layernorm_bwd_fp8_pg(dyb, dyf, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), M, H, _p(dres), _p(dres_bf16), _p(dx_f32),
                                                         _p(dx_res_bf16), _p(dx_bf16), d.seed, d.thr16, d.scale, dx_fp8.data_ptr(), row_dequant.data_ptr(),
                                                         dgamma.data_ptr(), dbeta.data_ptr(), _stream()), "layernorm_bwd_fp8_pg")
            return
        check(_lib.load().clibd_layernorm_bwd_fp8(dyb, dyf, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), M, H, _p(dres), _p(dres_bf16), _p(dx_f32),
                                                  _p(dx_res_bf16), _p(dx_bf16), d.seed, d.thr16, d.scale, dx_fp8.data_ptr(), row_dequant.data_ptr(),
                                                  _stream()), "layernorm_bwd_fp8")
        return
    if dres_bf16 is not None or dx_res_bf16 is not None:
        if dres is not None:
            raise ValueError("layernorm_bwd: the residual gradient is either fp32 (dres) or bf16 (dres_bf16)")
        for nm, t in (("dres_bf16", dres_bf16), ("dx_res_bf16", dx_res_bf16), ("dx_bf16", dx_bf16)):
            if t is not None:
                _chk(t, BF16, nm)
                if tuple(t.shape) != (M, H):
                    raise ValueError(f"layernorm_bwd: {nm} shape")
        _chk(dy, dy.dtype if dy.dtype in (BF16, F32) else BF16, "dy")
        if tuple(dy.shape) != (M, H):
            raise ValueError("layernorm_bwd: dy shape")
        _chk(stats, F32, "stats")
        _chk(gamma, F32, "gamma")
        d = drop if (drop is not None and drop.thr16 > 0) else Drop(0.0, 0)
        dyb, dyf = (dy.data_ptr(), None) if dy.dtype == BF16 else (None, dy.data_ptr())
        if dx_f32 is not None or dgamma is not None or dbeta is not None:
            # full fine-tune on the bf16 stream: parameter gradients ride along, the bottom layer hands an fp32 gradient to the embeddings
            if dx_f32 is not None:
                _chk(dx_f32, F32, "dx_f32")
                if tuple(dx_f32.shape) != (M, H):
                    raise ValueError("layernorm_bwd: dx_f32 shape")
            if (dgamma is None) != (dbeta is None):
                raise ValueError("layernorm_bwd: dgamma / dbeta come together")
            if dgamma is not None:
                _chk(dgamma, F32, "dgamma"); _chk(dbeta, F32, "dbeta")
                if dgamma.numel() != H or dbeta.numel() != H:
                    raise ValueError("layernorm_bwd: dgamma / dbeta must have H elements")
            check(_lib.load().clibd_layernorm_bwd_any(dyb, dyf, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), M, H, None, _p(dres_bf16), _p(dx_f32),
                                                      _p(dx_res_bf16), _p(dx_bf16), d.seed, d.thr16, d.scale, _p(dgamma), _p(dbeta), _stream()),
                  "layernorm_bwd_any")
            return
        check(_lib.load().clibd_layernorm_bwd_res16(dyb, dyf, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), M, H, _p(dres_bf16), _p(dx_res_bf16),
                                                    _p(dx_bf16), d.seed, d.thr16, d.scale, _stream()), "layernorm_bwd_res16")
        return
    if dy.dtype == BF16:
        _chk(dy, BF16, "dy")
        dyb, dyf = dy.data_ptr(), None
    else:
        _chk(dy, F32, "dy")
        dyb, dyf = None, dy.data_ptr()
    if tuple(dy.shape) != (M, H):
        raise ValueError("layernorm_bwd: dy shape")
    _chk(stats, F32, "stats")
    _chk(gamma, F32, "gamma")
    for nm, t, dt in (("dres", dres, F32), ("dx_f32", dx_f32, F32), ("dx_bf16", dx_bf16, BF16)):
        if t is not None:
            _chk(t, dt, nm)
            if tuple(t.shape) != (M, H):
                raise ValueError(f"layernorm_bwd: {nm} shape")
    if dgamma is not None or dbeta is not None:
        _chk(dgamma, F32, "dgamma"); _chk(dbeta, F32, "dbeta")
        if dgamma.numel() != H or dbeta.numel() != H:
            raise ValueError("layernorm_bwd: dgamma / dbeta must have H elements")
        d = drop if (drop is not None and drop.thr16 > 0) else Drop(0.0, 0)
        check(_lib.load().clibd_layernorm_bwd_pg(dyb, dyf, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), M, H, _p(dres), _p(dx_f32),
                                                 _p(dx_bf16), d.seed, d.thr16, d.scale, dgamma.data_ptr(), dbeta.data_ptr(), _stream()),
              "layernorm_bwd_pg")
        return
    if drop is not None and drop.thr16 > 0:
        check(_lib.load().clibd_layernorm_bwd_drop(dyb, dyf, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), M, H, _p(dres), _p(dx_f32),
                                                   _p(dx_bf16), drop.seed, drop.thr16, drop.scale, _stream()), "layernorm_bwd_drop")
        return
    check(_lib.load().clibd_layernorm_bwd(dyb, dyf, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), M, H, _p(dres), _p(dx_f32),
                                          _p(dx_bf16), _stream()), "layernorm_bwd")


def attention_fwd(qkv: torch.Tensor, B: int, S: int, nheads: int, key_mask: Optional[torch.Tensor], out: torch.Tensor,
                  nq: Optional[int] = None, drop=None, out_fp8_scale: float = 0.0, lse: Optional[torch.Tensor] = None,
                  o_lo: Optional[torch.Tensor] = None) -> None:
    """nq: evaluate only the first nq query rows of every sequence; `out` is then [B*nq, H].
    out_fp8_scale > 0 (fp8-forward mode): `out` is float8_e4m3fn and receives e4m3(o * out_fp8_scale).
    lse (fp32 [B*nheads*S]) + o_lo (bf16 [B*S, H]): training forward for attention_bwd_sp (clibd_attention_fwd_save)."""
    _chk(qkv, BF16, "qkv")
    if lse is not None or o_lo is not None:
        _chk(out, BF16, "out"); _chk(lse, F32, "lse"); _chk(o_lo, BF16, "o_lo")
        H_ = nheads * 64
        if (nq is not None and nq != S) or out_fp8_scale > 0 or tuple(qkv.shape) != (B * S, 3 * H_) or tuple(out.shape) != (B * S, H_) \
                or tuple(o_lo.shape) != (B * S, H_) or lse.numel() != B * nheads * S:
            raise ValueError("attention_fwd: the saving form needs nq = S, bf16 out [B*S,H], o_lo [B*S,H], lse [B*nheads*S]")
        if key_mask is not None:
            _chk(key_mask, I32, "key_mask")
        d = drop if (drop is not None and drop.thr16 > 0) else Drop(0.0, 0)
        check(_lib.load().clibd_attention_fwd_save(qkv.data_ptr(), B, S, nheads, _p(key_mask), out.data_ptr(), d.seed, d.thr16, d.scale,
                                                   lse.data_ptr(), o_lo.data_ptr(), _stream()), "attention_fwd_save")
        return
    _chk(out, FP8 if out_fp8_scale > 0 else BF16, "out")
    H = nheads * 64
    nq = S if nq is None else nq
    if tuple(qkv.shape) != (B * S, 3 * H) or tuple(out.shape) != (B * nq, H):
        raise ValueError("attention_fwd: qkv must be [B*S,3H], out [B*nq,H] with H = 64*nheads")
    if key_mask is not None:
        _chk(key_mask, I32, "key_mask")
        if tuple(key_mask.shape) != (B, S):
            raise ValueError("attention_fwd: key_mask must be [B,S]")
    if out_fp8_scale > 0:
        d = drop if (drop is not None and drop.thr16 > 0) else Drop(0.0, 0)
        check(_lib.load().clibd_attention_fwd_fp8(qkv.data_ptr(), B, S, nheads, _p(key_mask), out.data_ptr(), nq, nq, d.seed, d.thr16, d.scale,
                                                  float(out_fp8_scale), _stream()), "attention_fwd_fp8")
        return
    if drop is not None and drop.thr16 > 0:
        check(_lib.load().clibd_attention_fwd_drop(qkv.data_ptr(), B, S, nheads, _p(key_mask), out.data_ptr(), nq, nq, drop.seed, drop.thr16,
                                                   drop.scale, _stream()), "attention_fwd_drop")
        return
    check(_lib.load().clibd_attention_fwd(qkv.data_ptr(), B, S, nheads, _p(key_mask), out.data_ptr(), nq, nq, _stream()), "attention_fwd")


def attention_bwd(qkv, dout, B, S, nheads, key_mask, dqkv, nq: Optional[int] = None, drop=None) -> None:
    """nq: `dout` is [B*nq, H] — the gradient of the first nq query rows of every sequence (the rest is zero)."""
    _chk(qkv, BF16, "qkv")
    _chk(dout, BF16, "dout")
    _chk(dqkv, BF16, "dqkv")
    H = nheads * 64
    nq = S if nq is None else nq
    if tuple(qkv.shape) != (B * S, 3 * H) or tuple(dout.shape) != (B * nq, H) or tuple(dqkv.shape) != (B * S, 3 * H):
        raise ValueError("attention_bwd: shapes")
    if key_mask is not None:
        _chk(key_mask, I32, "key_mask")
    if drop is not None and drop.thr16 > 0:
        check(_lib.load().clibd_attention_bwd_drop(qkv.data_ptr(), dout.data_ptr(), B, S, nheads, _p(key_mask), dqkv.data_ptr(), nq, nq,
                                                   drop.seed, drop.thr16, drop.scale, _stream()), "attention_bwd_drop")
        return
    check(_lib.load().clibd_attention_bwd(qkv.data_ptr(), dout.data_ptr(), B, S, nheads, _p(key_mask), dqkv.data_ptr(), nq, nq, _stream()),
          "attention_bwd")


def attention_bwd_sp(qkv, dout, out, o_lo, lse, B, S, nheads, dqkv, drop=None) -> None:
    """Single-pass backward from the forward's saved output, rounding residual and log-sum-exp (clibd_attention_bwd_sp):
    full sequences, no key mask, S <= 224."""
    H = nheads * 64
    for nm, t, shape in (("qkv", qkv, (B * S, 3 * H)), ("dout", dout, (B * S, H)), ("out", out, (B * S, H)), ("o_lo", o_lo, (B * S, H)),
                         ("dqkv", dqkv, (B * S, 3 * H))):
        _chk(t, BF16, nm)
        if tuple(t.shape) != shape:
            raise ValueError(f"attention_bwd_sp: {nm} must be {shape}")
    _chk(lse, F32, "lse")
    if lse.numel() != B * nheads * S:
        raise ValueError("attention_bwd_sp: lse must have B*nheads*S elements")
    d = drop if (drop is not None and drop.thr16 > 0) else Drop(0.0, 0)
    check(_lib.load().clibd_attention_bwd_sp(qkv.data_ptr(), dout.data_ptr(), out.data_ptr(), o_lo.data_ptr(), lse.data_ptr(), B, S, nheads,
                                             dqkv.data_ptr(), d.seed, d.thr16, d.scale, _stream()), "attention_bwd_sp")


def lora_pack(a_q, a_v, b_q, b_v, v_fwd, v_bwd, a_cat, w_dt) -> None:
    H = a_q.shape[1]
    for nm, t, shape in (("a_q", a_q, (4, H)), ("a_v", a_v, (4, H)), ("b_q", b_q, (H, 4)), ("b_v", b_v, (H, 4))):
        _chk(t, F32, nm)
        if tuple(t.shape) != shape:
            raise ValueError(f"lora_pack: {nm} must be {shape}")
    for nm, t, shape in (("v_fwd", v_fwd, (3 * H, 8)), ("v_bwd", v_bwd, (H, 8)), ("a_cat", a_cat, (8, H)), ("w_dt", w_dt, (16, 3 * H))):
        _chk(t, BF16, nm)
        if tuple(t.shape) != shape:
            raise ValueError(f"lora_pack: {nm} must be {shape}")
    check(_lib.load().clibd_lora_pack(a_q.data_ptr(), a_v.data_ptr(), b_q.data_ptr(), b_v.data_ptr(), H, v_fwd.data_ptr(),
                                      v_bwd.data_ptr(), a_cat.data_ptr(), w_dt.data_ptr(), _stream()), "lora_pack")


def lora_down_proj(x: torch.Tensor, a_cat: torch.Tensor) -> torch.Tensor:
    """t [M,8] = bf16(x [M,H] . a_cat [8,H]^T): the adapters' down-projection as its own kernel (rank slots after the first, r > 4)."""
    _chk(x, BF16, "x", contiguous=False)
    _chk(a_cat, BF16, "a_cat")
    M, H = x.shape
    if tuple(a_cat.shape) != (8, H):
        raise ValueError("lora_down_proj: a_cat must be [8,H]")
    t = torch.empty((M, 8), dtype=BF16, device=x.device)
    check(_lib.load().clibd_lora_down_proj(x.data_ptr(), _rowmajor(x, "x"), a_cat.data_ptr(), M, H, t.data_ptr(), _stream()), "lora_down_proj")
    return t


def _lora_workspace(M: int, H: int, device, workspace):
    """(pointer, bytes) of the adapters' partials workspace: the caller's buffer, a fresh one (deterministic sums, no contended
    float atomics), or (None, 0) when `workspace` is False (the float atomics of rounds 1-4) / the shape takes the VALU kernel."""
    if workspace is False:
        return None, 0
    need = int(_lib.load().clibd_lora_workspace_bytes(M, H))
    if need == 0:
        return None, 0
    if workspace is None or workspace is True:
        workspace = torch.empty((need,), dtype=torch.uint8, device=device)
    if workspace.numel() * workspace.element_size() < need:
        raise ValueError("lora workspace too small (clibd_lora_workspace_bytes)")
    return workspace, need


def lora_wgrad(dqkv, x, t, dt, dA_q, dA_v, dB_q, dB_v, workspace=None) -> None:
    _chk(dqkv, BF16, "dqkv")
    _chk(x, BF16, "x")
    _chk(t, BF16, "t")
    _chk(dt, BF16, "dt")
    M, H = x.shape
    if tuple(dqkv.shape) != (M, 3 * H) or tuple(t.shape) != (M, 8) or dt.shape[0] != M or dt.shape[1] < 8:
        raise ValueError("lora_wgrad: shapes")
    for nm, g, shape in (("dA_q", dA_q, (4, H)), ("dA_v", dA_v, (4, H)), ("dB_q", dB_q, (H, 4)), ("dB_v", dB_v, (H, 4))):
        _chk(g, F32, nm)
        if tuple(g.shape) != shape:
            raise ValueError(f"lora_wgrad: {nm} must be {shape}")
    ws, nb = _lora_workspace(M, H, x.device, workspace)
    check(_lib.load().clibd_lora_wgrad(dqkv.data_ptr(), 3 * H, x.data_ptr(), t.data_ptr(), dt.data_ptr(), dt.shape[1], M, H,
                                       dA_q.data_ptr(), dA_v.data_ptr(), dB_q.data_ptr(), dB_v.data_ptr(),
                                       ws.data_ptr() if ws is not None else None, nb, _stream()), "lora_wgrad")


def lora_backward(dqkv, x, t, w_dt, dt, dA_q, dA_v, dB_q, dB_v, workspace=None) -> None:
    """dt = dqkv . w_dt^T (written, bf16 [M,16]) and the adapters' four parameter gradients (accumulated); see clibd_lora_backward.
    workspace: None = a partials buffer is allocated (deterministic, atomics-free sums), a uint8 tensor = the caller's, False = float atomics."""
    _chk(dqkv, BF16, "dqkv")
    _chk(x, BF16, "x")
    _chk(t, BF16, "t")
    _chk(w_dt, BF16, "w_dt")
    _chk(dt, BF16, "dt")
    M, H = x.shape
    if tuple(dqkv.shape) != (M, 3 * H) or tuple(t.shape) != (M, 8) or tuple(w_dt.shape) != (16, 3 * H) or tuple(dt.shape) != (M, 16):
        raise ValueError("lora_backward: shapes")
    for nm, g, shape in (("dA_q", dA_q, (4, H)), ("dA_v", dA_v, (4, H)), ("dB_q", dB_q, (H, 4)), ("dB_v", dB_v, (H, 4))):
        _chk(g, F32, nm)
        if tuple(g.shape) != shape:
            raise ValueError(f"lora_backward: {nm} must be {shape}")
    ws, nb = _lora_workspace(M, H, x.device, workspace)
    check(_lib.load().clibd_lora_backward(dqkv.data_ptr(), 3 * H, x.data_ptr(), t.data_ptr(), w_dt.data_ptr(), dt.data_ptr(), 16, M, H,
                                          dA_q.data_ptr(), dA_v.data_ptr(), dB_q.data_ptr(), dB_v.data_ptr(),
                                          ws.data_ptr() if ws is not None else None, nb, _stream()), "lora_backward")


def patchify(image: torch.Tensor) -> torch.Tensor:
    """fp32 [B,3,224,224] in [0,1] — or the dataset's uint8 bytes, read as u8 / 255 (bit-identical to the fp32 form of the same
    image: clibd_patchify_u8) — -> bf16 patch matrix [B*196, 768]."""
    _chk(image, torch.uint8 if image.dtype == torch.uint8 else F32, "image")
    if image.dim() != 4 or tuple(image.shape[1:]) != (3, 224, 224):
        raise ValueError("patchify: image must be [B,3,224,224]")
    B = image.shape[0]
    out = torch.empty((B * 196, 768), dtype=BF16, device=image.device)
    if image.dtype == torch.uint8:
        check(_lib.load().clibd_patchify_u8(image.data_ptr(), B, out.data_ptr(), _stream()), "patchify_u8")
    else:
        check(_lib.load().clibd_patchify(image.data_ptr(), B, out.data_ptr(), _stream()), "patchify")
    return out


def vit_assemble_tokens(proj: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, B: int) -> torch.Tensor:
    """tok[b,0] = cls + pos[0]; tok[b,1+p] = proj[b*P+p] + pos[1+p]  -> fp32 [B,S,H]"""
    _chk(proj, F32, "proj")
    _chk(cls, F32, "cls")
    _chk(pos, F32, "pos")
    H = proj.shape[1]
    S = pos.numel() // H
    if proj.shape[0] != B * (S - 1) or cls.numel() != H:
        raise ValueError("vit_assemble_tokens: shapes")
    tok = torch.empty((B, S, H), dtype=F32, device=proj.device)
    check(_lib.load().clibd_vit_assemble_tokens(proj.data_ptr(), cls.data_ptr(), pos.data_ptr(), B, S, H, tok.data_ptr(), _stream()),
          "vit_assemble_tokens")
    return tok


def gelu_bwd(dy: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    _chk(dy, BF16, "dy")
    _chk(pre, BF16, "pre")
    if dy.shape != pre.shape:
        raise ValueError("gelu_bwd: shapes")
    dx = torch.empty_like(dy)
    check(_lib.load().clibd_gelu_bwd_bf16(dy.data_ptr(), pre.data_ptr(), dy.numel(), dx.data_ptr(), _stream()), "gelu_bwd")
    return dx


def bert_embed(ids, token_type, word, pos, typ, out) -> None:
    _chk(ids, I64, "ids")
    B, S = ids.shape
    H = word.shape[1]
    if token_type is not None:
        _chk(token_type, I64, "token_type")
    for nm, t in (("word", word), ("pos", pos), ("type", typ), ("out", out)):
        _chk(t, F32, nm)
    if pos.shape[0] < S:
        raise ValueError("bert_embed: sequence longer than the position table")
    check(_lib.load().clibd_bert_embed(ids.data_ptr(), _p(token_type), B, S, H, word.shape[0], word.data_ptr(), pos.data_ptr(),
                                       typ.data_ptr(), out.data_ptr(), _stream()), "bert_embed")


def softmax_mean_fwd(logits: torch.Tensor, B: int, S: int) -> torch.Tensor:
    _chk(logits, BF16, "logits")
    Cc = logits.shape[1]
    out = torch.empty((B, Cc), dtype=F32, device=logits.device)
    check(_lib.load().clibd_softmax_mean_fwd(logits.data_ptr(), B, S, Cc, out.data_ptr(), _stream()), "softmax_mean_fwd")
    return out


def softmax_mean_bwd(logits: torch.Tensor, dout: torch.Tensor, B: int, S: int) -> torch.Tensor:
    _chk(logits, BF16, "logits")
    _chk(dout, F32, "dout")
    Cc = logits.shape[1]
    dl = torch.empty_like(logits)
    check(_lib.load().clibd_softmax_mean_bwd(logits.data_ptr(), dout.data_ptr(), B, S, Cc, dl.data_ptr(), _stream()), "softmax_mean_bwd")
    return dl


def token_mean_fwd(x: torch.Tensor) -> torch.Tensor:
    _chk(x, F32, "x")
    B, S, H = x.shape
    out = torch.empty((B, H), dtype=BF16, device=x.device)
    check(_lib.load().clibd_token_mean_fwd(x.data_ptr(), B, S, H, out.data_ptr(), _stream()), "token_mean_fwd")
    return out


def token_mean_bwd(dout: torch.Tensor, S: int) -> torch.Tensor:
    _chk(dout, F32, "dout")
    B, H = dout.shape
    dx = torch.empty((B, S, H), dtype=F32, device=dout.device)
    check(_lib.load().clibd_token_mean_bwd(dout.data_ptr(), B, S, H, dx.data_ptr(), _stream()), "token_mean_bwd")
    return dx


def colsum_bf16(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[N] (fp32) += column sums of x[M,N] (bf16)."""
    _chk(x, BF16, "x", contiguous=False)
    _chk(out, F32, "out")
    M, N = x.shape
    check(_lib.load().clibd_colsum_bf16(x.data_ptr(), _rowmajor(x, "x"), M, N, out.data_ptr(), _stream()), "colsum_bf16")


def gather_rows(x: torch.Tensor) -> torch.Tensor:
    _chk(x, F32, "x")
    B, S, H = x.shape
    out = torch.empty((B, H), dtype=F32, device=x.device)
    check(_lib.load().clibd_gather_rows(x.data_ptr(), B, S, H, out.data_ptr(), _stream()), "gather_rows")
    return out


def scatter_rows(dcls: torch.Tensor, S: int, *, bf16: bool = True, f32: bool = False):
    _chk(dcls, F32, "dcls")
    B, H = dcls.shape
    ob = torch.empty((B * S, H), dtype=BF16, device=dcls.device) if bf16 else None
    of = torch.empty((B * S, H), dtype=F32, device=dcls.device) if f32 else None
    check(_lib.load().clibd_scatter_rows_bf16(dcls.data_ptr(), B, S, H, _p(ob), _p(of), _stream()), "scatter_rows")
    return ob, of


def l2norm_fwd(x: torch.Tensor):
    _chk(x, F32, "x")
    N, D = x.shape
    y = torch.empty_like(x)
    inv = torch.empty((N,), dtype=F32, device=x.device)
    check(_lib.load().clibd_l2norm_fwd(x.data_ptr(), N, D, y.data_ptr(), inv.data_ptr(), _stream()), "l2norm_fwd")
    return y, inv


def l2norm_bwd(dy: torch.Tensor, y: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
    _chk(dy, F32, "dy")
    _chk(y, F32, "y")
    _chk(inv, F32, "inv")
    N, D = y.shape
    dx = torch.empty_like(y)
    check(_lib.load().clibd_l2norm_bwd(dy.data_ptr(), y.data_ptr(), inv.data_ptr(), N, D, dx.data_ptr(), _stream()), "l2norm_bwd")
    return dx


def softce_workspace(Nx: int, N: int, D: int, device) -> torch.Tensor:
    nbytes = _lib.load().clibd_softce_workspace_bytes(Nx, N, D)
    return torch.empty((nbytes,), dtype=torch.uint8, device=device)


def softce_rows_fwd(x, y, labels, row0: int, scale: torch.Tensor, loss_sum: torch.Tensor, ws: torch.Tensor) -> None:
    """`scale` is a 1-element fp32 DEVICE tensor (no host sync on the temperature)."""
    _chk(scale, F32, "scale")
    _chk(x, F32, "x")
    _chk(y, F32, "y")
    _chk(labels, I64, "labels")
    _chk(loss_sum, F32, "loss_sum")
    Nx, D = x.shape
    N = y.shape[0]
    if y.shape[1] != D or labels.numel() != N:
        raise ValueError("softce_rows_fwd: shapes")
    check(_lib.load().clibd_softce_rows_fwd(x.data_ptr(), y.data_ptr(), labels.data_ptr(), Nx, N, D, row0, scale.data_ptr(),
                                            loss_sum.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "softce_rows_fwd")


def softce_rows_bwd(labels, Nx, N, D, row0, scale, weight, dx, dy, dscale, ws, weight_scale=None) -> None:
    _chk(labels, I64, "labels")
    _chk(scale, F32, "scale")
    _chk(dx, F32, "dx")
    _chk(dy, F32, "dy")
    if tuple(dx.shape) != (Nx, D) or tuple(dy.shape) != (N, D):
        raise ValueError("softce_rows_bwd: shapes")
    if dscale is not None:
        _chk(dscale, F32, "dscale")
    check(_lib.load().clibd_softce_rows_bwd(labels.data_ptr(), Nx, N, D, row0, scale.data_ptr(), float(weight), _p(weight_scale), dx.data_ptr(), dy.data_ptr(),
                                            _p(dscale), ws.data_ptr(), ws.numel(), _stream()), "softce_rows_bwd")


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0) -> None:
    for nm, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, F32, nm)
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n:
        raise ValueError("adamw_step: size mismatch")
    check(_lib.load().clibd_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, float(lr), float(beta1), float(beta2),
                                       float(eps), float(weight_decay), int(step), float(grad_scale), _stream()), "adamw_step")


def topk_ip(q: torch.Tensor, keys: torch.Tensor, k: int = 5):
    """Exact fp32 inner-product top-k (faiss.IndexFlatIP.search): returns (similarities fp32 [Q,k], indices int64 [Q,k]).
    Streaming: scores never leave the MFMA accumulators' running top-8 lists (no [Q,Nk] matrix)."""
    _chk(q, F32, "q")
    _chk(keys, F32, "keys")
    Q, D = q.shape
    Nk = keys.shape[0]
    if keys.shape[1] != D:
        raise ValueError("topk_ip: dimension mismatch")
    idx = torch.empty((Q, k), dtype=I64, device=q.device)
    sim = torch.empty((Q, k), dtype=F32, device=q.device)
    lib = _lib.load()
    ws = torch.empty((max(int(lib.clibd_topk_ip_workspace_bytes(Q, Nk)), 16),), dtype=torch.uint8, device=q.device)
    check(lib.clibd_topk_ip(q.data_ptr(), keys.data_ptr(), Q, Nk, D, k, idx.data_ptr(), sim.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "topk_ip")
    return sim, idx


class KeyBank:
    """A key bank prepared once for the pre-filtered search: the fp32 keys, their bf16 image and max ||key|| (clibd_topk_prepare_keys).
    The reference builds one faiss.IndexFlatIP per key set and searches it with every query batch (util/util.py:521-528)."""

    MAX_KEYS, MAX_D = 1 << 24, 2048   # the C entry points refuse beyond these too (topk.hip: TK_FAST_MAX_KEYS / TK_FAST_MAX_D)

    def __init__(self, keys: torch.Tensor):
        _chk(keys, F32, "keys")
        Nk, D = keys.shape
        if D % 64 != 0 or D > self.MAX_D or Nk >= self.MAX_KEYS:
            raise ValueError("KeyBank: the pre-filtered search needs D % 64 == 0, D <= 2048 and fewer than 2^24 keys (use topk_ip)")
        self.keys = keys
        self.keys_bf16 = torch.empty((Nk, D), dtype=BF16, device=keys.device)
        self.max_norm = torch.empty((1,), dtype=F32, device=keys.device)
        check(_lib.load().clibd_topk_prepare_keys(keys.data_ptr(), Nk, D, self.keys_bf16.data_ptr(), self.max_norm.data_ptr(), _stream()),
              "topk_prepare_keys")


def topk_ip_fast(q: torch.Tensor, bank: "KeyBank", k: int = 5):
    """clibd_topk_ip_fast: the top-k of `topk_ip` — same indices, same similarities, bit for bit — from bf16 approximate scores
    followed by an exact re-score of every key within the rigorous error band of the k-th best.  Returns (sim, idx, overflow):
    overflow int32 [Q] flags the queries whose candidate lists were full inside the band (their rows are unspecified; `topk_search`
    re-runs them through the exact kernel)."""
    _chk(q, F32, "q")
    Q, D = q.shape
    Nk = bank.keys.shape[0]
    if bank.keys.shape[1] != D:
        raise ValueError("topk_ip_fast: dimension mismatch")
    idx = torch.empty((Q, k), dtype=I64, device=q.device)
    sim = torch.empty((Q, k), dtype=F32, device=q.device)
    ovf = torch.empty((Q,), dtype=I32, device=q.device)
    lib = _lib.load()
    ws = torch.empty((int(lib.clibd_topk_ip_fast_workspace_bytes(Q, Nk, D)),), dtype=torch.uint8, device=q.device)
    check(lib.clibd_topk_ip_fast(q.data_ptr(), bank.keys.data_ptr(), bank.keys_bf16.data_ptr(), bank.max_norm.data_ptr(), Q, Nk, D, k, idx.data_ptr(),
                                 sim.data_ptr(), ovf.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "topk_ip_fast")
    return sim, idx, ovf


def kmer_tokenize(seq_u8: torch.Tensor, k: int = 5) -> torch.Tensor:
    """uint8 [B,L] ('N'-padded ASCII) -> int64 [B, 1 + L/k] token ids."""
    _chk(seq_u8, torch.uint8, "seq_u8")
    B, L = seq_u8.shape
    out = torch.empty((B, 1 + L // k), dtype=I64, device=seq_u8.device)
    check(_lib.load().clibd_kmer_tokenize(seq_u8.data_ptr(), B, L, k, out.data_ptr(), _stream()), "kmer_tokenize")
    return out


# ------------------------------------------------------------------------------------------------ full fine-tune mode (f4)
def layernorm_param_grads(dy: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
                          drop: Optional[Drop] = None) -> None:
    """dgamma += sum_m dy*xhat, dbeta += sum_m dy  (dy [M,H] bf16 or fp32; x fp32 [M,H]; stats fp32 [M,2])."""
    if dy.dtype not in (BF16, F32):
        raise TypeError("dy: expected bf16 or fp32")
    _chk(dy, dy.dtype, "dy", contiguous=False)
    _chk(x, F32, "x"); _chk(stats, F32, "stats"); _chk(dgamma, F32, "dgamma"); _chk(dbeta, F32, "dbeta")
    M, H = x.shape
    if tuple(dy.shape) != (M, H) or stats.numel() != 2 * M or dgamma.numel() != H or dbeta.numel() != H:
        raise ValueError("layernorm_param_grads: shape mismatch")
    d = drop if drop is not None else Drop(0.0, 0)
    check(_lib.load().clibd_layernorm_param_grads(dy.data_ptr(), int(dy.dtype == F32), _rowmajor(dy, "dy"), x.data_ptr(), stats.data_ptr(), M, H,
                                                  dgamma.data_ptr(), dbeta.data_ptr(), d.seed, d.thr16, d.scale, _stream()), "layernorm_param_grads")


def batch_sum(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[...] += x.sum(0)   (x fp32 [B, ...], out fp32 with x.shape[1:] elements)."""
    _chk(x, F32, "x"); _chk(out, F32, "out")
    B = x.shape[0]
    R = x.numel() // B
    if out.numel() != R:
        raise ValueError("batch_sum: out must have x.numel() / B elements")
    check(_lib.load().clibd_batch_sum_f32(x.data_ptr(), B, R, out.data_ptr(), _stream()), "batch_sum")


def bert_embed_bwd(ids: torch.Tensor, token_type: Optional[torch.Tensor], de: torch.Tensor, dword: Optional[torch.Tensor],
                   dtype_table: Optional[torch.Tensor]) -> None:
    """Scatter the embedding gradient de [M,H] into the word table (by ids) and the token-type table (accumulating)."""
    _chk(ids, torch.int64, "ids"); _chk(de, F32, "de")
    M, H = de.shape
    if ids.numel() != M:
        raise ValueError("bert_embed_bwd: ids / de mismatch")
    if token_type is not None:
        _chk(token_type, torch.int64, "token_type")
    vocab = dword.shape[0] if dword is not None else 1
    tv = dtype_table.shape[0] if dtype_table is not None else 1
    for t, n in ((dword, "dword"), (dtype_table, "dtype")):
        if t is not None:
            _chk(t, F32, n)
    check(_lib.load().clibd_bert_embed_bwd(ids.data_ptr(), _p(token_type), de.data_ptr(), M, H, vocab, tv, _p(dword), _p(dtype_table), _stream()),
          "bert_embed_bwd")


def slice_rows_cast_bf16(x: torch.Tensor, s0: int, s1: int) -> torch.Tensor:
    """x fp32 [B,S,H] -> bf16 [B*(s1-s0), H]: rows s0..s1-1 of every sequence."""
    _chk(x, F32, "x")
    B, S, H = x.shape
    out = torch.empty((B * (s1 - s0), H), dtype=BF16, device=x.device)
    check(_lib.load().clibd_slice_rows_cast_bf16(x.data_ptr(), B, S, H, s0, s1, out.data_ptr(), _stream()), "slice_rows_cast_bf16")
    return out


def dropout_apply(x: torch.Tensor, drop: Drop) -> torch.Tensor:
    """x * dropout_factor(seed, flat element index)  (fp32): the gradient through y = dropout(.)."""
    _chk(x, F32, "x")
    y = torch.empty_like(x)
    check(_lib.load().clibd_dropout_apply_f32(x.data_ptr(), x.numel(), y.data_ptr(), drop.seed, drop.thr16, drop.scale, _stream()), "dropout_apply")
    return y


_splitk_ws = {}


def gemm_tn_splitk(a: torch.Tensor, b: torch.Tensor, out_f32: torch.Tensor, accumulate: bool = True, colsum: Optional[torch.Tensor] = None) -> bool:
    """out[Na,Nb] (+)= a[M,Na]^T @ b[M,Nb], both operands read in place (clibd_gemm_bf16_tn_splitk): the weight gradient
    dW = dy^T x without transposes; colsum (fp32 [Na]) += column sums of a in the same pass (the bias gradient).
    Returns False (nothing launched) when the shape is outside the kernel's."""
    _chk(a, BF16, "a", contiguous=False); _chk(b, BF16, "b", contiguous=False); _chk(out_f32, F32, "out_f32")
    M, Na = a.shape
    Nb = b.shape[1]
    if b.shape[0] != M or tuple(out_f32.shape) != (Na, Nb):
        raise ValueError("gemm_tn_splitk: shape mismatch")
    if M % 128 or M < 256 or Na % 256 or Nb % 256:
        return False
    lib = _lib.load()
    need = lib.clibd_gemm_splitk_workspace_bytes(Na, Nb)
    key = (a.device, torch.cuda.current_stream(a.device).cuda_stream)
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(((need + 3) // 4,), dtype=F32, device=a.device)   # one workspace per (device, stream), shared with the NT mode
        _splitk_ws[key] = ws
    if colsum is not None:
        _chk(colsum, F32, "colsum")
        if colsum.numel() != Na:
            raise ValueError("gemm_tn_splitk: colsum must have Na elements")
    rc = lib.clibd_gemm_bf16_tn_splitk(a.data_ptr(), _rowmajor(a, "a"), b.data_ptr(), _rowmajor(b, "b"), M, Na, Nb, out_f32.data_ptr(), Nb,
                                       int(accumulate), _p(colsum), ws.data_ptr(), ws.numel() * 4, _stream())
    if rc != 0 and b"shape not supported" in (lib.clibd_last_error() or b""):
        return False    # declined before anything was enqueued: the caller takes the transpose + NT path
    check(rc, "gemm_bf16_tn_splitk")
    return True


def gemm_nt_splitk(a: torch.Tensor, w: torch.Tensor, out_f32: torch.Tensor, accumulate: bool = True) -> bool:
    """out (+)= a @ w.T for a long contraction and few output tiles (weight gradients), through the split-K workspace path
    of the 256x256 kernel.  Returns False (nothing launched) when the shape is outside that path — use gemm_nt(split_k=)."""
    _chk(a, BF16, "a", contiguous=False); _chk(w, BF16, "w", contiguous=False); _chk(out_f32, F32, "out_f32")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or tuple(out_f32.shape) != (M, N):
        raise ValueError("gemm_nt_splitk: shape mismatch")
    if N % 256 or K % 128 or K < 512:
        return False
    lib = _lib.load()
    need = lib.clibd_gemm_splitk_workspace_bytes(M, N)
    key = (a.device, torch.cuda.current_stream(a.device).cuda_stream)
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(((need + 3) // 4,), dtype=F32, device=a.device)   # one workspace per (device, stream)
        _splitk_ws[key] = ws
    rc = lib.clibd_gemm_bf16_nt_splitk(a.data_ptr(), _rowmajor(a, "a"), w.data_ptr(), _rowmajor(w, "w"), M, N, K, out_f32.data_ptr(), N,
                                       int(accumulate), ws.data_ptr(), ws.numel() * 4, _stream())
    if rc != 0 and b"shape not supported" in (lib.clibd_last_error() or b""):
        return False    # declined before anything was enqueued: the caller uses gemm_nt(split_k=)
    check(rc, "gemm_bf16_nt_splitk")
    return True
