"""Tower engines: forward + hand-derived backward of the three CLIBD encoders on the HIP kernels.

Each tower is ONE torch.autograd.Function (inputs = the trainable parameters, output = the tower's [B, D]
embedding): the forward enqueues the fused kernels layer by layer and keeps exactly the activations the
LoRA backward needs; the backward walks the layers in reverse (dgrad only for the frozen base weights,
rank-4 adapter gradients, head weight gradients).  Numerics follow torch.autocast(bf16) in the reference
(epoch/train_epoch.py:42-46): bf16 GEMM/attention operands with fp32 accumulation, fp32 residual stream,
fp32 LayerNorm / softmax statistics.

Reference arithmetic being replaced (all third-party model code the reference delegates to):
  ViT block        timm vision_transformer.Block       via model/image_encoder.py:106-107
  BERT layer       HF BertLayer (post-LN)              via model/dna_encoder.py:137, language_encoder.py:89
  LoRA             model/image_encoder.py:40-46, model/dna_encoder.py:75-77
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


class NotSupportedYet(NotImplementedError):
    pass


# ---- numerics switches of the backward --------------------------------------------------------------------------------------------
# Explicit per-stack settings (`TransformerStack.numerics`, set through `SimpleCLIP.set_numerics(...)` / `Trainer(numerics=...)`);
# the environment variables only give the DEFAULT a stack is constructed with, so two trainers in one process can differ and a
# bench line / training state records which arithmetic produced it (`bench.py` config.numerics, `checkpoint.save_training_state`).
#
# residual_grad ("bf16" default | "fp32"; env CLIBD_RESIDUAL_GRAD).  Frozen-base (LoRA) backward: the gradient of the residual stream
#   travels between blocks as bf16 instead of fp32 — the LayerNorm backward then moves 10 instead of 16 bytes per element and the
#   BERT dgrad GEMMs join the stream with a bf16 add (CLIBD_ACT_ADD_AUX) instead of an fp32 read + write.  The reference's autograd
#   keeps this stream in fp32 (also under autocast); the extra rounding (one bf16 rounding per block-half: ~0.1 % rms each,
#   independent) is budgeted in DESIGN.md §4.  Full fine-tune mode (disable_lora) follows the same switch since round 4 — its bottom
#   layer still hands an fp32 gradient to the embedding backward; tests/test_model_gpu.py::test_full_finetune_residual_grad_streams_agree
#   gates the bf16 stream against the fp32 one on the same model.
# gelu_grad ("bf16" default | "u8" | "e4m7"; env CLIBD_GELU_GRAD).  gelu'(fc1 out) is all the MLP's backward needs (frozen fc1 / fc2 carry no
#   weight gradient that would want the activation), and it lies in [-0.129, 1.129].  "u8" keeps it as ONE BYTE per element
#   (CLIBD_ACT_GELU_SAVE_GRAD_U8, |error| <= 2.5e-3 everywhere; bf16's half-spacing is 2.0e-3 in [0.5, 1), 3.9e-3 in [1, 2), finer
#   below 0.5) and the fc2 dgrad multiplies by the decoded byte (CLIBD_ACT_MUL_AUX_U8): 50 GB less per step at b=2048, 285.1 -> 282.5
#   ms.  Gradient effect at full size: 5e-4 (ViT) / 4e-5 (BERT) relative, invisible against the oracle; on the tiny
#   fixtures the WORST per-tensor error moves by +-10 % (medians unchanged), which the tightest gate (x1.6 of the reference's own
#   autocast error) does not always absorb — so the default stays the bf16 form.  The fp8-forward mode always uses bf16.
#   Round 6: "e4m7" keeps the bf16 value of gelu' in TWELVE bits — sign, 4-bit exponent, bf16's 7 mantissa bits; gelu' lies in [-0.129, 1.129],
#   sixteen binades [2^-14, 2) cover it — so every bf16 gelu' of magnitude >= 2^-14 = 6.1e-5 is reproduced BIT FOR BIT and smaller ones
#   (pre-activations below about -4.55) become zero: the fc2 dgrad multiplies by the same operand as the bf16 form, at 1.5 instead of 2 bytes per
#   element on the fc1 forward's write and the fc2 dgrad's read (CLIBD_ACT_GELU_SAVE_GRAD_E12 / CLIBD_ACT_MUL_AUX_E12; tests/test_ops_gpu.py::
#   test_gelu_grad_e4m7_is_the_bf16_value, tests/test_model_gpu.py::test_gelu_grad_e4m7_towers_equal_the_bf16_form).  Numerically free — and a measured
#   NO-GO as the default: the step is 3.0 % SLOWER with it (294.9 against 286.3 ms, three interleaved pairs on one box, profiles/r06_exp_gelu_grad_e4m7.log).
#   The twelve-bit pack / unpack has no conversion instruction behind it (~100 VALU operations per row of eight columns where the bf16 form takes
#   four v_cvt_pk and the one-byte form eight v_cvt_pk_u8), and the GEMM's epilogue is exposed time: it roughly doubles the VALU work of the fc1
#   epilogue, which the 12.5 % fewer output bytes do not buy back; a hand-packed encode (v_pk_sub_u16 clamp on the bf16 pairs, ~32 operations per
#   row) would still cost more cycles than the bytes save.  Opt-in, for completeness of the byte ladder 2 / 1.5 / 1.
# attn_bwd ("2phase" default | "sp"; env CLIBD_ATTN_BWD).  "2phase" = the kernel that derives the softmax statistics itself and
#   evaluates every score twice; "sp" = single pass (ops.attention_bwd_sp; the training forward then also saves its per-layer output,
#   the rounding residual of that output and the log-sum-exp: 4 more bytes per element of activation memory per layer).  The
#   single-pass form needs full sequences without a key mask, S <= 224 and the bf16 forward; everything else (text tower, the
#   class-row-only last ViT block, fp8-forward mode) keeps the two-phase kernel.  Round 3 measurement: parity-green,
#   1.7 x fewer MFMAs and half the exponentials per head, but as built 1.17 - 1.24 x SLOWER — it stays opt-in.
# ln_fold ("off" default | "on"; env CLIBD_LN_FOLD; pre-LN stacks = the ViT, frozen base, bf16 forward).  A FORWARD switch: norm2 -> mlp.fc1 of a
#   block runs as the algebraic fold  fc1(LN(x)) = rstd (x . (gamma o W)^T - mean s) + b'  — the projection's epilogue also writes a bf16 copy
#   of the residual stream and per-slice row sums, a 5-us kernel turns those into (mean, rstd), and fc1 takes the copy and the gamma-scaled
#   weight image and applies the row terms in its epilogue: the LayerNorm's own HBM pass (6 bytes per element) disappears.  The bf16 rounding
#   moves from LN(x) to x; measured error of the fc1 pre-activation against fp64: the same (tools/ln_fold_study.py: ratio 1.00 at random
#   init, 1.2 with 100-sigma outlier channels).  The backward is the unchanged LayerNorm backward (same statistics).  DESIGN.md §3.1c.
# * dgrad (round 5, BASELINE.json configs[4]): "bf16" — the activation-gradient GEMMs on bf16 operands; "fp8" — the MLP pair and the attention
#   projection of every layer take their gradient operand as e4m3 with ONE power-of-two scale per token row (written by the LayerNorm backward
#   that produces it, from the row maximum it already holds; d(fc1 out) inherits the row scale times a per-layer constant derived from an l1 bound
#   on the fc2 weight, so nothing saturates and no history of maxima is kept) against the transposed frozen weight as e4m3 with one scale per
#   input channel, on v_mfma_f32_16x16x128_f8f6f4.  The QKV dgrad (its operand feeds the adapters' bf16 weight gradients), attention and every
#   weight gradient stay bf16.  Oracle study (full-size towers, tools/fp8_policy_study.py, profiles/r05_exp_fp8_dgrad_study.log): gradient cosine
#   against the bf16 dgrad 0.9998 on the training batch / 0.9875 on a fresh one with BOTH towers on it (MI355X, trained weights: 0.9998-1.0000 on
#   the mean-pooled towers, 0.987-0.9998 with the ViT's too; SimpleCLIP.enable_fp8_dgrad selects towers).  The bf16 residual-gradient stream only
#   (gelu' as bf16 or as its one-byte code); a call whose token count is not a multiple of 4 takes the bf16 GEMMs.  Round 6: also with TRAINABLE base
#   weights (disable_lora, the reference's final recipe) — the e4m3 weight images are re-made every step, the LayerNorm backward writes the e4m3 rows
#   AND the bf16 copy the weight gradient contracts (clibd_layernorm_bwd_fp8_pg), the fc2 dgrad writes d(fc1 out) as e4m3 AND as bf16; every weight /
#   bias / LayerNorm gradient is the bf16 path's arithmetic on those copies.  DESIGN.md §3.1d.
NUMERICS_CHOICES = dict(residual_grad=("bf16", "fp32"), gelu_grad=("bf16", "u8", "e4m7"), attn_bwd=("2phase", "sp"), ln_fold=("off", "on"),
                        dgrad=("bf16", "fp8"))
_NUMERICS_ENV = dict(residual_grad="CLIBD_RESIDUAL_GRAD", gelu_grad="CLIBD_GELU_GRAD", attn_bwd="CLIBD_ATTN_BWD", ln_fold="CLIBD_LN_FOLD",
                     dgrad="CLIBD_DGRAD")


def default_numerics() -> dict:
    """The settings a new stack starts with: the first choice of each switch unless its environment variable names another."""
    out = {}
    for k, choices in NUMERICS_CHOICES.items():
        v = os.environ.get(_NUMERICS_ENV[k], choices[0]).lower()
        out[k] = v if v in choices else choices[0]
    return out


def check_numerics(settings: dict) -> dict:
    for k, v in settings.items():
        if k not in NUMERICS_CHOICES:
            raise ValueError(f"unknown numerics switch {k!r} (known: {sorted(NUMERICS_CHOICES)})")
        if v not in NUMERICS_CHOICES[k]:
            raise ValueError(f"numerics switch {k}: {v!r} is not one of {NUMERICS_CHOICES[k]}")
    return settings


def _mul_aux_act(h, FF: int) -> int:
    """the fc2 dgrad's epilogue for the way gelu' was saved: bf16 [M,FF], one byte per element (uint8 [M,FF]) or the 12-bit e4m7 form (uint8 [M,3FF/2])"""
    if h.dtype == torch.uint8:
        return ops.ACT_MUL_AUX_E12 if h.shape[1] == 3 * FF // 2 else ops.ACT_MUL_AUX_U8
    return ops.ACT_MUL_AUX


@dataclass
class LoraParams:
    """The four adapter matrices of one layer (nn.Parameters, fp32): a_* [4,H], b_* [H,4]."""
    a_q: torch.nn.Parameter
    b_q: torch.nn.Parameter
    a_v: torch.nn.Parameter
    b_v: torch.nn.Parameter

    def tensors(self):
        return [self.a_q, self.b_q, self.a_v, self.b_v]


@dataclass
class LayerSpec:
    """Parameter handles of one transformer layer (fp32 masters living in the nn.Module tree)."""
    qkv_w: Sequence[torch.Tensor]   # one fused [3H,H] tensor or (q,k,v) [H,H] each
    qkv_b: Sequence[torch.Tensor]
    proj_w: torch.Tensor
    proj_b: torch.Tensor
    fc1_w: torch.Tensor
    fc1_b: torch.Tensor
    fc2_w: torch.Tensor
    fc2_b: torch.Tensor
    ln1_w: torch.Tensor             # ViT: norm1 (before attention).  BERT: attention.output.LayerNorm
    ln1_b: torch.Tensor
    ln2_w: torch.Tensor             # ViT: norm2 (before MLP).        BERT: output.LayerNorm
    ln2_b: torch.Tensor
    lora: Optional[LoraParams] = None

    def frozen(self):
        return [*self.qkv_w, *self.qkv_b, self.proj_w, self.proj_b, self.fc1_w, self.fc1_b, self.fc2_w, self.fc2_b, self.ln1_w,
                self.ln1_b, self.ln2_w, self.ln2_b]


class _LayerCache:
    """bf16 device images of one layer's frozen weights: W [N,K] for forward, W^T [K,N] for dgrad.
    fp8-forward mode adds the e4m3 forward images and their per-channel dequantisation factors (w*8, cs_*)."""
    __slots__ = ("wqkv", "wqkv_t", "bqkv", "wo", "wo_t", "bo", "w1", "w1_t", "b1", "w2", "w2_t", "b2", "g1", "be1", "g2", "be2",
                 "v_fwd", "v_bwd", "a_cat", "w_dt", "slot2", "wqkv8", "cs_qkv", "wo8", "cs_o", "w18", "cs_1", "w28", "cs_2",
                 "w1g", "s1", "b1f",   # ln_fold: bf16(gamma2 o W1), its row sums, b1 + W1 beta2
                 "wo_t8", "cs_ot", "w1_t8", "cs_1t", "w2_t8", "cs_2t", "c2")   # dgrad = fp8: e4m3 images of the transposed weights, c2 = d(fc1 out)'s fixed scale


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t if (t.dtype == F32 and t.is_contiguous()) else t.to(F32).contiguous()


def _rank_slots(lp: "LoraParams"):
    """The kernels carry the adapters as rank 4 + 4 (q and v in one MFMA k-slot of 8).  The reference accepts any r > 0
    (image_encoder.py:53; every shipped config uses 4): ranks 1-3 run zero-padded — A gets zero rows, B zero columns, the same
    product — and ranks above 4 as ceil(r / 4) such slots, rows / columns 4j .. 4j+3 of every adapter matrix
    (B A x = sum_j B_j A_j x): every slot after the first takes a pass of its own for its down-projection, rank update and gradients
    (TransformerStack._slot2_*; round 4: any r > 0, as the reference accepts — round 3 stopped at two slots, r <= 8).
    Returns [(a_q, a_v, b_q, b_v)] per slot, each [4,H] / [H,4] fp32."""
    a_q, a_v, b_q, b_v = _f32c(lp.a_q), _f32c(lp.a_v), _f32c(lp.b_q), _f32c(lp.b_v)
    r, H = a_q.shape

    def slot(lo):
        hi = min(lo + 4, r)
        k = hi - lo
        pa = lambda a: a[lo:hi].contiguous() if k == 4 else torch.cat([a[lo:hi], a.new_zeros((4 - k, H))], dim=0)
        pb = lambda b: (b[:, lo:hi] if k == 4 else torch.cat([b[:, lo:hi], b.new_zeros((H, 4 - k))], dim=1)).contiguous()
        return pa(a_q), pa(a_v), pb(b_q), pb(b_v)

    return [slot(lo) for lo in range(0, r, 4)]


class TransformerStack:
    """Shared machinery of the ViT (pre-LN) and BERT (post-LN) encoder stacks."""

    def __init__(self, layers: List[LayerSpec], hidden: int, heads: int, pre_ln: bool, eps: float):
        if hidden != heads * 64:
            raise NotSupportedYet(f"attention kernel is specialised for head_dim 64 (hidden={hidden}, heads={heads})")
        if hidden % 64 or hidden > 1024:
            raise NotSupportedYet("hidden size must be a multiple of 64 and <= 1024")
        self.layers, self.H, self.heads, self.pre_ln, self.eps = layers, hidden, heads, pre_ln, eps
        self.FF = layers[0].fc1_w.shape[0]
        self._cache: List[_LayerCache] = []
        self._cache_key = None
        self.fp8 = None  # fp8-forward mode: [{site: activation scale}] per layer; see enable_fp8
        self._calib = None
        self.numerics = default_numerics()   # backward arithmetic switches (see NUMERICS_CHOICES above); set_numerics() changes them
        self._c2, self._c2_age = None, 0   # dgrad = "fp8" under full fine-tune: the per-layer d(fc1 out) scales and how many refreshes old they are
        self.dgrad8_sites = ("mlp", "proj")   # dgrad = "fp8": which of the covered GEMMs take it ("mlp" = the fc2 -> fc1 pair, "proj"); tools/dgrad8_sites_study.py

    DGRAD8_C2_EVERY = 64   # full fine-tune: refreshes (= steps) between two host reads of the fc2 images' l1 norms

    def set_numerics(self, **settings):
        before = (self.numerics.get("ln_fold"), self.numerics.get("dgrad"))
        self.numerics.update(check_numerics(settings))
        if (self.numerics.get("ln_fold"), self.numerics.get("dgrad")) != before:
            self._cache_key = None   # the fold's / the 8-bit dgrad's weight images are built with the frozen-weight images, only while the switch is on
            self._c2 = None

    def _dgrad8_ok(self) -> bool:
        """dgrad = "fp8" is on and this stack can take it (see NUMERICS_CHOICES); raises for a configuration that cannot."""
        if self.numerics["dgrad"] != "fp8":
            return False
        if self.numerics["residual_grad"] != "bf16":
            raise NotSupportedYet("dgrad=fp8 needs residual_grad=bf16")
        if self.H % 256 or self.H < 512 or self.FF % 256:
            raise NotSupportedYet("dgrad=fp8 needs hidden % 256 == 0, hidden >= 512, intermediate % 256 == 0")
        return True

    # ---- fp8-forward mode (BASELINE.json configs[4]) ----------------------------------------------------------------
    # activation sites, named by the GEMM that consumes them
    FP8_SITES = ("qkv_in", "proj_in", "fc1_in", "fc2_in")
    FP8_SCALES = dict(qkv_in=8.0, proj_in=32.0, fc1_in=8.0, fc2_in=4.0)

    def enable_fp8(self, scales: Optional[dict] = None, amax: Optional[list] = None, margin: float = 2.0, sites: Optional[Sequence[str]] = None):
        """The four forward GEMMs of every layer run on the fp8 MFMA (ops.gemm_fp8_nt): frozen weights are quantised per
        output channel once; activations per tensor, inside the kernels that produce them (LayerNorm, attention, fc1
        epilogue), as e4m3(value * scale) saturating at +-448.  Scales are powers of two, per layer and site:
          * amax = [{site: max |activation|} per layer] (from calibrate(): one bf16 forward over a representative batch):
            scale = 2^floor(log2(448 / (margin * amax)));
          * otherwise the static `scales` / FP8_SCALES for every layer (fits unit-variance LayerNorm outputs, random-init or
            lightly trained towers; pretrained checkpoints with outlier channels want the calibration).
        The backward is unchanged bf16 (it needs gelu', qkv, statistics and — for the adapters — the bf16 LayerNorm output
        only): gradients are those of the bf16 network evaluated at the fp8 forward's activations.  LoRA / frozen-base mode only.
        sites (round 5): a subset of FP8_SITES — ("fc1_in", "fc2_in") runs the MLP of every block on the fp8 MFMA
        and leaves QKV, attention and the projection on bf16 operands (the per-layer dicts then hold those sites only, which is also
        how the oracle is told: a site without a scale is a bf16 site).  The oracle study behind it: profiles/r05_exp_fp8_vit_sites.log."""
        if sites is not None:
            sites = tuple(sites)
            if any(s not in self.FP8_SITES for s in sites):
                raise ValueError(f"enable_fp8: sites must come from {self.FP8_SITES}")
            if set(sites) != set(self.FP8_SITES):
                if set(sites) != {"fc1_in", "fc2_in"}:
                    raise NotSupportedYet("fp8 site selection: only the MLP pair (fc1_in, fc2_in) is built")
        if self.full_mode():
            raise NotSupportedYet("fp8 forward needs frozen base weights (their gradients would need the bf16 GEMM inputs)")
        if self.H % 256 or self.H < 512 or self.FF % 256:
            raise NotSupportedYet("fp8 forward needs hidden % 256 == 0, hidden >= 512, intermediate % 256 == 0")
        base = dict(self.FP8_SCALES, **(scales or {}))
        if sites is not None:
            base = {k: v for k, v in base.items() if k in sites}
        per_layer = []
        for i in range(len(self.layers)):
            d = dict(base)
            if amax is not None:
                for site, v in amax[i].items():
                    if site not in d:
                        continue
                    v = float(v)
                    if v > 0.0 and math.isfinite(v):
                        d[site] = 2.0 ** math.floor(math.log2(448.0 / (margin * v)))
            per_layer.append(d)
        self.fp8 = per_layer
        self._cache_key = None

    def disable_fp8(self):
        self.fp8 = None
        self._cache_key = None

    def calibrate(self, on: bool = True):
        """on: the next bf16 forward records max |activation| per layer and fp8 site (device scalars, no sync);
        off: returns them as [{site: float}] and stops recording."""
        if on:
            self._calib = []
            return None
        rec, self._calib = self._calib, None
        return [{k: float(v) for k, v in d.items()} for d in (rec or [])]

    # ---- frozen-weight images ---------------------------------------------------------------------------------
    def _key(self):
        k = 0
        for L in self.layers:
            for p in L.frozen():
                k += p._version + (p.data_ptr() & 0xFFFF)
        return (k, self.layers[0].proj_w.device)

    def full_mode(self) -> bool:
        """True when any base (non-adapter) weight of the stack is trainable: model_config.disable_lora.
        The forward then keeps the extra GEMM inputs the weight gradients need and the backward runs down to layer 0."""
        return any(p.requires_grad for L in self.layers for p in L.frozen())

    def base_params(self):
        return [p for L in self.layers for p in L.frozen()]

    def refresh(self):
        key = self._key()
        if key == self._cache_key and not self.full_mode():  # the fused optimizer updates weights in place (no version bump)
            return
        self._cache = []
        with torch.no_grad():
            for L in self.layers:
                c = _LayerCache()
                wqkv = _f32c(L.qkv_w[0]) if len(L.qkv_w) == 1 else torch.cat([_f32c(w) for w in L.qkv_w], dim=0)
                c.wqkv, c.wqkv_t = ops.cast_bf16(wqkv), ops.cast_transpose_bf16(wqkv)
                c.bqkv = _f32c(L.qkv_b[0]).clone() if len(L.qkv_b) == 1 else torch.cat([_f32c(b) for b in L.qkv_b], dim=0)
                c.wo, c.wo_t, c.bo = ops.cast_bf16(_f32c(L.proj_w)), ops.cast_transpose_bf16(_f32c(L.proj_w)), _f32c(L.proj_b)
                c.w1, c.w1_t, c.b1 = ops.cast_bf16(_f32c(L.fc1_w)), ops.cast_transpose_bf16(_f32c(L.fc1_w)), _f32c(L.fc1_b)
                c.w2, c.w2_t, c.b2 = ops.cast_bf16(_f32c(L.fc2_w)), ops.cast_transpose_bf16(_f32c(L.fc2_w)), _f32c(L.fc2_b)
                c.g1, c.be1, c.g2, c.be2 = _f32c(L.ln1_w), _f32c(L.ln1_b), _f32c(L.ln2_w), _f32c(L.ln2_b)
                if self.fp8 is not None:
                    f8 = self.fp8[len(self._cache)]
                    if "qkv_in" in f8:   # (a site selection leaves the other sites' images unbuilt)
                        c.wqkv8, c.cs_qkv = ops.quantize_rows_fp8(wqkv.contiguous(), f8["qkv_in"])
                        c.wo8, c.cs_o = ops.quantize_rows_fp8(_f32c(L.proj_w), f8["proj_in"])
                    c.w18, c.cs_1 = ops.quantize_rows_fp8(_f32c(L.fc1_w), f8["fc1_in"])
                    c.w28, c.cs_2 = ops.quantize_rows_fp8(_f32c(L.fc2_w), f8["fc2_in"])
                c.w1g = c.s1 = c.b1f = None
                if self.pre_ln and self.numerics["ln_fold"] == "on" and not self.full_mode():   # operand image of the norm2 -> fc1 fold, once per weight version
                    c.w1g, c.s1, c.b1f = ops.ln_fold_weights(_f32c(L.fc1_w), c.g2, c.be2, c.b1)
                c.v_fwd = c.v_bwd = c.a_cat = c.w_dt = c.slot2 = None
                c.wo_t8 = c.cs_ot = c.w1_t8 = c.cs_1t = c.w2_t8 = c.cs_2t = c.c2 = None
                self._cache.append(c)
            if self._dgrad8_ok():   # e4m3 images of the transposed weights, once per weight version
                # Full fine-tune (round 6): the weights move every step, so the images are re-quantised every step with the bf16 images they
                # are made from.  The per-layer constant c2 needs the l1 norm on the HOST (a kernel argument): it is re-derived every
                # DGRAD8_C2_EVERY refreshes only, and carries one binade of headroom (c2 / 2) for the growth of l1max in between.
                dev = self._cache[0].w2_t.device
                full = self.full_mode()
                reuse = full and self._c2 is not None and len(self._c2) == len(self._cache) and self._c2_age < self.DGRAD8_C2_EVERY
                l1 = None if reuse else torch.zeros((len(self._cache),), dtype=F32, device=dev)
                for i, c in enumerate(self._cache):
                    c.w2_t8, c.cs_2t = ops.quantize_rows_fp8_bf16(c.w2_t, 1.0, None if reuse else l1[i:i + 1])
                    c.wo_t8, c.cs_ot = ops.quantize_rows_fp8_bf16(c.wo_t, 1.0)
                if reuse:
                    self._c2_age += 1
                else:
                    # d(fc1 out) = e4m3(acc * col_scale * gelu' * c2): |acc * col_scale| <= 256 * l1max (scaled row maxima < 256), |gelu'| <= 1.13
                    self._c2 = [2.0 ** (math.floor(math.log2(448.0 / (256.0 * 1.13 * max(v, 1e-30)))) - (1 if full else 0))
                                for v in l1.cpu().tolist()]   # (one synchronisation per weight version; per DGRAD8_C2_EVERY steps under full fine-tune)
                    self._c2_age = 1
                for c, c2 in zip(self._cache, self._c2):
                    c.c2 = c2
                    c.w1_t8, c.cs_1t = ops.quantize_rows_fp8_bf16(c.w1_t, c.c2)   # its col_scale carries 1 / c2
        self._cache_key = key

    def pack_lora(self):
        """Per step: adapter parameters are trainable, so their bf16 operand images are rebuilt."""
        H = self.H
        for L, c in zip(self.layers, self._cache):
            if L.lora is None:
                continue
            dev = L.proj_w.device
            images = lambda: dict(v_fwd=torch.empty((3 * H, 8), dtype=BF16, device=dev), v_bwd=torch.empty((H, 8), dtype=BF16, device=dev),
                                  a_cat=torch.empty((8, H), dtype=BF16, device=dev), w_dt=torch.empty((16, 3 * H), dtype=BF16, device=dev))
            if c.v_fwd is None:
                im = images()
                c.v_fwd, c.v_bwd, c.a_cat, c.w_dt = im["v_fwd"], im["v_bwd"], im["a_cat"], im["w_dt"]
            slots = _rank_slots(L.lora)
            ops.lora_pack(*slots[0], c.v_fwd, c.v_bwd, c.a_cat, c.w_dt)
            if len(slots) > 1:      # ranks above 4: one more rank-(4+4) slot per four ranks (c.slot2: the list of their operand images)
                if self.fp8 is not None:
                    raise NotSupportedYet("fp8 forward with LoRA rank > 4")
                if c.slot2 is None or len(c.slot2) != len(slots) - 1:
                    c.slot2 = [images() for _ in slots[1:]]
                for sl, im in zip(slots[1:], c.slot2):
                    ops.lora_pack(*sl, im["v_fwd"], im["v_bwd"], im["a_cat"], im["w_dt"])
            else:
                c.slot2 = None

    # ---- further rank slots (r > 4): a pass each, built from the existing kernels — a zero-operand GEMM whose rank-8 MFMA step
    # carries u . V^T into an fp32 addend, which the layer's real GEMM then takes as its fp32 residual: ONE rounding of the sum,
    # as in the reference's B(A x) over all r columns
    def _slot2_addend(self, u, v, N, residual=None):
        M = u.shape[0]
        out = torch.empty((M, N), dtype=F32, device=u.device)
        ops.gemm_nt(torch.zeros((M, 64), dtype=BF16, device=u.device), torch.zeros((N, 64), dtype=BF16, device=u.device), rank_u=u, rank_v=v,
                    residual=residual, out_f32=out)
        return out

    def _slot2_fwd(self, c, x_bf16):
        """(fp32 [M,3H] addend sum_j (x A_j^T) B_j^T over the slots j >= 1 on the q and v columns, [t_j]): the slots chain through the
        addend (each zero-operand GEMM takes the previous sum as its fp32 residual), so the layer's GEMM still rounds the whole sum once."""
        add32, ts = None, []
        for im in c.slot2:
            tj = ops.lora_down_proj(x_bf16, im["a_cat"])
            add32 = self._slot2_addend(tj, im["v_fwd"], 3 * self.H, residual=add32)
            ts.append(tj)
        return add32, ts

    def _slot2_bwd_addend(self, c, dt2, residual=None):
        """fp32 [M,H] sum_j dt_j . A_cat_j over the slots j >= 1 (+ residual): joins the QKV dgrad as its fp32 addend."""
        add32 = residual
        for im, dtj in zip(c.slot2, dt2):
            add32 = self._slot2_addend(dtj, im["v_bwd"], self.H, residual=add32)
        return add32

    def lora_a(self, i: int):
        if i >= len(self.layers) or self.layers[i].lora is None:
            return None
        return self._cache[i].a_cat

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward(self, x_f32, x_bf16, t0, B: int, S: int, key_mask, save: bool, cls_only_last: bool = False, drop=None, full: bool = False,
                x_fp8=None):
        """x_f32 [M,H] residual stream entering layer 0.  Post-LN stacks also pass its bf16 image and the layer-0
        adapter down-projection t0 (both produced by the embedding LayerNorm).  Returns (x_f32, x_bf16, saved).
        cls_only_last (pre-LN only): the caller consumes token 0 only, so the LAST block evaluates attention for that
        query, and projection + MLP for that row, per sequence — the returned x_f32 is then [B,H] (the other rows of
        the last block are dead code the reference computes and discards).
        drop (post-LN only): (p_hidden, p_attention, base_seed) — HF BERT train-mode dropout; site seeds via ops.derive_seed.
        full: full fine-tune mode — every layer keeps its own attention output, MLP input and GELU output (the X operands of
        the weight gradients) instead of sharing temporaries.
        x_fp8 (post-LN, fp8-forward mode): the e4m3 image of x (scale fp8[0]["qkv_in"]) from the embedding LayerNorm."""
        H, FF, M = self.H, self.FF, B * S
        dev = x_f32.device
        saved = []
        new = lambda cols, dt: torch.empty((M, cols), dtype=dt, device=dev)
        keep = save and full
        f8s = self.fp8
        if f8s is not None and full:
            raise NotSupportedYet("fp8 forward with trainable base weights")
        cal = self._calib if f8s is None else None      # calibration pass: bf16 forward recording max |operand| per site
        amax = lambda t_: t_.abs().amax().float()
        f8 = None
        AT = ops.FP8 if f8s is not None else BF16    # dtype of the GEMM-operand temporaries
        mlp_only = f8s is not None and "qkv_in" not in f8s[0]   # fp8 site selection: QKV / attention / projection stay bf16
        o = None if keep else new(H, BF16 if mlp_only else AT)   # attention output (temporary, reused by every layer)
        a = None if keep else new(FF, AT)           # post-GELU activation (temporary)
        xn2 = new(H, AT) if (self.pre_ln and not keep) else None
        xn8 = new(H, ops.FP8) if (f8s is not None and not mlp_only) else None   # fp8 image of the first LayerNorm's output (temporary)
        # storage of gelu'(fc1 out): bf16 (default), one byte, or (round 6) the 12-bit e4m7 form of the bf16 value; the fp8-forward fc1
        # and the LN -> fc1 fold's consumer epilogue write bf16 only
        gg = self.numerics["gelu_grad"]
        if f8s is not None or (self.pre_ln and self.numerics["ln_fold"] == "on"):
            gg = "bf16"
        GG = BF16 if gg == "bf16" else torch.uint8
        HC = 3 * FF // 2 if gg == "e4m7" else FF                                                     # columns of that buffer
        act_save = {"bf16": ops.ACT_GELU_SAVE_GRAD, "u8": ops.ACT_GELU_SAVE_GRAD_U8, "e4m7": ops.ACT_GELU_SAVE_GRAD_E12}[gg]
        h_tmp = new(FF, BF16) if (f8s is not None and not save) else None       # the fp8 fc1 form always writes gelu'
        t = t0
        sp_ok = save and key_mask is None and f8s is None and S <= 224 and self.numerics["attn_bwd"] == "sp"
        # norm2 -> fc1 as the algebraic fold (numerics ln_fold): pre-LN, frozen base, bf16 forward, shapes the 256x256 kernel takes
        fold = (self.pre_ln and self.numerics["ln_fold"] == "on" and self._cache[0].w1g is not None and f8s is None and not full and GG == BF16 and H % 256 == 0 and FF % 256 == 0
                and M >= 1024 and ((M + 255) // 256) * (H // 256) >= 128 and H % 128 == 0)
        fold_sums = torch.empty((H // 128, M, 2), dtype=F32, device=dev) if fold else None
        if fold and h_tmp is None and not save:
            h_tmp = new(FF, BF16)   # the consumer epilogue always writes gelu' (eval forward: into a scratch buffer)
        for i, (L, c) in enumerate(zip(self.layers, self._cache)):
            has_lora = L.lora is not None
            rec = {}
            att_sv = None
            t2 = None   # further rank slots' down-projections (LoRA ranks above 4): a list
            f8 = f8s[i] if f8s is not None else None
            crec = {} if cal is not None else None
            if keep:
                o, a = new(H, BF16), new(FF, BF16)
                xn2 = new(H, BF16) if self.pre_ln else None
            if self.pre_ln and cls_only_last and i == len(self.layers) - 1:
                xn = new(H, BF16)
                st1 = torch.empty((M, 2), dtype=F32, device=dev)
                t = torch.empty((M, 8), dtype=BF16, device=dev) if has_lora else None
                qkv = new(3 * H, BF16)
                if f8 is not None and not mlp_only:   # the full-size GEMM of this block; its class-row remainder stays bf16
                    ops.layernorm_fwd(x_f32, c.g1, c.be1, self.eps, y_bf16=xn, stats=st1, lora_a=c.a_cat if has_lora else None, t_out=t,
                                      y_fp8=xn8, fp8_scale=f8["qkv_in"])
                    ops.gemm_fp8_nt(xn8, c.wqkv8, c.cs_qkv, bias=c.bqkv, rank_u=t, rank_v=c.v_fwd if has_lora else None, out_bf16=qkv)
                else:
                    ops.layernorm_fwd(x_f32, c.g1, c.be1, self.eps, y_bf16=xn, stats=st1, lora_a=c.a_cat if has_lora else None, t_out=t)
                    add32, t2 = self._slot2_fwd(c, xn) if c.slot2 is not None else (None, None)
                    ops.gemm_nt(xn, c.wqkv, bias=c.bqkv, rank_u=t, rank_v=c.v_fwd if has_lora else None, residual=add32, out_bf16=qkv)
                    if crec is not None:
                        crec["qkv_in"] = amax(xn)
                newB = lambda cols, dt: torch.empty((B, cols), dtype=dt, device=dev)
                o_cls = newB(H, BF16)
                ops.attention_fwd(qkv, B, S, self.heads, key_mask, o_cls, nq=1)
                x_cls = ops.gather_rows(x_f32.view(B, S, H))
                x1 = newB(H, F32)
                ops.gemm_nt(o_cls, c.wo, bias=c.bo, residual=x_cls, out_f32=x1)
                st2 = torch.empty((B, 2), dtype=F32, device=dev)
                xn2c = newB(H, BF16)
                ops.layernorm_fwd(x1, c.g2, c.be2, self.eps, y_bf16=xn2c, stats=st2)
                h = newB(HC, GG) if save else None
                ac = newB(FF, BF16)
                ops.gemm_nt(xn2c, c.w1, bias=c.b1, act=act_save if save else ops.ACT_GELU, out_pre=h, out_bf16=ac)
                x2 = newB(H, F32)
                ops.gemm_nt(ac, c.w2, bias=c.b2, residual=x1, out_f32=x2)
                if save:
                    rec = dict(x_in=x_f32, st1=st1, xn=xn, t=t, t2=t2, qkv=qkv, x1=x1, st2=st2, h=h, cls_only=True)
                    if keep:
                        rec.update(o=o_cls, xn2=xn2c, a=ac)
                x_f32 = x2
            elif self.pre_ln:
                # xn = LN1(x) (+ t = xn·A^T);  qkv = xn Wqkv^T + b + t·B^T
                xn = new(H, BF16)
                st1 = torch.empty((M, 2), dtype=F32, device=dev)
                t = torch.empty((M, 8), dtype=BF16, device=dev) if has_lora else None
                qkv = new(3 * H, BF16)
                x1 = new(H, F32)
                st2 = torch.empty((M, 2), dtype=F32, device=dev)
                h = new(HC, GG) if save else None        # holds gelu'(fc1 out): all the backward needs
                x2 = new(H, F32)
                if f8 is not None and mlp_only:   # fp8 on the MLP pair only: the attention half of the block is the bf16 path's
                    ops.layernorm_fwd(x_f32, c.g1, c.be1, self.eps, y_bf16=xn, stats=st1, lora_a=c.a_cat if has_lora else None, t_out=t)
                    add32, t2 = self._slot2_fwd(c, xn) if c.slot2 is not None else (None, None)
                    ops.gemm_nt(xn, c.wqkv, bias=c.bqkv, rank_u=t, rank_v=c.v_fwd if has_lora else None, residual=add32, out_bf16=qkv)
                    ops.attention_fwd(qkv, B, S, self.heads, key_mask, o)
                    ops.gemm_nt(o, c.wo, bias=c.bo, residual=x_f32, out_f32=x1)
                    ops.layernorm_fwd(x1, c.g2, c.be2, self.eps, stats=st2, y_fp8=xn2, fp8_scale=f8["fc1_in"])
                    ops.gemm_fp8_nt(xn2, c.w18, c.cs_1, bias=c.b1, gelu_out_fp8=a, gelu_out_scale=f8["fc2_in"], out_pre=h if save else h_tmp)
                    ops.gemm_fp8_nt(a, c.w28, c.cs_2, bias=c.b2, residual=x1, out_f32=x2)
                elif f8 is not None:
                    ops.layernorm_fwd(x_f32, c.g1, c.be1, self.eps, y_bf16=xn, stats=st1, lora_a=c.a_cat if has_lora else None, t_out=t,
                                      y_fp8=xn8, fp8_scale=f8["qkv_in"])
                    ops.gemm_fp8_nt(xn8, c.wqkv8, c.cs_qkv, bias=c.bqkv, rank_u=t, rank_v=c.v_fwd if has_lora else None, out_bf16=qkv)
                    ops.attention_fwd(qkv, B, S, self.heads, key_mask, o, out_fp8_scale=f8["proj_in"])
                    ops.gemm_fp8_nt(o, c.wo8, c.cs_o, bias=c.bo, residual=x_f32, out_f32=x1)
                    ops.layernorm_fwd(x1, c.g2, c.be2, self.eps, stats=st2, y_fp8=xn2, fp8_scale=f8["fc1_in"])
                    ops.gemm_fp8_nt(xn2, c.w18, c.cs_1, bias=c.b1, gelu_out_fp8=a, gelu_out_scale=f8["fc2_in"], out_pre=h if save else h_tmp)
                    ops.gemm_fp8_nt(a, c.w28, c.cs_2, bias=c.b2, residual=x1, out_f32=x2)
                else:
                    ops.layernorm_fwd(x_f32, c.g1, c.be1, self.eps, y_bf16=xn, stats=st1, lora_a=c.a_cat if has_lora else None, t_out=t)
                    add32, t2 = self._slot2_fwd(c, xn) if c.slot2 is not None else (None, None)
                    ops.gemm_nt(xn, c.wqkv, bias=c.bqkv, rank_u=t, rank_v=c.v_fwd if has_lora else None, residual=add32, out_bf16=qkv)
                    if sp_ok:   # training forward of the single-pass attention backward: this layer's o, its rounding residual, the lse
                        o = o if keep else new(H, BF16)
                        att_sv = dict(o_att=o, o_lo=new(H, BF16), lse=torch.empty((B * self.heads * S,), dtype=F32, device=dev))
                        ops.attention_fwd(qkv, B, S, self.heads, None, o, lse=att_sv["lse"], o_lo=att_sv["o_lo"])
                    else:
                        ops.attention_fwd(qkv, B, S, self.heads, key_mask, o)
                    if fold and cal is None:
                        # norm2 never makes a pass of its own: the projection writes x1, bf16(x1) (into the xn2 temporary) and its row
                        # sums; (mean, rstd) -> st2 (what the LayerNorm backward reads); fc1 applies them to (x1b . (gamma o W1)^T)
                        ops.gemm_nt(o, c.wo, bias=c.bo, residual=x_f32, out_f32=x1, out_bf16=xn2, row_sums=fold_sums)
                        ops.rowsum_finalize(fold_sums, self.eps, st2)
                        ops.gemm_nt(xn2, c.w1g, bias=c.b1f, act=ops.ACT_GELU_SAVE_GRAD, out_pre=h if save else h_tmp, out_bf16=a, row_stats=st2, col_sum_w=c.s1)
                    else:
                        ops.gemm_nt(o, c.wo, bias=c.bo, residual=x_f32, out_f32=x1)
                        ops.layernorm_fwd(x1, c.g2, c.be2, self.eps, y_bf16=xn2, stats=st2)
                        ops.gemm_nt(xn2, c.w1, bias=c.b1, act=act_save if save else ops.ACT_GELU, out_pre=h if save else None, out_bf16=a)
                    ops.gemm_nt(a, c.w2, bias=c.b2, residual=x1, out_f32=x2)
                    if crec is not None:
                        crec.update(qkv_in=amax(xn), proj_in=amax(o), fc1_in=amax(xn2), fc2_in=amax(a))
                if save:
                    rec = dict(x_in=x_f32, st1=st1, xn=xn, t=t, t2=t2, qkv=qkv, x1=x1, st2=st2, h=h)
                    if keep:
                        rec.update(o=o, xn2=xn2, a=a)
                    if att_sv is not None:
                        rec.update(att_sv)
                x_f32 = x2
            else:
                d_att = d_h1 = d_h2 = None
                if drop is not None:
                    p_h, p_a, base = drop
                    d_att = ops.Drop(p_a, ops.derive_seed(base, i, 0))
                    d_h1 = ops.Drop(p_h, ops.derive_seed(base, i, 1))
                    d_h2 = ops.Drop(p_h, ops.derive_seed(base, i, 2))
                qkv = new(3 * H, BF16)
                s1 = new(H, F32)
                x1_f32 = new(H, F32)
                st1 = torch.empty((M, 2), dtype=F32, device=dev)
                h = new(HC, GG) if save else None
                s2 = new(H, F32)
                x2_f32, x2_bf16 = new(H, F32), new(H, BF16)
                st2 = torch.empty((M, 2), dtype=F32, device=dev)
                nxt = self.lora_a(i + 1)
                t_next = torch.empty((M, 8), dtype=BF16, device=dev) if nxt is not None else None
                ru, rv = (t if has_lora else None), (c.v_fwd if has_lora else None)
                if f8 is not None and mlp_only:   # fp8 on the MLP pair only (round 5): the attention half is the bf16 path's, LayerNorm 1 feeds fc1 as e4m3
                    x1_bf16 = None
                    x18 = new(H, ops.FP8)
                    ops.gemm_nt(x_bf16, c.wqkv, bias=c.bqkv, rank_u=ru, rank_v=rv, out_bf16=qkv)
                    ops.attention_fwd(qkv, B, S, self.heads, key_mask, o, drop=d_att)
                    ops.gemm_nt(o, c.wo, bias=c.bo, residual=x_f32, out_f32=s1, drop=d_h1)
                    ops.layernorm_fwd(s1, c.g1, c.be1, self.eps, y_f32=x1_f32, stats=st1, y_fp8=x18, fp8_scale=f8["fc1_in"])
                    ops.gemm_fp8_nt(x18, c.w18, c.cs_1, bias=c.b1, gelu_out_fp8=a, gelu_out_scale=f8["fc2_in"], out_pre=h if save else h_tmp)
                    ops.gemm_fp8_nt(a, c.w28, c.cs_2, bias=c.b2, residual=x1_f32, out_f32=s2, drop=d_h2)
                    ops.layernorm_fwd(s2, c.g2, c.be2, self.eps, y_bf16=x2_bf16, y_f32=x2_f32, stats=st2, lora_a=nxt, t_out=t_next)
                elif f8 is not None:
                    if x_fp8 is None:
                        raise ValueError("fp8 forward (post-LN): the caller passes the e4m3 image of x")
                    x1_bf16 = None
                    x18 = xn8
                    ops.gemm_fp8_nt(x_fp8, c.wqkv8, c.cs_qkv, bias=c.bqkv, rank_u=ru, rank_v=rv, out_bf16=qkv)
                    ops.attention_fwd(qkv, B, S, self.heads, key_mask, o, drop=d_att, out_fp8_scale=f8["proj_in"])
                    ops.gemm_fp8_nt(o, c.wo8, c.cs_o, bias=c.bo, residual=x_f32, out_f32=s1, drop=d_h1)
                    ops.layernorm_fwd(s1, c.g1, c.be1, self.eps, y_f32=x1_f32, stats=st1, y_fp8=x18, fp8_scale=f8["fc1_in"])
                    ops.gemm_fp8_nt(x18, c.w18, c.cs_1, bias=c.b1, gelu_out_fp8=a, gelu_out_scale=f8["fc2_in"], out_pre=h if save else h_tmp)
                    ops.gemm_fp8_nt(a, c.w28, c.cs_2, bias=c.b2, residual=x1_f32, out_f32=s2, drop=d_h2)
                    x8_next = new(H, ops.FP8)
                    ops.layernorm_fwd(s2, c.g2, c.be2, self.eps, y_bf16=x2_bf16, y_f32=x2_f32, stats=st2, lora_a=nxt, t_out=t_next,
                                      y_fp8=x8_next, fp8_scale=f8s[min(i + 1, len(f8s) - 1)]["qkv_in"])
                    x_fp8 = x8_next
                else:
                    x1_bf16 = new(H, BF16)
                    add32, t2 = self._slot2_fwd(c, x_bf16) if c.slot2 is not None else (None, None)
                    ops.gemm_nt(x_bf16, c.wqkv, bias=c.bqkv, rank_u=ru, rank_v=rv, residual=add32, out_bf16=qkv)
                    if sp_ok:
                        o = o if keep else new(H, BF16)
                        att_sv = dict(o_att=o, o_lo=new(H, BF16), lse=torch.empty((B * self.heads * S,), dtype=F32, device=dev))
                        ops.attention_fwd(qkv, B, S, self.heads, None, o, drop=d_att, lse=att_sv["lse"], o_lo=att_sv["o_lo"])
                    else:
                        ops.attention_fwd(qkv, B, S, self.heads, key_mask, o, drop=d_att)
                    ops.gemm_nt(o, c.wo, bias=c.bo, residual=x_f32, out_f32=s1, drop=d_h1)
                    ops.layernorm_fwd(s1, c.g1, c.be1, self.eps, y_bf16=x1_bf16, y_f32=x1_f32, stats=st1)
                    ops.gemm_nt(x1_bf16, c.w1, bias=c.b1, act=act_save if save else ops.ACT_GELU, out_pre=h if save else None, out_bf16=a)
                    ops.gemm_nt(a, c.w2, bias=c.b2, residual=x1_f32, out_f32=s2, drop=d_h2)
                    ops.layernorm_fwd(s2, c.g2, c.be2, self.eps, y_bf16=x2_bf16, y_f32=x2_f32, stats=st2, lora_a=nxt, t_out=t_next)
                    if crec is not None:
                        crec.update(qkv_in=amax(x_bf16), proj_in=amax(o), fc1_in=amax(x1_bf16), fc2_in=amax(a))
                if save:
                    rec = dict(x_bf16=x_bf16, t=t if has_lora else None, t2=t2, qkv=qkv, s1=s1, st1=st1, h=h, s2=s2, st2=st2,
                               d_att=d_att, d_h1=d_h1, d_h2=d_h2)
                    if keep:
                        rec.update(o=o, x1_bf16=x1_bf16, a=a)
                    if att_sv is not None:
                        rec.update(att_sv)
                x_f32, x_bf16, t = x2_f32, x2_bf16, t_next
            saved.append(rec)
            if cal is not None:
                cal.append(crec)
        return x_f32, x_bf16, saved

    # ---- backward -----------------------------------------------------------------------------------------------
    def layer_params(self, i: int):
        """Every parameter of layer i a gradient can be produced for (base weights + adapters)."""
        L = self.layers[i]
        return L.frozen() + (L.lora.tensors() if L.lora is not None else [])

    def backward(self, dx_f32, dx_bf16, saved, B: int, S: int, key_mask, grads: dict, full: bool = False, on_layer_done=None):
        """dx = gradient w.r.t. the stack output (fp32 residual stream; pre-LN also needs its bf16 image).
        on_layer_done(i): called once layer i's parameter gradients are complete (enqueued on the current stream) — the
        trainer's hook for starting that layer's gradient all-reduce under the rest of the backward.
        Fills grads[id(param)] for the adapters — and, in full fine-tune mode, for every base weight / bias / LayerNorm
        parameter present in `grads`; returns the fp32 gradient w.r.t. the stack input then (None in LoRA mode, where the
        walk stops at the lowest adapted layer because nothing below it is trainable)."""
        H, FF, M = self.H, self.FF, B * S
        dev = dx_f32.device
        new = lambda cols, dt: torch.empty((M, cols), dtype=dt, device=dev)
        dh = None   # d(fc1 out), bf16 [M,FF]: allocated by the first layer that takes the bf16 dgrad
        dtmp = new(H, BF16)
        dqkv = new(3 * H, BF16)
        dt = torch.empty((M, 16), dtype=BF16, device=dev)
        first_lora = min((i for i, L in enumerate(self.layers) if L.lora is not None), default=len(self.layers))
        if full:
            first_lora = -1  # every layer has trainable parameters and the input gradient is needed
        # bf16 residual-gradient stream (see NUMERICS_CHOICES).  Round 4: full fine-tune mode takes it too — the LayerNorm parameter
        # gradients ride along in the same kernel (clibd_layernorm_bwd_any), and the BOTTOM layer hands an fp32 gradient to the
        # embedding backward as before (`need32`).  409.6 ms per step at b = 2048 with the fp32 stream (profiles/r04_fullft_*_v1*).
        r16 = self.numerics["residual_grad"] == "bf16"
        # 8-bit dgrad (numerics dgrad = "fp8"): every LayerNorm backward below also writes its output as e4m3 rows + one dequantisation
        # factor per row (new8), which the next dgrad GEMM takes as its A operand; dx8 = that pair for the incoming stream gradient
        # Round 6: also with trainable base weights (full fine-tune, the reference's final recipe: `disable_lora: true`) — the dgrad takes the e4m3
        # rows, every weight gradient stays bf16 (the LayerNorm backward then writes both copies, and the fc2 dgrad writes d(fc1 out) twice: e4m3
        # for the fc1 dgrad, bf16 for the fc1 weight gradient).
        dg8 = self._dgrad8_ok() and self._cache[0].w2_t8 is not None and M % 4 == 0 and M * FF < 2 ** 32   # (the fp8 kernel addresses operands with 32-bit byte offsets)
        new8 = lambda cols: (torch.empty((M, cols), dtype=torch.uint8, device=dev).view(ops.FP8), torch.empty((M,), dtype=F32, device=dev))
        f8kw = lambda pair: dict(dx_fp8=pair[0], row_dequant=pair[1])
        dx8 = None
        # full fine-tune: the LayerNorm backward accumulates d(gamma), d(beta) in the same pass (it holds dy and xhat anyway)
        pg = lambda w, b: (dict(dgamma=grads[id(w)].view(-1), dbeta=grads[id(b)].view(-1)) if full and id(w) in grads else {})
        wg = lambda dy, x, ws, bs: linear_wgrad(dy, x, ws, bs, grads) if full else None
        for i in range(len(self.layers) - 1, -1, -1):
            L, c, rec = self.layers[i], self._cache[i], saved[i]
            has_lora = L.lora is not None
            dt2 = None   # further rank slots' dt (LoRA ranks above 4): a list, set by _lora_grads
            if i < first_lora:
                break  # nothing trainable at or below this layer
            if self.pre_ln and rec.get("cls_only"):
                # dx_f32 / dx_bf16 are [B,H]: the gradient of the class-token row of this block's output
                newB = lambda cols, dt: torch.empty((B, cols), dtype=dt, device=dev)
                dhc, dtc = newB(FF, BF16), newB(H, BF16)
                wg(dx_bf16, rec.get("a"), [L.fc2_w], [L.fc2_b])
                ops.gemm_nt(dx_bf16, c.w2_t, act=_mul_aux_act(rec["h"], FF), aux=rec["h"], out_bf16=dhc)
                wg(dhc, rec.get("xn2"), [L.fc1_w], [L.fc1_b])
                ops.gemm_nt(dhc, c.w1_t, out_bf16=dtc)
                dx1_f32, dx1_bf16 = newB(H, F32), newB(H, BF16)
                ops.layernorm_bwd(dtc, rec["x1"], rec["st2"], c.g2, dres=dx_f32, dx_f32=dx1_f32, dx_bf16=dx1_bf16, **pg(L.ln2_w, L.ln2_b))
                wg(dx1_bf16, rec.get("o"), [L.proj_w], [L.proj_b])
                ops.gemm_nt(dx1_bf16, c.wo_t, out_bf16=dtc)                                              # d(attn out), class rows
                ops.attention_bwd(rec["qkv"], dtc, B, S, self.heads, key_mask, dqkv, nq=1)
                if has_lora:
                    dt2 = self._lora_grads(L, c, dqkv, rec["xn"], rec["t"], dt, grads, rec.get("t2"))
                wg(dqkv, rec["xn"], L.qkv_w, L.qkv_b)
                if i > first_lora:
                    add32 = None if dt2 is None else self._slot2_bwd_addend(c, dt2)      # further rank slots: sum_j dt_j . A_cat_j, fp32
                    ops.gemm_nt(dqkv, c.wqkv_t, rank_u=dt if has_lora else None, rank_v=c.v_bwd if has_lora else None, residual=add32, out_bf16=dtmp)
                    if r16:
                        dres16, _ = ops.scatter_rows(dx1_f32, S, bf16=True, f32=False)           # residual path: class rows only
                        ndx_bf16 = new(H, BF16)
                        ndx_f32 = new(H, F32) if (full and i == 0) else None
                        dx8 = new8(H) if (dg8 and "mlp" in self.dgrad8_sites) else None
                        ops.layernorm_bwd(dtmp, rec["x_in"], rec["st1"], c.g1, dres_bf16=dres16, dx_bf16=ndx_bf16, dx_f32=ndx_f32, **pg(L.ln1_w, L.ln1_b),
                                          **(f8kw(dx8) if dx8 is not None else {}))
                        dx_f32, dx_bf16 = ndx_f32, ndx_bf16
                    else:
                        _, dres_full = ops.scatter_rows(dx1_f32, S, bf16=False, f32=True)        # residual path: class rows only
                        ndx_f32, ndx_bf16 = new(H, F32), new(H, BF16)
                        ops.layernorm_bwd(dtmp, rec["x_in"], rec["st1"], c.g1, dres=dres_full, dx_f32=ndx_f32, dx_bf16=ndx_bf16, **pg(L.ln1_w, L.ln1_b))
                        dx_f32, dx_bf16 = ndx_f32, ndx_bf16
            elif self.pre_ln:
                wg(dx_bf16, rec.get("a"), [L.fc2_w], [L.fc2_b])
                if dg8 and dx8 is not None:   # d(fc1 out) leaves as e4m3 with the rows' scales x c2; the fc1 dgrad divides both back out
                    dh8 = torch.empty((M, FF), dtype=torch.uint8, device=dev).view(ops.FP8)
                    if full:   # ... and once more as bf16: the operand of fc1's weight gradient
                        dh = new(FF, BF16) if dh is None else dh
                    ops.gemm_fp8_dgrad_nt(dx8[0], c.w2_t8, c.cs_2t, aux=rec["h"], act=_mul_aux_act(rec["h"], FF), out_fp8=dh8, out_fp8_scale=c.c2,
                                          a_row_dequant=dx8[1] if full else None, out_bf16_dual=dh if full else None)
                    wg(dh, rec.get("xn2"), [L.fc1_w], [L.fc1_b])
                    ops.gemm_fp8_dgrad_nt(dh8, c.w1_t8, c.cs_1t, a_row_dequant=dx8[1], out_bf16=dtmp)
                else:
                    dh = new(FF, BF16) if dh is None else dh
                    ops.gemm_nt(dx_bf16, c.w2_t, act=_mul_aux_act(rec["h"], FF), aux=rec["h"], out_bf16=dh)      # d(fc1 out)
                    wg(dh, rec.get("xn2"), [L.fc1_w], [L.fc1_b])
                    ops.gemm_nt(dh, c.w1_t, out_bf16=dtmp)                                               # d(LN2 out)
                dx18 = new8(H) if (dg8 and "proj" in self.dgrad8_sites) else None
                if r16:
                    dx1_f32, dx1_bf16 = None, new(H, BF16)
                    ops.layernorm_bwd(dtmp, rec["x1"], rec["st2"], c.g2, dres_bf16=dx_bf16, dx_bf16=dx1_bf16, **pg(L.ln2_w, L.ln2_b),
                                      **(f8kw(dx18) if dx18 is not None else {}))
                else:
                    dx1_f32, dx1_bf16 = new(H, F32), new(H, BF16)
                    ops.layernorm_bwd(dtmp, rec["x1"], rec["st2"], c.g2, dres=dx_f32, dx_f32=dx1_f32, dx_bf16=dx1_bf16, **pg(L.ln2_w, L.ln2_b))
                wg(dx1_bf16, rec.get("o"), [L.proj_w], [L.proj_b])
                if dx18 is not None:
                    ops.gemm_fp8_dgrad_nt(dx18[0], c.wo_t8, c.cs_ot, a_row_dequant=dx18[1], out_bf16=dtmp)   # d(attn out)
                else:
                    ops.gemm_nt(dx1_bf16, c.wo_t, out_bf16=dtmp)                                         # d(attn out)
                self._attention_bwd(rec, dtmp, B, S, key_mask, dqkv)
                if has_lora:
                    dt2 = self._lora_grads(L, c, dqkv, rec["xn"], rec["t"], dt, grads, rec.get("t2"))
                wg(dqkv, rec["xn"], L.qkv_w, L.qkv_b)
                if i > first_lora:
                    add32 = None if dt2 is None else self._slot2_bwd_addend(c, dt2)      # further rank slots: sum_j dt_j . A_cat_j, fp32
                    ops.gemm_nt(dqkv, c.wqkv_t, rank_u=dt if has_lora else None, rank_v=c.v_bwd if has_lora else None, residual=add32, out_bf16=dtmp)
                    if r16:
                        ndx_f32, ndx_bf16 = (new(H, F32) if (full and i == 0) else None), new(H, BF16)
                        dx8 = new8(H) if (dg8 and "mlp" in self.dgrad8_sites) else None
                        ops.layernorm_bwd(dtmp, rec["x_in"], rec["st1"], c.g1, dres_bf16=dx1_bf16, dx_bf16=ndx_bf16, dx_f32=ndx_f32, **pg(L.ln1_w, L.ln1_b),
                                          **(f8kw(dx8) if dx8 is not None else {}))
                    else:
                        ndx_f32, ndx_bf16 = new(H, F32), new(H, BF16)
                        ops.layernorm_bwd(dtmp, rec["x_in"], rec["st1"], c.g1, dres=dx1_f32, dx_f32=ndx_f32, dx_bf16=ndx_bf16, **pg(L.ln1_w, L.ln1_b))
                    dx_f32, dx_bf16 = ndx_f32, ndx_bf16
            elif r16:
                # post-LN, bf16 stream: `dx_f32` holds the incoming gradient of the layer output (fp32 from the head at the top layer,
                # bf16 below).  Each LayerNorm backward writes the un-dropped residual copy (*_res) and, under dropout, the masked
                # copy the dense branch's dgrad consumes; the two dgrads that re-join the stream add the residual copy in their epilogue.
                def ln_back(dy, xs, st, gam, drop_site, pgk):
                    res = new(H, BF16)
                    if dg8 and full:   # the dgrad takes the e4m3 rows, the weight gradient the (masked) bf16 copy; parameter gradients ride along
                        q8 = new8(H)
                        dropped = drop_site is not None and drop_site.thr16 > 0
                        masked = new(H, BF16) if dropped else None
                        ops.layernorm_bwd(dy, xs, st, gam, dx_res_bf16=res, dx_bf16=masked, drop=drop_site, **pgk, **f8kw(q8))
                        return res, (masked if dropped else res), q8
                    if dg8:   # the dense branch's dgrad takes the e4m3 rows: the masked bf16 copy is not written at all
                        q8 = new8(H)
                        ops.layernorm_bwd(dy, xs, st, gam, dx_res_bf16=res, drop=drop_site, **f8kw(q8))
                        return res, None, q8
                    if drop_site is not None and drop_site.thr16 > 0:
                        masked = new(H, BF16)
                        ops.layernorm_bwd(dy, xs, st, gam, dx_res_bf16=res, dx_bf16=masked, drop=drop_site, **pgk)
                        return res, masked, None
                    ops.layernorm_bwd(dy, xs, st, gam, dx_res_bf16=res, **pgk)
                    return res, res, None

                ds2_res, ds2_b, ds2_8 = ln_back(dx_f32, rec["s2"], rec["st2"], c.g2, rec["d_h2"], pg(L.ln2_w, L.ln2_b))
                dx1 = new(H, BF16)
                if dg8:
                    dh8 = torch.empty((M, FF), dtype=torch.uint8, device=dev).view(ops.FP8)
                    if full:
                        wg(ds2_b, rec.get("a"), [L.fc2_w], [L.fc2_b])
                        dh = new(FF, BF16) if dh is None else dh
                    ops.gemm_fp8_dgrad_nt(ds2_8[0], c.w2_t8, c.cs_2t, aux=rec["h"], act=_mul_aux_act(rec["h"], FF), out_fp8=dh8, out_fp8_scale=c.c2,
                                          a_row_dequant=ds2_8[1] if full else None, out_bf16_dual=dh if full else None)
                    wg(dh, rec.get("x1_bf16"), [L.fc1_w], [L.fc1_b])
                    ops.gemm_fp8_dgrad_nt(dh8, c.w1_t8, c.cs_1t, a_row_dequant=ds2_8[1], aux=ds2_res, act=ops.ACT_ADD_AUX, out_bf16=dx1)
                else:
                    dh = new(FF, BF16) if dh is None else dh
                    wg(ds2_b, rec.get("a"), [L.fc2_w], [L.fc2_b])
                    ops.gemm_nt(ds2_b, c.w2_t, act=_mul_aux_act(rec["h"], FF), aux=rec["h"], out_bf16=dh)
                    wg(dh, rec.get("x1_bf16"), [L.fc1_w], [L.fc1_b])
                    ops.gemm_nt(dh, c.w1_t, act=ops.ACT_ADD_AUX, aux=ds2_res, out_bf16=dx1)
                ds1_res, ds1_b, ds1_8 = ln_back(dx1, rec["s1"], rec["st1"], c.g1, rec["d_h1"], pg(L.ln1_w, L.ln1_b))
                if dg8:
                    wg(ds1_b, rec.get("o"), [L.proj_w], [L.proj_b])
                    ops.gemm_fp8_dgrad_nt(ds1_8[0], c.wo_t8, c.cs_ot, a_row_dequant=ds1_8[1], out_bf16=dtmp)
                else:
                    wg(ds1_b, rec.get("o"), [L.proj_w], [L.proj_b])
                    ops.gemm_nt(ds1_b, c.wo_t, out_bf16=dtmp)
                self._attention_bwd(rec, dtmp, B, S, key_mask, dqkv, drop=rec["d_att"])
                if has_lora:
                    dt2 = self._lora_grads(L, c, dqkv, rec["x_bf16"], rec["t"], dt, grads, rec.get("t2"))
                wg(dqkv, rec["x_bf16"], L.qkv_w, L.qkv_b)
                if i > first_lora:
                    add32 = None if dt2 is None else self._slot2_bwd_addend(c, dt2)
                    if full and i == 0:   # the embedding backward takes an fp32 gradient (generic epilogue: one launch per tower)
                        ndx = new(H, F32)
                        ops.gemm_nt(dqkv, c.wqkv_t, rank_u=dt if has_lora else None, rank_v=c.v_bwd if has_lora else None,
                                    act=ops.ACT_ADD_AUX, aux=ds1_res, residual=add32, out_f32=ndx)
                    else:
                        ndx = new(H, BF16)
                        ops.gemm_nt(dqkv, c.wqkv_t, rank_u=dt if has_lora else None, rank_v=c.v_bwd if has_lora else None,
                                    act=ops.ACT_ADD_AUX, aux=ds1_res, residual=add32, out_bf16=ndx)
                    dx_f32 = ndx
            else:
                ds2_f32, ds2_bf16 = new(H, F32), new(H, BF16)
                ops.layernorm_bwd(dx_f32, rec["s2"], rec["st2"], c.g2, dx_f32=ds2_f32, dx_bf16=ds2_bf16, drop=rec["d_h2"], **pg(L.ln2_w, L.ln2_b))
                wg(ds2_bf16, rec.get("a"), [L.fc2_w], [L.fc2_b])
                dh = new(FF, BF16) if dh is None else dh
                ops.gemm_nt(ds2_bf16, c.w2_t, act=_mul_aux_act(rec["h"], FF), aux=rec["h"], out_bf16=dh)
                wg(dh, rec.get("x1_bf16"), [L.fc1_w], [L.fc1_b])
                dx1 = new(H, F32)
                ops.gemm_nt(dh, c.w1_t, residual=ds2_f32, out_f32=dx1)
                ds1_f32, ds1_bf16 = new(H, F32), new(H, BF16)
                ops.layernorm_bwd(dx1, rec["s1"], rec["st1"], c.g1, dx_f32=ds1_f32, dx_bf16=ds1_bf16, drop=rec["d_h1"], **pg(L.ln1_w, L.ln1_b))
                wg(ds1_bf16, rec.get("o"), [L.proj_w], [L.proj_b])
                ops.gemm_nt(ds1_bf16, c.wo_t, out_bf16=dtmp)
                self._attention_bwd(rec, dtmp, B, S, key_mask, dqkv, drop=rec["d_att"])
                if has_lora:
                    dt2 = self._lora_grads(L, c, dqkv, rec["x_bf16"], rec["t"], dt, grads, rec.get("t2"))
                wg(dqkv, rec["x_bf16"], L.qkv_w, L.qkv_b)
                if i > first_lora:
                    ndx = new(H, F32)
                    res32 = ds1_f32 if dt2 is None else self._slot2_bwd_addend(c, dt2, residual=ds1_f32)
                    ops.gemm_nt(dqkv, c.wqkv_t, rank_u=dt if has_lora else None, rank_v=c.v_bwd if has_lora else None,
                                residual=res32, out_f32=ndx)
                    dx_f32 = ndx
            if on_layer_done is not None:
                on_layer_done(i)
        return dx_f32 if full else None

    def _attention_bwd(self, rec, dout, B, S, key_mask, dqkv, drop=None):
        if "lse" in rec:   # the forward saved what the single-pass kernel needs
            ops.attention_bwd_sp(rec["qkv"], dout, rec["o_att"], rec["o_lo"], rec["lse"], B, S, self.heads, dqkv, drop=drop)
        else:
            ops.attention_bwd(rec["qkv"], dout, B, S, self.heads, key_mask, dqkv, drop=drop)

    def _lora_grads(self, L, c, dqkv, x_bf16, t, dt, grads, t2=None):
        """dt[:, 0:4] = dq·B_q, dt[:, 4:8] = dv·B_v (the rank-8 operand of the QKV dgrad that follows) and the four adapter gradients
        in one call: dq and dv are streamed once for dt and dB together (clibd_lora_backward); the k segment of dqkv is never read.
        Ranks other than 4 run on rank-4 slots (_rank_slots): scratch gradients, sliced back; ranks above 4 take one more call per slot
        with that slot's images and down-projection t2[j], whose dt (returned as a list) join the QKV dgrad as an fp32 addend (_slot2_bwd_addend)."""
        H = self.H
        lp = L.lora
        r = lp.a_q.shape[0]
        if r == 4:
            ops.lora_backward(dqkv, x_bf16, t, c.w_dt, dt, grads[id(lp.a_q)], grads[id(lp.a_v)], grads[id(lp.b_q)], grads[id(lp.b_v)])
            return None
        dev = dqkv.device
        dt2 = []
        nslots = (r + 3) // 4
        for k in range(nslots):
            lo = 4 * k
            if k == 0:
                tk, w_dt, dtk = t, c.w_dt, dt
            else:
                tk, w_dt, dtk = t2[k - 1], c.slot2[k - 1]["w_dt"], torch.empty_like(dt)
                dt2.append(dtk)
            ga_q, ga_v = torch.zeros((4, H), dtype=F32, device=dev), torch.zeros((4, H), dtype=F32, device=dev)
            gb_q, gb_v = torch.zeros((H, 4), dtype=F32, device=dev), torch.zeros((H, 4), dtype=F32, device=dev)
            ops.lora_backward(dqkv, x_bf16, tk, w_dt, dtk, ga_q, ga_v, gb_q, gb_v)
            n = min(4, r - lo)   # the padded rows / columns receive exact zeros' worth of signal
            grads[id(lp.a_q)][lo:lo + n].add_(ga_q[:n]); grads[id(lp.a_v)][lo:lo + n].add_(ga_v[:n])
            grads[id(lp.b_q)][:, lo:lo + n].add_(gb_q[:, :n]); grads[id(lp.b_v)][:, lo:lo + n].add_(gb_v[:, :n])
        dt2 = dt2 or None
        return dt2


def linear_wgrad(dy_bf16: torch.Tensor, x_bf16: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], grads: dict):
    """Weight / bias gradients of y = x W^T + b for the parameters present in `grads` (accumulating):
    dW [N,K] += dy^T x — read in place by the rows-contracting kernel when the shape allows (M % 128 == 0, N and K % 256 == 0:
    every full-size layer), else as the NT GEMM (dy^T [N,Mp]) (x^T [K,Mp])^T over zero-padded transposes —
    db += column sums of dy.  `weights` may be the row-wise pieces of a fused projection (BERT query / key / value)."""
    if not any(id(w) in grads for w in weights) and not any(id(b) in grads for b in biases):
        return
    if x_bf16 is None:
        raise RuntimeError("full fine-tune backward needs the layer's GEMM input (forward ran without full=True)")
    M = dy_bf16.shape[0]
    want_b = [b is not None and id(b) in grads for b in biases]
    bias_done = False
    K = x_bf16.shape[1]
    tn_ok = (M % 128 == 0 and M >= 256 and K % 256 == 0 and dy_bf16.stride(0) % 8 == 0 and x_bf16.stride(0) % 8 == 0
             and all(w.shape[0] % 256 == 0 for w in weights))
    pending = [id(w) in grads for w in weights]   # weight gradients still to be produced
    if tn_ok and any(pending):
        # in-place rows-contracting GEMM (gemm256_tn.hip): no dy^T / x^T; the bias gradient (column sums of dy) rides along.
        # A piece the kernel declines (ops.gemm_tn_splitk returns False) stays pending and takes the transpose path below.
        n0 = 0
        done = []
        for j, (w, b, wb) in enumerate(zip(weights, biases, want_b)):
            n1 = n0 + w.shape[0]
            took = False
            if pending[j]:
                cs = grads[id(b)].view(-1) if wb else None
                took = ops.gemm_tn_splitk(dy_bf16[:, n0:n1], x_bf16, grads[id(w)].view(w.shape[0], -1), accumulate=True, colsum=cs)
                pending[j] = not took
            done.append(took and wb)
            n0 = n1
        want_b = [wb and not d for wb, d in zip(want_b, done)]   # biases whose weight is frozen still take the column-sum kernel
    if any(pending):
        N = dy_bf16.shape[1]
        csum, direct = None, False
        if any(want_b) and N % 8 == 0 and dy_bf16.stride(0) % 8 == 0:
            if len(weights) == 1:
                csum, direct = grads[id(biases[0])].view(-1), True        # one bias over all N columns: accumulate in place
            else:
                csum = torch.zeros((N,), dtype=F32, device=dy_bf16.device)   # column sums of dy in the same pass as its transpose
        dyT = ops.transpose_bf16(dy_bf16, pad_to=128, colsum=csum)   # [N, Mp]
        xT = ops.transpose_bf16(x_bf16, pad_to=128)     # [K, Mp]
        Mp = dyT.shape[1]
        split = max(1, min(32, Mp // 2048))
        n0 = 0
        for j, w in enumerate(weights):
            n1 = n0 + w.shape[0]
            if pending[j]:
                gw = grads[id(w)].view(w.shape[0], -1)
                if Mp >= 4096 and ops.gemm_nt_splitk(dyT[n0:n1], xT, gw, accumulate=True):
                    pass  # 256x256 kernel, one (tile, K-slice) per CU, partials summed by a second kernel
                elif split > 1:
                    ops.gemm_nt(dyT[n0:n1], xT, out_f32=gw, split_k=split)
                else:
                    ops.gemm_nt(dyT[n0:n1], xT, out_f32=gw, residual=gw)
            n0 = n1
        if csum is not None:
            n0 = 0
            for w, b, wb in zip(weights, biases, want_b):
                n1 = n0 + w.shape[0]
                if wb and not direct:
                    grads[id(b)].view(-1).add_(csum[n0:n1])
                n0 = n1
            bias_done = True
    if not bias_done:
        n0 = 0
        for w, b, wb in zip(weights, biases, want_b):
            n1 = n0 + w.shape[0]
            if wb:
                ops.colsum_bf16(dy_bf16[:, n0:n1], grads[id(b)])
            n0 = n1


class GradBucket:
    """One flat, zero-initialised fp32 buffer holding the gradients of a tower's trainable parameters."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros((max(n, 1),), dtype=F32, device=dev)
        self.views, off = {}, 0
        for p in self.params:
            self.views[id(p)] = self.flat[off : off + p.numel()].view(p.shape)
            off += p.numel()

    def ordered(self):
        return [self.views[id(p)] for p in self.params]


def dense_head_backward(dout_f32: torch.Tensor, x_bf16: torch.Tensor, weight: torch.nn.Parameter, bias: torch.nn.Parameter,
                        grads: dict, out_bf16: bool):
    """y = x W^T + b with a TRAINABLE W [D,K]:  dW += dy^T x, db += colsum(dy), returns dx = dy W.
    dout_f32 [M,D] fp32 (M small: one row per sample) or bf16 [M,D]."""
    dy_b = ops.cast_bf16(dout_f32) if dout_f32.dtype == F32 else dout_f32
    M, D = dy_b.shape
    K = x_bf16.shape[1]
    fused_bias = D % 8 == 0 and dy_b.stride(0) % 8 == 0
    dyT = ops.transpose_bf16(dy_b, pad_to=128, colsum=grads[id(bias)].view(-1) if fused_bias else None)   # [D, Mp] (+ db)
    xT = ops.transpose_bf16(x_bf16, pad_to=128)         # [K, Mp]
    Mp = dyT.shape[1]
    split = max(1, min(32, Mp // 2048))
    gw = grads[id(weight)]
    if Mp >= 4096 and ops.gemm_nt_splitk(dyT, xT, gw, accumulate=True):
        pass  # long contraction (MLM decoder over all tokens): 256x256 split-K workspace path
    elif split > 1:
        ops.gemm_nt(dyT, xT, out_f32=gw, split_k=split)      # atomically accumulates into the zeroed bucket
    else:
        ops.gemm_nt(dyT, xT, out_f32=gw, residual=gw)        # accumulate in place
    if not fused_bias:
        ops.colsum_bf16(dy_b, grads[id(bias)])
    w_t = ops.cast_transpose_bf16(_f32c(weight))   # [K, D]
    if out_bf16:
        dx = torch.empty((M, K), dtype=BF16, device=dy_b.device)
        ops.gemm_nt(dy_b, w_t, out_bf16=dx)
    else:
        dx = torch.empty((M, K), dtype=F32, device=dy_b.device)
        ops.gemm_nt(dy_b, w_t, out_f32=dx)
    return dx
