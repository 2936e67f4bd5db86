"""CPU ORACLE for the CLIBD contrastive training step — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (clibd_amd/) never does and has no CPU fallback.

What it is: a plain-PyTorch fp32 restatement of the arithmetic on the reference's hot path
(scripts/train_cl.py -> bioscanclip/epoch/train_epoch.py:21-63), module by module, with the reference's
attribute / state-dict names so that (a) the reference's own wrapper classes can wrap these bodies when the
golden vectors are generated (tests/golden/make_golden.py, run in the authoring container only) and (b) golden
state dicts load here unchanged.

Pinning status:
  * losses            — pinned: compared with the imported reference `ContrastiveLoss` / `ClipLoss`
                        (bioscanclip/model/loss_func.py:25-69,110-201) on committed golden vectors.
  * BERT towers       — pinned: the reference `CLIBDDNAEncoder` / `CLIBDLanguageEncoder`
                        (model/dna_encoder.py:80-137, model/language_encoder.py:36-89) wrap HF transformers'
                        BertForMaskedLM / BertModel; golden IO + grads committed.
  * LoRA surgery/head — pinned by importing the reference `CLIBDImageEncoder` / `_LoRA_qkv_timm`
                        (model/image_encoder.py:13-107) around `VisionTransformer` below.
  * ViT body          — third-party timm ~=1.0.9 (requirements.txt:9) is NOT in /root/reference and not installed:
                        restated from its published architecture (vit_base_patch16_224) and cross-checked against
                        transformers.ViTModel.  "parity unpinned" at that boundary in the strict sense.
  * top-k             — faiss-gpu 1.7.2 (requirements.txt:22) absent: exact fp32 inner product + stable top-k.

`precision("bf16")` switches every linear layer / attention product to bf16-rounded operands with fp32
accumulation — the same rounding points as the HIP kernels (torch.autocast(bf16) in the reference,
epoch/train_epoch.py:42-46); `precision("fp32")` is the reference's CPU path.

`precision("fp8")` (BASELINE.json configs[4]; the reference has no such mode, so this one is pinned to the reference only
through the bf16 mode it extends): the bf16 mode with the four forward GEMMs of every transformer layer (QKV, attention
output, fc1, fc2) evaluated on OCP e4m3 operands exactly as the kernels quantise them — the activation as
e4m3(clamp(value x scale, +-448)) taken from the producer's fp32 value, the frozen weight per output row as
e4m3(w x 448 / max|w_row|), fp32 accumulation, de-quantisation by 1 / (row scale x activation scale) — while the backward
stays the bf16 network's (dgrad through the bf16 weight, straight-through quantisers).  Scales per layer and site come from
`set_fp8_scales` (the values the HIP towers use) or default to FP8_SCALES.
"""
from __future__ import annotations

import contextlib
import itertools
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

_PRECISION = "fp32"


@contextlib.contextmanager
def precision(mode: str):
    global _PRECISION
    assert mode in ("fp32", "bf16", "fp8")
    old, _PRECISION = _PRECISION, mode
    try:
        yield
    finally:
        _PRECISION = old


_LN_FOLD = False


@contextlib.contextmanager
def ln_fold(on: bool = True):
    """bf16 mode only: norm2 -> mlp.fc1 of every ViT block but the last evaluated as the kernels' algebraic fold
    (clibd_amd.engine numerics ln_fold="on"): h = rstd (bf16(x) . bf16(gamma o W)^T - mean s) + (b + W beta), statistics from fp32 row
    sums (var = E[x^2] - mean^2).  The VALUE is the fold's; the gradient is the unfolded pair's (the kernels' backward is the unchanged
    LayerNorm / dgrad backward), through a straight-through substitution."""
    global _LN_FOLD
    old, _LN_FOLD = _LN_FOLD, bool(on)
    try:
        yield
    finally:
        _LN_FOLD = old


_DGRAD8 = False


@contextlib.contextmanager
def dgrad8(on: bool = True):
    """bf16 / fp8 modes: the activation-gradient products of attention projection, fc1 and fc2 of every layer evaluated as the kernels'
    8-bit dgrad (clibd_amd.engine numerics dgrad="fp8", include/clibd_hip.h clibd_gemm_fp8_dgrad_nt / clibd_layernorm_bwd_fp8):
      gradient rows  : e4m3(g[m,:] * s_m), s_m = 2^(7 - floor(log2 max|g[m,:]|)) taken from the fp32 gradient (1 for a zero row);
      d(fc1 out)     : e4m3(value * s_m * c2) with the s_m of the fc2 dgrad's operand rows and the per-layer constant
                       c2 = 2^floor(log2(448 / (256 * 1.13 * l1max))), l1max the largest row l1 norm of the quantised fc2^T image;
      weights        : bf16(W)^T with one power-of-two scale per row (input channel), e4m3; fp32 accumulation.
    Weight gradients (full fine-tune) stay the bf16 network's: the mode has no 8-bit weight gradient."""
    global _DGRAD8
    old, _DGRAD8 = _DGRAD8, bool(on)
    try:
        yield
    finally:
        _DGRAD8 = old


def _dg8() -> bool:
    return _DGRAD8 and _PRECISION != "fp32"


def _r(x: torch.Tensor) -> torch.Tensor:
    """bf16 rounding with a straight-through gradient (autocast casts are differentiable identities)."""
    if _PRECISION == "fp32":
        return x
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def _rg(x: torch.Tensor) -> torch.Tensor:
    """round the *gradient* flowing back through x to bf16 (grad of a bf16 tensor is bf16 under autocast)."""
    if _PRECISION == "fp32" or not x.requires_grad:
        return x

    class _G(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            return g.to(torch.bfloat16).to(g.dtype)

    return _G.apply(x)


# ---- dropout with the kernels' counter-based masks (include/clibd_hip.h "Dropout") --------------------------------------
_DROPOUT = None  # (p_hidden, p_attention, base_seed) while active


@contextlib.contextmanager
def dropout(p_hidden: float, p_attention: float, base_seed: int):
    """HF BERT train-mode dropout, evaluated with the same (seed, element index) hash masks as the HIP kernels."""
    global _DROPOUT
    old, _DROPOUT = _DROPOUT, (p_hidden, p_attention, base_seed)
    try:
        yield
    finally:
        _DROPOUT = old


def _lowbias32(x):
    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    return x ^ (x >> 16)


def derive_seed(base: int, layer: int, site: int) -> int:
    return int(_lowbias32(torch.tensor((base ^ ((layer * 8 + site + 1) * 0x9E3779B1)) & 0xFFFFFFFF, dtype=torch.int64)))


def drop_factor(seed: int, idx: torch.Tensor, p: float) -> torch.Tensor:
    """0 or 1/(1-p) per element index (int64 tensor): 16 bits of lowbias32((idx >> 1) ^ seed), low half for even idx."""
    thr = int(round(p * 65536.0))
    if thr == 0:
        return torch.ones(idx.shape)
    h = _lowbias32((idx >> 1) ^ seed)
    bits = torch.where((idx & 1) == 1, h >> 16, h & 0xFFFF)
    return (bits >= thr).float() * (1.0 / (1.0 - thr / 65536.0))


def _hidden_drop(y, layer, site):
    """dropout on a [B,S,H] activation, element index (b*S+s)*H + col"""
    if _DROPOUT is None:
        return y
    p_h, _, base = _DROPOUT
    idx = torch.arange(y.numel(), dtype=torch.int64).view(y.shape)
    return y * drop_factor(derive_seed(base, layer, site), idx, p_h).to(y.dtype)


# ---- fp8-forward mode (clibd_amd.engine.TransformerStack.enable_fp8) ------------------------------------
FP8_SITES = ("qkv_in", "proj_in", "fc1_in", "fc2_in")          # activation sites, named by the GEMM that consumes them
FP8_SCALES = dict(qkv_in=8.0, proj_in=32.0, fc1_in=8.0, fc2_in=4.0)
_E4M3_MAX = 448.0


def e4m3(x: torch.Tensor) -> torch.Tensor:
    """values of OCP e4m3 (round to nearest even, saturating at +-448): what v_cvt_pk_fp8_f32 gives on clamped inputs"""
    return x.clamp(-_E4M3_MAX, _E4M3_MAX).to(torch.float8_e4m3fn).to(x.dtype)


def quantize_rows_e4m3(w: torch.Tensor):
    """per output channel: (e4m3(w_n * s_n), s_n) with s_n = 448 / max_k |w_nk| (1 for an all-zero row) — clibd_quantize_rows_fp8"""
    amax = w.abs().amax(dim=1, keepdim=True)
    s = torch.where(amax > 0, _E4M3_MAX / amax, torch.ones_like(amax))
    return e4m3(w * s), s


class _Fp8Linear(torch.autograd.Function):
    """forward: (e4m3(x * sa) @ e4m3(w * s_n)^T) / (s_n * sa) in fp32;  backward: the bf16 network's dgrad dy_bf16 @ w_bf16
    (no weight gradient: the mode needs frozen base weights)."""

    @staticmethod
    def forward(ctx, x, weight, sa):
        w8, sn = quantize_rows_e4m3(weight.detach().float())
        ctx.save_for_backward(weight)
        return F.linear(e4m3(x.detach().float() * sa), w8) * (1.0 / (sn * sa)).view(-1)

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        rb = lambda t: t.to(torch.bfloat16).to(t.dtype)
        return rb(dy) @ rb(weight.detach()), None, None


def set_fp8_scales(layers, per_layer=None, last_block_qkv_only: bool = False):
    """Attach the per-layer activation scales of the fp8 mode to a stack of oracle layers (ViT `blocks` or BERT
    `encoder.layer`): per_layer = [{site: scale}] as in TransformerStack.fp8 (None: FP8_SCALES everywhere).
    last_block_qkv_only: the ViT tower evaluates its last block on the class row only and keeps that remainder
    (projection, MLP) in bf16 (clibd_amd/engine.py, cls_only_last)."""
    n = len(layers)
    for i, layer in enumerate(layers):
        d = dict(FP8_SCALES if per_layer is None else per_layer[i])
        if last_block_qkv_only and i == n - 1:
            d = {k: v for k, v in d.items() if k == "qkv_in"}   # (a site selection may not carry qkv_in at all)
        for m in layer.modules():
            m._fp8 = d


def _fp8_scale(module, site):
    if _PRECISION != "fp8":
        return None
    d = getattr(module, "_fp8", None)
    if d is None:
        d = FP8_SCALES
    return d.get(site)


def pow2_row_scale(v: torch.Tensor) -> torch.Tensor:
    """s = 2^(7 - floor(log2 max|row|)) per row of the last axis (1 for an all-zero row): the scaled row maximum lies in [128, 256)"""
    amax = v.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax))))
    return torch.where(amax > 0, torch.exp2(7.0 - e), torch.ones_like(amax))


def quantize_rows_e4m3_pow2(w: torch.Tensor):
    """(e4m3(w_n * s_n), s_n) with the power-of-two row scale — clibd_quantize_rows_fp8_bf16"""
    s = pow2_row_scale(w)
    return e4m3(w * s), s


_DG8_ROWS = {}   # id(fc2 weight) -> s_m * c2 of the layer's current backward (set by the fc2 dgrad, read by the fc1 dgrad that follows it)


class _Dgrad8Linear(torch.autograd.Function):
    """value: y (computed by the caller on the mode's forward operands); input gradient: the 8-bit dgrad, see dgrad8()."""

    @staticmethod
    def forward(ctx, x, weight, y, site, pair_id):
        # Round 6: trainable base weights (full fine-tune) — the input gradient is still the 8-bit dgrad, the WEIGHT gradient the bf16 network's
        # (bf16 operands dy^T . x, fp32 accumulation: the kernels hand the weight gradient a bf16 copy of the same fp32 gradient), and the
        # per-layer constant c2 carries one binade of headroom (engine.TransformerStack.refresh re-derives it every 64 steps only).
        ctx.trainable = bool(weight.requires_grad)
        ctx.save_for_backward(weight, x if ctx.trainable else None)
        ctx.site, ctx.pair_id = site, pair_id
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        weight, x = ctx.saved_tensors
        rb = lambda t: t.to(torch.bfloat16).to(t.dtype)
        w8, sn = quantize_rows_e4m3_pow2(rb(weight.detach().float()).t().contiguous())   # [in, out]: rows = the dgrad's output channels
        g = dy.float()
        if ctx.site == "fc1":      # d(fc1 out): the scale it was WRITTEN with by the fc2 dgrad's epilogue
            sm = _DG8_ROWS.pop(ctx.pair_id)
        else:
            sm = pow2_row_scale(g)
            if ctx.site == "fc2":
                l1max = float((w8.abs().sum(dim=1, keepdim=True) / sn).max())
                _DG8_ROWS[ctx.pair_id] = sm * (2.0 ** (math.floor(math.log2(448.0 / (256.0 * 1.13 * max(l1max, 1e-30)))) - (1 if ctx.trainable else 0)))
        dx = (e4m3(g * sm) @ (w8 / sn).t()) / sm
        dw = None
        if ctx.trainable:
            dw = rb(g).reshape(-1, g.shape[-1]).t() @ rb(x.detach().float()).reshape(-1, x.shape[-1])
        return dx, dw, None, None, None


def olinear(x, weight, bias=None, round_out=True, fp8_scale=None, dgrad=None):
    """nn.Linear with the kernels' rounding points: bf16 operands, fp32 accumulate, fp32 bias, bf16 output.
    round_out=False: the GEMM epilogue keeps fp32 (residual add / fp32 head output fused before any rounding);
    the gradient entering the GEMM is still bf16 (the dgrad GEMM's A operand).
    fp8_scale (fp8 mode, one of the four layer GEMMs): e4m3 operands, x quantised from its fp32 value with that scale.
    dgrad = (site, fc2 weight) for the three GEMMs the 8-bit dgrad covers ("proj", "fc1", "fc2"): under dgrad8() the input gradient is
    that mode's, taken from the fp32 output gradient (the kernels quantise what the LayerNorm backward / the GELU epilogue holds in fp32)."""
    y = _Fp8Linear.apply(x, weight, float(fp8_scale)) if fp8_scale is not None else F.linear(_r(x), _r(weight), None)
    dg = dgrad is not None and _dg8()
    if dg:
        y = _Dgrad8Linear.apply(x, weight, y.detach(), dgrad[0], id(dgrad[1]))
    if bias is not None:
        y = y + bias
    y = _r(y) if round_out else y
    return y if dg else _rg(y)


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# =====================================================================================================
# ViT (timm vit_base_patch16_224 architecture; created at model/simple_clip.py:150-153)
# =====================================================================================================
class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch=16, in_chans=3, dim=768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)
        self.num_patches = (img_size // patch) ** 2
        self.patch = patch

    def forward(self, x):
        # conv with stride == kernel is a GEMM over flattened patches, k = c*P*P + py*P + px
        B = x.shape[0]
        P = self.patch
        cols = F.unfold(x, kernel_size=P, stride=P).transpose(1, 2)  # [B, L, C*P*P]
        w = self.proj.weight.reshape(self.proj.weight.shape[0], -1)
        return olinear(cols, w, self.proj.bias, round_out=False).reshape(B, self.num_patches, -1)


def _attn_drop(shape, layer):
    if _DROPOUT is None or layer is None:
        return None
    _, p_a, base = _DROPOUT
    B, nh, Sq, Sk = shape
    bh = torch.arange(B * nh, dtype=torch.int64).view(B, nh, 1, 1)
    q = torch.arange(Sq, dtype=torch.int64).view(1, 1, Sq, 1)
    key = torch.arange(Sk, dtype=torch.int64).view(1, 1, 1, Sk)
    return drop_factor(derive_seed(base, layer, 0), ((bh * Sq + q) << 8) + key, p_a)


def attention_core(q, k, v, mask_add=None, layer=None, round_out=True):
    """q,k,v [B,h,S,dh]; scores and softmax in fp32, probabilities rounded to bf16 before P·V (bf16 mode).
    round_out=False (fp8 mode): the kernel converts its fp32 result straight to e4m3, with no bf16 rounding in between."""
    s = (_r(q) @ _r(k).transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if mask_add is not None:
        s = s + mask_add
    fm = _attn_drop(s.shape, layer)
    if _PRECISION == "fp32":
        p = torch.softmax(s, dim=-1)
        return (p if fm is None else p * fm.to(p.dtype)) @ v
    # kernel rounding points: the UN-normalised exp(s - max) is what gets rounded to bf16 for the P·V product,
    # and the fp32 row sum divides the fp32 result (same function as softmax, same relative rounding error)
    e = torch.exp(s - s.max(dim=-1, keepdim=True).values.detach())
    em = e if fm is None else e * fm.to(e.dtype)
    o = (_r(em) @ _r(v)) / e.sum(dim=-1, keepdim=True)
    return _rg(_r(o) if round_out else o)


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.head_dim = dim // heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x) if not isinstance(self.qkv, nn.Linear) else olinear(x, self.qkv.weight, self.qkv.bias,
                                                                              fp8_scale=_fp8_scale(self, "qkv_in"))
        q, k, v = qkv.reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        sp = _fp8_scale(self, "proj_in")
        o = attention_core(q, k, v, round_out=sp is None).transpose(1, 2).reshape(B, N, C)
        return olinear(o, self.proj.weight, self.proj.bias, round_out=False, fp8_scale=sp, dgrad=("proj", None))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, fold_from=None):
        h = olinear(x, self.fc1.weight, self.fc1.bias, fp8_scale=_fp8_scale(self, "fc1_in"), dgrad=("fc1", self.fc2.weight))
        if fold_from is not None:   # (x1, norm2): the value of the kernels' fold, the gradient of the line above
            x1, ln = fold_from
            rb = lambda t: t.to(torch.bfloat16).to(t.dtype)
            with torch.no_grad():
                xd = x1.detach()
                H = xd.shape[-1]
                mean = xd.sum(-1, keepdim=True) / H
                var = ((xd * xd).sum(-1, keepdim=True) / H - mean * mean).clamp_min(0.0)
                rstd = torch.rsqrt(var + ln.eps)
                wg = rb(self.fc1.weight.detach() * ln.weight.detach())
                hf = rstd * (rb(xd) @ wg.T - mean * wg.sum(-1)) + (self.fc1.bias.detach() + self.fc1.weight.detach() @ ln.bias.detach())
                hf = rb(hf)
            h = h + (hf - h).detach()
        s2 = _fp8_scale(self, "fc2_in")   # fp8: the fc1 epilogue converts gelu(bf16(h)) (fp32) straight to e4m3
        a = gelu_erf(h)
        a = _r(a) if s2 is None else a
        return olinear(a if _dg8() else _rg(a), self.fc2.weight, self.fc2.bias, round_out=False, fp8_scale=s2, dgrad=("fc2", self.fc2.weight))


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        if getattr(self, "_no_fold", False) and _DGRAD8:   # the last block (class row only in the kernels) keeps the bf16 dgrad
            with dgrad8(False):
                return self._forward(x)
        return self._forward(x)

    def _forward(self, x):
        x = x + self.attn(self.norm1(x))
        if _LN_FOLD and _PRECISION == "bf16" and not getattr(self, "_no_fold", False):
            return x + self.mlp(self.norm2(x), fold_from=(x, self.norm2))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    """timm-compatible attribute names: patch_embed.proj, cls_token, pos_embed, blocks[i].{norm1,attn.{qkv,proj},
    norm2,mlp.{fc1,fc2}}, norm, head, reset_classifier — what model/image_encoder.py:49-107 touches."""

    def __init__(self, img_size=224, patch=16, dim=768, depth=12, heads=12, num_classes=1000):
        super().__init__()
        self.embed_dim = dim
        self.patch_embed = PatchEmbed(img_size, patch, 3, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, dim) * 0.02)
        self.blocks = nn.Sequential(*[Block(dim, heads) for _ in range(depth)])
        self.blocks[-1]._no_fold = True   # the kernels evaluate the last block on the class row only, unfolded (engine cls_only_last)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes) if num_classes > 0 else nn.Identity()

    def reset_classifier(self, num_classes: int):
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        x = self.blocks(x)
        return self.norm(x)

    def forward(self, x):
        x = self.forward_features(x)[:, 0]
        if isinstance(self.head, nn.Linear):
            return olinear(x, self.head.weight, self.head.bias, round_out=False)
        return x


class LoRAQKV(nn.Module):
    """model/image_encoder.py:13-46: qkv = W x + b; q += B_q(A_q x); v += B_v(A_v x); no alpha/r scale."""

    def __init__(self, qkv, linear_a_q, linear_b_q, linear_a_v, linear_b_v):
        super().__init__()
        self.qkv, self.linear_a_q, self.linear_b_q, self.linear_a_v, self.linear_b_v = qkv, linear_a_q, linear_b_q, linear_a_v, linear_b_v
        self.dim = qkv.in_features

    def forward(self, x):
        # the kernel adds bias after the rank update and rounds once; algebraically the same sum
        s8 = _fp8_scale(self, "qkv_in")   # fp8: only the base product runs on e4m3 operands; the adapters stay bf16
        base = F.linear(_r(x), _r(self.qkv.weight), None) if s8 is None else _Fp8Linear.apply(x, self.qkv.weight, float(s8))
        tq = olinear(x, self.linear_a_q.weight)
        tv = olinear(x, self.linear_a_v.weight)
        dq = F.linear(_r(tq), _r(self.linear_b_q.weight))
        dv = F.linear(_r(tv), _r(self.linear_b_v.weight))
        zeros = torch.zeros_like(dq)
        out = base + torch.cat([dq, zeros, dv], dim=-1) + self.qkv.bias
        return _rg(_r(out))


def _lora_init(a: nn.Linear, b: nn.Linear):
    nn.init.kaiming_uniform_(a.weight, a=math.sqrt(5))
    nn.init.zeros_(b.weight)


class ImageEncoder(nn.Module):
    """model/image_encoder.py:49-107 (state-dict prefix `base_image_encoder.`)."""

    def __init__(self, vit: VisionTransformer, r: int = 4, num_classes: int = 0, lora_layer=None):
        super().__init__()
        assert r > 0
        layers = lora_layer if lora_layer else list(range(len(vit.blocks)))  # `if lora_layer:` quirk, :54-57
        for p in vit.parameters():
            p.requires_grad = False
        for i, blk in enumerate(vit.blocks):
            if i not in layers:
                continue
            d = blk.attn.qkv.in_features
            aq, bq, av, bv = nn.Linear(d, r, bias=False), nn.Linear(r, d, bias=False), nn.Linear(d, r, bias=False), nn.Linear(r, d, bias=False)
            _lora_init(aq, bq)
            _lora_init(av, bv)
            blk.attn.qkv = LoRAQKV(blk.attn.qkv, aq, bq, av, bv)
        self.base_image_encoder = vit
        if num_classes > 0:
            vit.reset_classifier(num_classes)

    def forward(self, x):
        return self.base_image_encoder(x)


# =====================================================================================================
# BERT (HF transformers BertForMaskedLM / BertModel arithmetic; model/dna_encoder.py:25-37,
# model/language_encoder.py:12-20).  Parameter names follow HF so reference state dicts load.
# =====================================================================================================
class LoRALinear(nn.Module):
    """model/dna_encoder.py:68-77 `_LoRALayer`: w(x) + w_b(w_a(x))."""

    def __init__(self, w, w_a, w_b):
        super().__init__()
        self.w, self.w_a, self.w_b = w, w_a, w_b
        self.in_features = w.in_features

    def forward(self, x):
        s8 = _fp8_scale(self, "qkv_in")
        base = F.linear(_r(x), _r(self.w.weight), None) if s8 is None else _Fp8Linear.apply(x, self.w.weight, float(s8))
        t = olinear(x, self.w_a.weight)
        out = base + F.linear(_r(t), _r(self.w_b.weight)) + self.w.bias
        return _rg(_r(out))


def _lin(mod, x):
    if isinstance(mod, nn.Linear):
        return olinear(x, mod.weight, mod.bias, fp8_scale=_fp8_scale(mod, "qkv_in"))
    return mod(x)


class BertEmbeddings(nn.Module):
    def __init__(self, vocab, hidden, max_pos=512, type_vocab=2, eps=1e-12):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, hidden)
        self.position_embeddings = nn.Embedding(max_pos, hidden)
        self.token_type_embeddings = nn.Embedding(type_vocab, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)

    def forward(self, ids, token_type_ids=None):
        S = ids.shape[1]
        tt = token_type_ids if token_type_ids is not None else torch.zeros_like(ids)
        pos = torch.arange(S, device=ids.device)[None, :]
        return _hidden_drop(self.LayerNorm(self.word_embeddings(ids) + self.token_type_embeddings(tt) + self.position_embeddings(pos)), 255, 3)


class _SelfAttn(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(hidden, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, hidden)
        self.heads = heads

    def forward(self, x, mask_add, layer=None):
        B, S, Hd = x.shape
        sp = lambda t: t.view(B, S, self.heads, Hd // self.heads).transpose(1, 2)
        o = attention_core(sp(_lin(self.query, x)), sp(_lin(self.key, x)), sp(_lin(self.value, x)), mask_add, layer,
                           round_out=_fp8_scale(self, "proj_in") is None)
        return o.transpose(1, 2).reshape(B, S, Hd)


class _DenseLN(nn.Module):
    def __init__(self, din, dout, eps):
        super().__init__()
        self.dense = nn.Linear(din, dout)
        self.LayerNorm = nn.LayerNorm(dout, eps=eps)

    def forward(self, x, residual, layer=None, site=None, fp8_site=None):
        y = olinear(x, self.dense.weight, self.dense.bias, round_out=False, fp8_scale=_fp8_scale(self, fp8_site) if fp8_site else None,
                    dgrad=({"proj_in": "proj", "fc2_in": "fc2"}[fp8_site], self.dense.weight) if fp8_site else None)
        if layer is not None:
            y = _hidden_drop(y, layer, site)
        return self.LayerNorm(y + residual)


class _Attn(nn.Module):
    def __init__(self, hidden, heads, eps):
        super().__init__()
        self.self = _SelfAttn(hidden, heads)
        self.output = _DenseLN(hidden, hidden, eps)


class _Inter(nn.Module):
    def __init__(self, hidden, ff):
        super().__init__()
        self.dense = nn.Linear(hidden, ff)


class BertLayer(nn.Module):
    def __init__(self, hidden, heads, ff, eps=1e-12):
        super().__init__()
        self.attention = _Attn(hidden, heads, eps)
        self.intermediate = _Inter(hidden, ff)
        self.output = _DenseLN(ff, hidden, eps)

    def forward(self, x, mask_add=None, layer=None):
        a = self.attention.output(self.attention.self(x, mask_add, layer), x, layer, 1, "proj_in")
        h = olinear(a, self.intermediate.dense.weight, self.intermediate.dense.bias, fp8_scale=_fp8_scale(self, "fc1_in"),
                    dgrad=("fc1", self.output.dense.weight))
        g = gelu_erf(h)
        g = _r(g) if _fp8_scale(self, "fc2_in") is None else g
        return self.output(g if _dg8() else _rg(g), a, layer, 2, "fc2_in")


class _Encoder(nn.Module):
    def __init__(self, n, hidden, heads, ff):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(hidden, heads, ff) for _ in range(n)])


class _Pooler(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)


class BertModel(nn.Module):
    def __init__(self, vocab=30522, hidden=512, layers=4, heads=8, ff=2048, max_pos=512, pooler=True):
        super().__init__()
        self.embeddings = BertEmbeddings(vocab, hidden, max_pos)
        self.encoder = _Encoder(layers, hidden, heads, ff)
        if pooler:
            self.pooler = _Pooler(hidden)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        x = self.embeddings(input_ids, token_type_ids)
        mask_add = None
        if attention_mask is not None:
            mask_add = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
        for i, layer in enumerate(self.encoder.layer):
            x = layer(x, mask_add, i)
        return x  # last_hidden_state


class _Transform(nn.Module):
    def __init__(self, hidden, eps=1e-12):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)


class _Predictions(nn.Module):
    def __init__(self, hidden, vocab):
        super().__init__()
        self.transform = _Transform(hidden)
        self.decoder = nn.Linear(hidden, vocab)
        self.bias = nn.Parameter(torch.zeros(vocab))


class _Cls(nn.Module):
    def __init__(self, hidden, vocab):
        super().__init__()
        self.predictions = _Predictions(hidden, vocab)


class BertForMaskedLM(nn.Module):
    def __init__(self, vocab=1027, hidden=768, layers=12, heads=12, ff=3072, max_pos=512):
        super().__init__()
        self.bert = BertModel(vocab, hidden, layers, heads, ff, max_pos, pooler=False)
        self.cls = _Cls(hidden, vocab)

    def forward(self, ids):
        x = self.bert(ids)
        t = self.cls.predictions.transform
        h = t.LayerNorm(_rg(gelu_erf(olinear(x, t.dense.weight, t.dense.bias))))  # fp32 GELU output feeds the LayerNorm
        d = self.cls.predictions.decoder
        return olinear(h, d.weight, d.bias)  # logits


def _add_bert_lora(layers, r, lora_layer):
    idx = lora_layer if lora_layer is not None else list(range(len(layers)))  # `is not None` test, dna_encoder.py:85-88
    for i, layer in enumerate(layers):
        if i not in idx:
            continue
        sa = layer.attention.self
        d = sa.query.in_features
        aq, bq, av, bv = nn.Linear(d, r, bias=False), nn.Linear(r, d, bias=False), nn.Linear(d, r, bias=False), nn.Linear(r, d, bias=False)
        _lora_init(aq, bq)
        _lora_init(av, bv)
        sa.query = LoRALinear(sa.query, aq, bq)
        sa.value = LoRALinear(sa.value, av, bv)


class DNAEncoder(nn.Module):
    """model/dna_encoder.py:80-137: LoRA on query/value, MLM decoder replaced by Linear(H, num_classes),
    output = logits.softmax(-1).mean(1)."""

    def __init__(self, model: BertForMaskedLM, r: int = 4, num_classes: int = 0, lora_layer=None):
        super().__init__()
        assert r > 0
        for p in model.parameters():
            p.requires_grad = False
        _add_bert_lora(model.bert.encoder.layer, r, lora_layer)
        self.base_dna_encoder = model
        if num_classes > 0:
            dec = model.cls.predictions.decoder
            model.cls.predictions.decoder = nn.Linear(dec.in_features, num_classes)

    def forward(self, sequence):
        return self.base_dna_encoder(sequence).softmax(dim=-1).mean(dim=1)


class LanguageEncoder(nn.Module):
    """model/language_encoder.py:36-89: proj(last_hidden_state.mean(1)) — mean includes padded positions."""

    def __init__(self, model: BertModel, r: int = 4, num_classes: int = 0, lora_layer=None):
        super().__init__()
        assert r > 0
        for p in model.parameters():
            p.requires_grad = False
        _add_bert_lora(model.encoder.layer, r, lora_layer)
        self.base_language_encoder = model
        if num_classes > 0:
            self.proj = nn.Linear(model.pooler.dense.out_features, num_classes)

    def forward(self, x: dict):
        h = self.base_language_encoder(x["input_ids"], x.get("token_type_ids"), x.get("attention_mask"))
        return olinear(h.mean(dim=1), self.proj.weight, self.proj.bias, round_out=False)


# =====================================================================================================
# SimpleCLIP + losses (model/simple_clip.py:21-61, model/loss_func.py)
# =====================================================================================================
class SimpleCLIP(nn.Module):
    def __init__(self, image_encoder, dna_encoder, language_encoder, init_logit_scale: float = math.log(1 / 0.07)):
        super().__init__()
        self.image_encoder, self.dna_encoder, self.language_encoder = image_encoder, dna_encoder, language_encoder
        self.logit_scale = nn.Parameter(torch.ones([]) * init_logit_scale)

    def forward(self, image_input, dna_input, language_input):
        img = dna = txt = None
        if self.dna_encoder is not None:
            dna = F.normalize(self.dna_encoder(dna_input).float(), p=2, dim=-1)
        if self.image_encoder is not None:
            img = F.normalize(self.image_encoder(image_input).float(), p=2, dim=-1)
        if self.language_encoder is not None:
            txt = F.normalize(self.language_encoder(language_input).float(), p=2, dim=-1)
        return img, dna, txt, self.logit_scale.exp(), None


def label_matrix(labels):
    """loss_func.py:19-22"""
    return (labels[None, :] == labels[:, None]).float()


def soft_ce(logits, targets):
    """nn.CrossEntropyLoss() with probability targets: mean_i( -sum_j t_ij log_softmax(logits)_ij )."""
    return -(targets * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()


def contrastive_loss(features, labels, logit_scale, bind_to=None, no_image_text_loss=False):
    """loss_func.py:41-69 / :159-201.  `features` = [image, dna, text] (None allowed); the loop over ordered pairs,
    the second normalisation and both directions per pair are kept exactly as the reference evaluates them."""
    present = [(i, f) for i, f in enumerate(features) if f is not None]
    if len(present) < 2:
        raise ValueError("Too less element for calculating the contrastive loss.")
    T = label_matrix(labels)
    bind = {"image": 0, "dna": 1, "text": 2}.get(bind_to) if bind_to is not None else None
    feats = [f for _, f in present]
    terms = []
    # NB: the reference indexes the *filtered* list (idx 0/1/2 are positions after dropping None entries)
    for ia, fa in enumerate(feats):
        for ib, fb in enumerate(feats):
            if bind is not None and ia != bind and ib != bind:
                continue
            if ia == ib:
                continue
            if no_image_text_loss and (ia == 0 or ib == 0) and (ia == 2 or ib == 2):
                continue
            a, b = F.normalize(fa, p=2, dim=1), F.normalize(fb, p=2, dim=1)
            terms.append(soft_ce(logit_scale * a @ b.T, T))
            terms.append(soft_ce(logit_scale * b @ a.T, T))
    return sum(terms) / len(terms)


# =====================================================================================================
# batch contract helpers (util/util.py:77-98, model/dna_encoder.py:53-63) and eval top-k (util/util.py:521-528)
# =====================================================================================================
_KMER_ID = {"".join(k): 3 + i for i, k in enumerate(itertools.product("ACGT", repeat=5))}


def kmer_tokenize(seq: str, max_len: int = 660, k: int = 5) -> list:
    """[0] + ids of the non-overlapping 5-mers of the sequence padded with 'N' / truncated to 660 nt.
    specials <MASK>=0, <CLS>=1, <UNK>=2; k-mers in product('ACGT', repeat=5) order start at 3."""
    s = seq[:max_len] if len(seq) > max_len else seq + "N" * (max_len - len(seq))
    return [0] + [_KMER_ID.get(s[i : i + k], 2) for i in range(0, len(s) - k + 1, k)]


def topk_inner_product(query, keys, k=5):
    """faiss.IndexFlatIP.search on L2-normalised inputs: exact fp32 scores, ties -> lower index."""
    q = F.normalize(query.float(), dim=1)
    kk = F.normalize(keys.float(), dim=1)
    # per-pair sums in one fixed order (a blocked GEMM may round identical rows differently): small cases only
    s = torch.cat([(q[i : i + 16, None, :] * kk[None, :, :]).sum(-1) for i in range(0, q.shape[0], 16)], dim=0)
    idx = torch.argsort(-s, dim=1, stable=True)[:, :k]
    return torch.gather(s, 1, idx), idx


# =====================================================================================================
# one training step (epoch/train_epoch.py:21-63) on CPU — the cpu_baseline leg of bench.py
# =====================================================================================================
def build_image_dna_model(dim=768, depth=12, heads=12, dna_layers=12, out_dim=768, seed=42, img_size=224):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    vit = VisionTransformer(img_size=img_size, dim=dim, depth=depth, heads=heads, num_classes=0)
    bert = BertForMaskedLM(vocab=1027, hidden=dim, layers=dna_layers, heads=heads, ff=4 * dim)
    del g
    return SimpleCLIP(ImageEncoder(vit, 4, out_dim), DNAEncoder(bert, 4, out_dim), None)


def train_step(model: SimpleCLIP, optimizer, image, dna, labels, text=None):
    optimizer.zero_grad(set_to_none=True)
    img, dn, txt, scale, _ = model(image, dna, text)
    loss = contrastive_loss([img, dn, txt], labels, scale)
    loss.backward()
    optimizer.step()
    return loss.detach()
