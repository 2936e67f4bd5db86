"""Image tower: drop-in for `bioscanclip.model.image_encoder` (reference model/image_encoder.py:13-107).

`CLIBDImageEncoder(vit_model, r, num_classes=0, lora_layer=None)` keeps the reference's constructor, attribute
names (`base_image_encoder`, `w_As`, `w_Bs`, `lora_layer`) and state-dict keys
(`base_image_encoder.blocks.{i}.attn.qkv.{qkv,linear_a_q,linear_b_q,linear_a_v,linear_b_v}.weight`, ...), but its
forward runs the hand-written gfx950 kernels (clibd_amd.towers.ViTTower) instead of timm's PyTorch modules.
`vit_model` may be the parameter container below (`create_vit`) or a real timm VisionTransformer: only parameter
attributes are read.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..towers import ViTTower


class _ParamOnly(nn.Module):
    """Modules of the containers hold parameters; their arithmetic lives in the HIP engine."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter container; run it through the CLIBD encoder wrapper")


class _Attention(_ParamOnly):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.head_dim = dim // heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_ParamOnly):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(_ParamOnly):
    def __init__(self, dim, heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(_ParamOnly):
    def __init__(self, dim, patch=16, in_chans=3):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)


class VisionTransformer(_ParamOnly):
    """timm `vit_*_patch16_224`-shaped parameter tree (names as in timm.models.vision_transformer)."""

    def __init__(self, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, num_classes=1000):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        self.num_classes = num_classes
        self.patch_embed = _PatchEmbed(embed_dim, patch_size)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.Sequential(*[_Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self._init_weights()

    def _init_weights(self):
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def reset_classifier(self, num_classes: int, global_pool=None):
        self.num_classes = num_classes
        dev = self.norm.weight.device
        self.head = (nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()).to(dev)


_VIT_ZOO = {
    "vit_base_patch16_224": dict(embed_dim=768, depth=12, num_heads=12),
    "vit_small_patch16_224": dict(embed_dim=384, depth=12, num_heads=6),
    "vit_large_patch16_224": dict(embed_dim=1024, depth=24, num_heads=16),
}


def create_vit(name: str = "vit_base_patch16_224", num_classes: int = 1000, **overrides) -> VisionTransformer:
    """Counterpart of `timm.create_model(name)` (model/simple_clip.py:150-153) without the pretrained download."""
    if name not in _VIT_ZOO:
        raise ValueError(f"unknown ViT '{name}' (have {sorted(_VIT_ZOO)})")
    cfg = dict(_VIT_ZOO[name])
    cfg.update(overrides)
    return VisionTransformer(num_classes=num_classes, **cfg)


class _LoRA_qkv_timm(_ParamOnly):
    """Holder with the reference's attribute names (model/image_encoder.py:13-46):
    qkv = W x + b;  q += B_q(A_q x);  v += B_v(A_v x)  — evaluated inside the fused QKV GEMM."""

    def __init__(self, qkv: nn.Module, linear_a_q: nn.Module, linear_b_q: nn.Module, linear_a_v: nn.Module, linear_b_v: nn.Module):
        super().__init__()
        self.qkv = qkv
        self.linear_a_q = linear_a_q
        self.linear_b_q = linear_b_q
        self.linear_a_v = linear_a_v
        self.linear_b_v = linear_b_v
        self.dim = qkv.in_features
        self.in_features, self.out_features = qkv.in_features, qkv.out_features


class CLIBDImageEncoder(nn.Module):
    def __init__(self, vit_model, r: int, num_classes: int = 0, lora_layer=None):
        super().__init__()
        assert r > 0
        # reference quirk (image_encoder.py:54-57): `if lora_layer:` — an empty list still wraps every block
        self.lora_layer = lora_layer if lora_layer else list(range(len(vit_model.blocks)))
        self.w_As, self.w_Bs = [], []
        for p in vit_model.parameters():
            p.requires_grad = False
        self._lora = {}
        for i, blk in enumerate(vit_model.blocks):
            if i not in self.lora_layer:
                continue
            base = blk.attn.qkv
            self.dim = base.in_features
            dev = base.weight.device
            a_q, b_q = nn.Linear(self.dim, r, bias=False).to(dev), nn.Linear(r, self.dim, bias=False).to(dev)
            a_v, b_v = nn.Linear(self.dim, r, bias=False).to(dev), nn.Linear(r, self.dim, bias=False).to(dev)
            self.w_As += [a_q, a_v]
            self.w_Bs += [b_q, b_v]
            blk.attn.qkv = _LoRA_qkv_timm(base, a_q, b_q, a_v, b_v)
            self._lora[i] = blk.attn.qkv
        self.reset_parameters()
        self.base_image_encoder = vit_model
        if num_classes > 0:
            self.base_image_encoder.reset_classifier(num_classes=num_classes)
        self._tower = None

    def reset_classifier(self, num_classes):
        self.base_image_encoder.reset_classifier(num_classes=num_classes)

    def reset_parameters(self) -> None:
        for w_A in self.w_As:
            nn.init.kaiming_uniform_(w_A.weight, a=math.sqrt(5))
        for w_B in self.w_Bs:
            nn.init.zeros_(w_B.weight)

    def tower(self) -> ViTTower:
        if self._tower is None:
            self._tower = ViTTower(self.base_image_encoder, self._lora)
        return self._tower

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.tower()(x)
