"""The three encoder towers as autograd Functions over the HIP engine (clibd_amd.engine).

ViTTower      = timm vit_base_patch16_224 forward/backward + LoRA(q,v) + trainable head
                (reference: model/image_encoder.py:49-107; timm created at model/simple_clip.py:150-153)
BertTower     = HF BERT encoder (post-LN) + LoRA(query,value) with one of two heads:
   head="mlm"   BertForMaskedLM transform + replaced decoder, softmax(-1).mean(1)   (model/dna_encoder.py:80-137)
   head="mean"  last_hidden_state.mean(1) -> proj                                  (model/language_encoder.py:36-89)
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .engine import BF16, F32, GradBucket, LayerSpec, LoraParams, NotSupportedYet, TransformerStack, _f32c, dense_head_backward, linear_wgrad


def _trainable(params):
    return [p for p in params if p.requires_grad]


class _TowerFn(torch.autograd.Function):
    """Generic bridge: forward(tower, save, inputs(tuple of non-differentiable tensors), *trainable_params)."""

    @staticmethod
    def forward(ctx, tower, save, inputs, *params):
        out, state = tower._forward(inputs, save)
        ctx.tower, ctx.state, ctx.params = tower, state, params
        return out

    @staticmethod
    def backward(ctx, dout):
        tower, state = ctx.tower, ctx.state
        if state is None:
            raise RuntimeError("backward through a tower that ran without saving activations")
        params = _trainable(tower.trainable_params())  # same objects/order as the apply() call
        assert len(params) == len(ctx.params)
        sink = tower.grad_sink
        if sink is not None and all(id(p) in sink for p in params):
            # trainer-owned flat gradient bucket (clibd_amd.optim.FusedAdamW): accumulate straight into it
            tower._backward(dout.contiguous().to(F32), state, sink)
            tower._ready(None)  # every gradient of this tower is final
            ctx.state = None
            return (None, None, None, *([None] * len(params)))
        bucket = GradBucket(params)
        tower._backward(dout.contiguous().to(F32), state, bucket.views)
        ctx.state = None
        return (None, None, None, *bucket.ordered())


class _Tower:
    grad_sink = None  # optional {id(param): fp32 accumulation tensor}; set by the trainer, see _TowerFn.backward

    def trainable_params(self) -> List[torch.nn.Parameter]:
        raise NotImplementedError

    training = False  # set by the owning nn.Module before each call (module.training)

    # Gradient-ready protocol (data-parallel full fine-tune: the trainer starts a group's all-reduce while the backward is
    # still walking down the tower).  grad_groups(): parameter lists in the order their gradients become final; the backward
    # calls on_grads_ready(k) after group k (None: everything), always from the stream the gradients were produced on.
    on_grads_ready = None

    def grad_groups(self) -> List[List[torch.nn.Parameter]]:
        return [self.trainable_params()]

    def _ready(self, k):
        if self.on_grads_ready is not None:
            self.on_grads_ready(k)

    def invalidate_weight_images(self):
        """Forget every cached bf16 / fp8 image of the frozen weights (they were rewritten in place behind autograd's back,
        e.g. by the trainer's start-up broadcast through `.data`): the next forward rebuilds them."""
        self.stack._cache_key = None
        for attr in ("_patch_key", "_head_key"):
            if hasattr(self, attr):
                setattr(self, attr, None)

    def _stack_groups(self, head, embed):
        n = len(self.stack.layers)
        return [head] + [self.stack.layer_params(i) for i in range(n - 1, -1, -1)] + [embed]

    def __call__(self, *inputs):
        params = _trainable(self.trainable_params())
        save = torch.is_grad_enabled() and len(params) > 0
        if not save:
            with torch.no_grad():
                out, _ = self._forward(inputs, False)
            return out
        return _TowerFn.apply(self, True, inputs, *params)


# =========================================================================================================
class ViTTower(_Tower):
    def __init__(self, vit, lora_modules: dict):
        """vit: timm-shaped module tree (patch_embed.proj, cls_token, pos_embed, blocks[i].{norm1,attn.{qkv,proj},norm2,
        mlp.{fc1,fc2}}, norm, head).  lora_modules: {block index: wrapper with qkv/linear_{a,b}_{q,v}}."""
        self.vit = vit
        pw = vit.patch_embed.proj.weight
        if tuple(pw.shape[1:]) != (3, 16, 16) or tuple(vit.pos_embed.shape[:2]) != (1, 197):
            raise NotSupportedYet("ViT tower is specialised for 224x224 images with 16x16 patches (vit_*_patch16_224)")
        H = pw.shape[0]
        layers = []
        for i, blk in enumerate(vit.blocks):
            q = blk.attn.qkv
            lora = None
            if i in lora_modules:
                w = lora_modules[i]
                base = w.qkv
                lora = LoraParams(w.linear_a_q.weight, w.linear_b_q.weight, w.linear_a_v.weight, w.linear_b_v.weight)
            else:
                base = q
            layers.append(LayerSpec([base.weight], [base.bias], blk.attn.proj.weight, blk.attn.proj.bias, blk.mlp.fc1.weight,
                                    blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.norm1.weight, blk.norm1.bias,
                                    blk.norm2.weight, blk.norm2.bias, lora))
        heads = getattr(vit.blocks[0].attn, "num_heads", H // 64)
        self.H = H
        self.stack = TransformerStack(layers, H, heads, pre_ln=True, eps=float(vit.blocks[0].norm1.eps))
        self._patch_key, self._patch_w = None, None

    def trainable_params(self):
        """Every parameter the tower can produce a gradient for (the caller filters by requires_grad): adapters and head
        always; base weights, embeddings and norms in full fine-tune mode (model_config.disable_lora)."""
        ps = []
        for L in self.stack.layers:
            if L.lora is not None:
                ps += L.lora.tensors()
        head = self.vit.head
        if isinstance(head, torch.nn.Linear):
            ps += [head.weight, head.bias]
        return ps + self.stack.base_params() + self._frozen_extra()

    def _frozen_extra(self):
        v = self.vit
        return [v.patch_embed.proj.weight, v.patch_embed.proj.bias, v.cls_token, v.pos_embed, v.norm.weight, v.norm.bias]

    def grad_groups(self):
        v = self.vit
        head = ([v.head.weight, v.head.bias] if isinstance(v.head, torch.nn.Linear) else []) + [v.norm.weight, v.norm.bias]
        return self._stack_groups(head, [v.patch_embed.proj.weight, v.patch_embed.proj.bias, v.cls_token, v.pos_embed])

    def _full(self):
        return self.stack.full_mode() or any(p.requires_grad for p in self._frozen_extra())

    def _forward(self, inputs, save):
        (image,) = inputs
        v = self.vit
        if image.dim() != 4 or tuple(image.shape[1:]) != (3, 224, 224):
            raise ValueError("image encoder expects [B,3,224,224]")
        full = save and self._full()
        self.stack.refresh()
        self.stack.pack_lora()
        B, S, H = image.shape[0], 197, self.H
        # uint8 images (the bytes the dataset holds) stay bytes until the patch gather reads them as u8 / 255: a quarter of the PCIe traffic
        img = image.detach().contiguous() if image.dtype == torch.uint8 else image.detach().to(F32).contiguous()
        pw = v.patch_embed.proj.weight
        key = (pw._version, pw.data_ptr())
        if key != self._patch_key or pw.requires_grad:  # trainable: the fused optimizer rewrites it in place every step
            self._patch_w = ops.cast_bf16(_f32c(pw).reshape(H, 768))
            self._patch_key = key
        patches = ops.patchify(img)
        proj = torch.empty((B * 196, H), dtype=F32, device=img.device)
        ops.gemm_nt(patches, self._patch_w, bias=_f32c(v.patch_embed.proj.bias), out_f32=proj)
        tok = ops.vit_assemble_tokens(proj, _f32c(v.cls_token).reshape(H), _f32c(v.pos_embed).reshape(S * H), B)
        # timm pools token 0 (global_pool='token'): only the class row of the last block is live
        xcls, _, saved = self.stack.forward(tok.view(B * S, H), None, None, B, S, None, save, cls_only_last=True, full=full)
        x = xcls
        # final norm on the class token only (LayerNorm is per token), then the trainable head
        st = torch.empty((B, 2), dtype=F32, device=x.device)
        xn = torch.empty((B, H), dtype=BF16, device=x.device)
        ops.layernorm_fwd(xcls, _f32c(v.norm.weight), _f32c(v.norm.bias), float(v.norm.eps), y_bf16=xn, stats=st)
        head = v.head
        if isinstance(head, torch.nn.Linear):
            D = head.weight.shape[0]
            out = torch.empty((B, D), dtype=F32, device=x.device)
            ops.gemm_nt(xn, ops.cast_bf16(_f32c(head.weight)), bias=_f32c(head.bias), out_f32=out)
        else:
            out = xn.to(F32)
        state = dict(saved=saved, xcls=xcls, st=st, xn=xn, B=B, full=full, patches=patches if full else None) if save else None
        return out, state

    def _backward(self, dout, state, grads):
        v, B, S, H = self.vit, state["B"], 197, self.H
        head = v.head
        if isinstance(head, torch.nn.Linear):
            dxn = dense_head_backward(dout, state["xn"], head.weight, head.bias, grads, out_bf16=True)
        else:
            dxn = ops.cast_bf16(dout)
        dxcls = torch.empty((B, H), dtype=F32, device=dout.device)
        dxcls_b = torch.empty((B, H), dtype=BF16, device=dout.device)
        full = state["full"]
        pg = dict(dgamma=grads[id(v.norm.weight)].view(-1), dbeta=grads[id(v.norm.bias)].view(-1)) if full and id(v.norm.weight) in grads else {}
        ops.layernorm_bwd(dxn, state["xcls"], state["st"], _f32c(v.norm.weight), dx_f32=dxcls, dx_bf16=dxcls_b, **pg)
        self._ready(0)
        nl = len(self.stack.layers)
        dtok = self.stack.backward(dxcls, dxcls_b, state["saved"], B, S, None, grads, full=full,   # [B,H]: the last block runs class-row-only
                                   on_layer_done=lambda i: self._ready(nl - i))
        if full:
            # tokens = [cls + pos[0] | patch_proj + pos[1:]]  (timm VisionTransformer._pos_embed)
            d3 = dtok.view(B, S, H)
            if id(v.pos_embed) in grads:
                ops.batch_sum(d3, grads[id(v.pos_embed)])
            if id(v.cls_token) in grads:
                ops.batch_sum(ops.gather_rows(d3), grads[id(v.cls_token)])
            pw, pb = v.patch_embed.proj.weight, v.patch_embed.proj.bias
            if id(pw) in grads or id(pb) in grads:
                linear_wgrad(ops.slice_rows_cast_bf16(d3, 1, S), state["patches"], [pw], [pb], grads)


# =========================================================================================================
class BertTower(_Tower):
    def __init__(self, bert, head: str, head_modules: dict):
        """bert: HF-shaped BertModel tree (embeddings.{word,position,token_type}_embeddings, embeddings.LayerNorm,
        encoder.layer[i].attention.self.{query,key,value}, attention.output.{dense,LayerNorm}, intermediate.dense,
        output.{dense,LayerNorm}); query/value may be LoRA wrappers exposing .w/.w_a/.w_b.
        head="mlm": head_modules = {transform_dense, transform_ln, decoder};  head="mean": {proj}."""
        self.bert, self.head_kind, self.hm = bert, head, head_modules
        emb = bert.embeddings
        H = emb.word_embeddings.weight.shape[1]
        layers = []
        for layer in bert.encoder.layer:
            sa = layer.attention.self
            q, k, vv = sa.query, sa.key, sa.value
            has_lora = hasattr(q, "w_a")
            if has_lora != hasattr(vv, "w_a"):
                raise NotSupportedYet("LoRA must wrap both query and value of a layer")
            qb, vb = (q.w, vv.w) if has_lora else (q, vv)
            lora = LoraParams(q.w_a.weight, q.w_b.weight, vv.w_a.weight, vv.w_b.weight) if has_lora else None
            ao, oo = layer.attention.output, layer.output
            layers.append(LayerSpec([qb.weight, k.weight, vb.weight], [qb.bias, k.bias, vb.bias], ao.dense.weight, ao.dense.bias,
                                    layer.intermediate.dense.weight, layer.intermediate.dense.bias, oo.dense.weight, oo.dense.bias,
                                    ao.LayerNorm.weight, ao.LayerNorm.bias, oo.LayerNorm.weight, oo.LayerNorm.bias, lora))
        heads = getattr(getattr(bert, "config", None), "num_attention_heads", None) or getattr(bert.encoder.layer[0].attention.self, "num_attention_heads", None) or H // 64
        self.H = H
        self.stack = TransformerStack(layers, H, int(heads), pre_ln=False, eps=float(bert.encoder.layer[0].output.LayerNorm.eps))
        self._head_key, self._head_cache = None, None
        cfg = getattr(bert, "config", None)
        self.p_hidden = float(getattr(cfg, "hidden_dropout_prob", 0.1))
        self.p_attn = float(getattr(cfg, "attention_probs_dropout_prob", 0.1))

    def trainable_params(self):
        ps = []
        for L in self.stack.layers:
            if L.lora is not None:
                ps += L.lora.tensors()
        if self.head_kind == "mlm":
            ps += [self.hm["decoder"].weight, self.hm["decoder"].bias]
        else:
            ps += [self.hm["proj"].weight, self.hm["proj"].bias]
        return ps + self.stack.base_params() + self._frozen_extra()

    def _frozen_extra(self):
        emb = self.bert.embeddings
        ps = [emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight, emb.LayerNorm.weight,
              emb.LayerNorm.bias]
        if self.head_kind == "mlm":
            ps += [self.hm["transform_dense"].weight, self.hm["transform_dense"].bias, self.hm["transform_ln"].weight,
                   self.hm["transform_ln"].bias]
        return ps

    def _full(self):
        return self.stack.full_mode() or any(p.requires_grad for p in self._frozen_extra())

    def grad_groups(self):
        emb = self.bert.embeddings
        if self.head_kind == "mlm":
            head = [self.hm["decoder"].weight, self.hm["decoder"].bias, self.hm["transform_dense"].weight, self.hm["transform_dense"].bias,
                    self.hm["transform_ln"].weight, self.hm["transform_ln"].bias]
        else:
            head = [self.hm["proj"].weight, self.hm["proj"].bias]
        return self._stack_groups(head, [emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight,
                                         emb.LayerNorm.weight, emb.LayerNorm.bias])

    def _head_images(self):
        if self.head_kind != "mlm":
            return None
        td = self.hm["transform_dense"]
        key = (td.weight._version, td.weight.data_ptr())
        if key != self._head_key or td.weight.requires_grad:
            w = _f32c(td.weight)
            self._head_cache = (ops.cast_bf16(w), ops.cast_transpose_bf16(w))
            self._head_key = key
        return self._head_cache

    def _forward(self, inputs, save):
        ids, token_type, attn_mask = inputs
        emb = self.bert.embeddings
        if ids.dim() != 2:
            raise ValueError("token ids must be [B,S]")
        B, S = ids.shape
        if S > 256:
            raise NotSupportedYet("sequence length > 256")
        H = self.H
        dev = ids.device
        full = save and self._full()
        self.stack.refresh()
        self.stack.pack_lora()
        M = B * S
        ids = ids.detach().to(torch.int64).contiguous()
        vocab = emb.word_embeddings.weight.shape[0]
        tt = None if token_type is None else token_type.detach().to(torch.int64).contiguous()
        key_mask = None if attn_mask is None else attn_mask.detach().to(torch.int32).contiguous()
        e = torch.empty((M, H), dtype=F32, device=dev)
        ops.bert_embed(ids, tt, _f32c(emb.word_embeddings.weight), _f32c(emb.position_embeddings.weight),
                       _f32c(emb.token_type_embeddings.weight), e)
        x_f32, x_bf16 = torch.empty((M, H), dtype=F32, device=dev), torch.empty((M, H), dtype=BF16, device=dev)
        a0 = self.stack.lora_a(0)
        t0 = torch.empty((M, 8), dtype=BF16, device=dev) if a0 is not None else None
        # HF BERT dropout (train mode only): embeddings, attention probabilities, both dense outputs of every layer
        drop, d_emb = None, None
        if self.training and (self.p_hidden > 0 or self.p_attn > 0):
            base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())  # CPU generator: reproducible under torch.manual_seed, no device sync
            drop = (self.p_hidden, self.p_attn, base)
            d_emb = ops.Drop(self.p_hidden, ops.derive_seed(base, 255, 3))
        st_e = torch.empty((M, 2), dtype=F32, device=dev) if full else None
        f8 = self.stack.fp8
        if f8 is not None and "qkv_in" not in f8[0]:
            f8 = None   # fp8 site selection (MLP pair only): QKV takes the bf16 image of the embeddings
        x_fp8 = torch.empty((M, H), dtype=ops.FP8, device=dev) if f8 is not None else None
        ops.layernorm_fwd(e, _f32c(emb.LayerNorm.weight), _f32c(emb.LayerNorm.bias), float(emb.LayerNorm.eps), y_bf16=x_bf16, y_f32=x_f32,
                          stats=st_e, lora_a=a0, t_out=t0, drop=d_emb, y_fp8=x_fp8, fp8_scale=f8[0]["qkv_in"] if f8 is not None else 0.0)
        x_f32, x_bf16, saved = self.stack.forward(x_f32, x_bf16, t0, B, S, key_mask, save, drop=drop, full=full, x_fp8=x_fp8)
        state = dict(saved=saved, B=B, S=S, key_mask=key_mask, full=full) if save else None
        if full:
            state.update(e=e, st_e=st_e, d_emb=d_emb, ids=ids, tt=tt, x_top=x_bf16)
        if self.head_kind == "mlm":
            wt, _ = self._head_images()
            td, tln, dec = self.hm["transform_dense"], self.hm["transform_ln"], self.hm["decoder"]
            hpre = torch.empty((M, H), dtype=BF16, device=dev)
            g = torch.empty((M, H), dtype=F32, device=dev)
            ops.gemm_nt(x_bf16, wt, bias=_f32c(td.bias), act=ops.ACT_GELU, out_pre=hpre, out_f32=g)
            hln = torch.empty((M, H), dtype=BF16, device=dev)
            st = torch.empty((M, 2), dtype=F32, device=dev)
            ops.layernorm_fwd(g, _f32c(tln.weight), _f32c(tln.bias), float(tln.eps), y_bf16=hln, stats=st)
            C = dec.weight.shape[0]
            logits = torch.empty((M, C), dtype=BF16, device=dev)
            ops.gemm_nt(hln, ops.cast_bf16(_f32c(dec.weight)), bias=_f32c(dec.bias), out_bf16=logits)
            out = ops.softmax_mean_fwd(logits, B, S)
            if save:
                state.update(hpre=hpre, g=g, st=st, hln=hln, logits=logits)
        else:
            proj = self.hm["proj"]
            mean = ops.token_mean_fwd(x_f32.view(B, S, H))
            out = torch.empty((B, proj.weight.shape[0]), dtype=F32, device=dev)
            ops.gemm_nt(mean, ops.cast_bf16(_f32c(proj.weight)), bias=_f32c(proj.bias), out_f32=out)
            if save:
                state.update(mean=mean)
        return out, state

    def _backward(self, dout, state, grads):
        B, S, H = state["B"], state["S"], self.H
        M = B * S
        dev = dout.device
        if self.head_kind == "mlm":
            _, wt_t = self._head_images()
            tln, dec = self.hm["transform_ln"], self.hm["decoder"]
            dlogits = ops.softmax_mean_bwd(state["logits"], dout, B, S)
            dhln = dense_head_backward(dlogits, state["hln"], dec.weight, dec.bias, grads, out_bf16=True)
            full = state["full"]
            pg = dict(dgamma=grads[id(tln.weight)].view(-1), dbeta=grads[id(tln.bias)].view(-1)) if full and id(tln.weight) in grads else {}
            dg = torch.empty((M, H), dtype=BF16, device=dev)
            ops.layernorm_bwd(dhln, state["g"], state["st"], _f32c(tln.weight), dx_bf16=dg, **pg)
            dhpre = ops.gelu_bwd(dg, state["hpre"])
            if full:
                td = self.hm["transform_dense"]
                linear_wgrad(dhpre, state["x_top"], [td.weight], [td.bias], grads)
            dx = torch.empty((M, H), dtype=F32, device=dev)
            ops.gemm_nt(dhpre, wt_t, out_f32=dx)
        else:
            proj = self.hm["proj"]
            dmean = dense_head_backward(dout, state["mean"], proj.weight, proj.bias, grads, out_bf16=False)
            dx = ops.token_mean_bwd(dmean, S).view(M, H)
        full = state["full"]
        self._ready(0)
        nl = len(self.stack.layers)
        dx0 = self.stack.backward(dx, None, state["saved"], B, S, state["key_mask"], grads, full=full, on_layer_done=lambda i: self._ready(nl - i))
        if full:
            # x0 = dropout(LayerNorm(word[ids] + position[s] + token_type[tt]))   (HF BertEmbeddings)
            emb = self.bert.embeddings
            lw, lb = emb.LayerNorm.weight, emb.LayerNorm.bias
            if state["d_emb"] is not None and state["d_emb"].thr16 > 0:
                dx0 = ops.dropout_apply(dx0, state["d_emb"])  # gradient w.r.t. the LayerNorm output
            pg = dict(dgamma=grads[id(lw)].view(-1), dbeta=grads[id(lb)].view(-1)) if id(lw) in grads else {}
            tabs = [emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight]
            need_de = any(id(t) in grads for t in tabs)
            if need_de or pg:
                de = torch.empty((M, H), dtype=F32, device=dev)
                ops.layernorm_bwd(dx0, state["e"], state["st_e"], _f32c(lw), dx_f32=de, **pg)   # + d(gamma), d(beta) in the same pass
            if need_de:
                if id(tabs[1]) in grads:
                    ops.batch_sum(de.view(B, S * H), grads[id(tabs[1])][:S])
                ops.bert_embed_bwd(state["ids"].view(-1), None if state["tt"] is None else state["tt"].view(-1), de,
                                   grads.get(id(tabs[0])), grads.get(id(tabs[2])))
