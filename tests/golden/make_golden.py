"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (authoring container only).

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.pt

The reference (bioscan-ml/clibd, /root/reference) is pure Python; its third-party imports that are not
installed here (timm, loratorch, torchtext, faiss, clip, open_clip, h5py, wandb, omegaconf, hydra, umap, ...)
are satisfied with EMPTY stub modules — none of them is on the arithmetic path that is exercised:
  * losses: bioscanclip.model.loss_func.{ContrastiveLoss, ClipLoss}       (imported as-is)
  * towers: bioscanclip.model.{dna_encoder.CLIBDDNAEncoder, language_encoder.CLIBDLanguageEncoder} wrapping
            HF transformers BertForMaskedLM / BertModel (the reference's own dependency, installed here);
            bioscanclip.model.image_encoder.CLIBDImageEncoder wrapping the oracle's timm-shaped ViT body
            (timm itself is absent; the body is cross-checked against transformers.ViTModel below);
  * bioscanclip.model.simple_clip.SimpleCLIP for the full step.
Only DATA (inputs, weights, expected outputs/gradients) is written; no reference source is copied.
"""
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import transformers  # noqa: E402  (must be imported before the stubs are installed)
from transformers import BertConfig, BertForMaskedLM, BertModel  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        return _Anything()


def install_stubs():
    _stub("loratorch")
    _stub("loratorch.layers", MultiheadAttention=_Anything)
    _stub("timm", create_model=_Anything())
    _stub("timm.models")
    _stub("timm.models.vision_transformer", VisionTransformer=object)
    _stub("torchtext")
    _stub("torchtext.vocab", build_vocab_from_iterator=_Anything(), vocab=_Anything())
    for name in ("faiss", "clip", "h5py", "wandb", "umap", "plotly", "plotly.express", "plotly.graph_objects", "hydra", "seaborn"):
        _stub(name)
    _stub("open_clip", get_tokenizer=_Anything(), create_model_and_transforms=_Anything())
    _stub("omegaconf", OmegaConf=_Anything, DictConfig=dict, open_dict=_Anything())
    for name in ("sklearn", "sklearn.preprocessing", "sklearn.metrics", "sklearn.neighbors", "sklearn.linear_model"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        _stub("matplotlib")
        _stub("matplotlib.pyplot")
    for name in ("torchvision", "torchvision.transforms", "PIL", "PIL.Image"):
        try:
            __import__(name)
        except Exception:
            _stub(name, transforms=_Anything(), Image=_Anything())


def import_reference():
    install_stubs()
    sys.path.insert(0, REF)
    from bioscanclip.model import loss_func  # noqa

    mods = {"loss_func": loss_func}
    for name in ("dna_encoder", "language_encoder", "image_encoder", "simple_clip"):
        try:
            mods[name] = __import__(f"bioscanclip.model.{name}", fromlist=["x"])
        except Exception as e:  # pragma: no cover
            print(f"[make_golden] could not import bioscanclip.model.{name}: {e!r}")
            raise
    return mods


def randomize_(module: nn.Module, gen: torch.Generator, std=0.05):
    """Deterministic non-trivial weights everywhere (LoRA B matrices included, so adapters are exercised)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() == 1 and ("LayerNorm.weight" in name or "norm" in name and name.endswith("weight")):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen))
            elif p.dim() == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(std * torch.randn(p.shape, generator=gen))


def grads_of(module, loss):
    params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
    gs = torch.autograd.grad(loss, [p for _, p in params], allow_unused=True)
    return {n: (g.clone() if g is not None else torch.zeros_like(p)) for (n, p), g in zip(params, gs)}


def main():
    ref = import_reference()
    from oracle import clibd_oracle as O

    torch.manual_seed(0)
    out = {}

    # ---------------------------------------------------------------- G1: losses
    lf = ref["loss_func"]
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    cases = []
    g = torch.Generator().manual_seed(0)
    for N, D, nmod, dup, bind_to, no_it in [(8, 128, 2, False, None, False), (32, 768, 2, False, None, False), (32, 128, 2, True, None, False),
                                             (64, 128, 3, False, None, False), (32, 128, 3, True, "dna", False), (16, 128, 3, True, None, True),
                                             (32, 128, 3, False, "image", False)]:
        feats = [torch.randn(N, D, generator=g) for _ in range(nmod)] + [None] * (3 - nmod)
        feats = [f.requires_grad_(True) if f is not None else None for f in feats]
        labels = torch.arange(N) // 2 if dup else torch.arange(N)
        ls = torch.tensor(2.6592600, requires_grad=True)  # log(1/0.07)
        crit_c = lf.ContrastiveLoss(criterion=nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
        crit_k = lf.ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=nn.CrossEntropyLoss(),
                             bind_to=bind_to, no_image_text_loss=no_it)
        loss_k = crit_k(feats[0], feats[1], feats[2], labels, ls.exp())
        present = [f for f in feats if f is not None]
        gk = torch.autograd.grad(loss_k, present + [ls])
        rec = {"N": N, "labels": labels, "bind_to": bind_to, "no_image_text_loss": no_it,
               "features": [None if f is None else f.detach().clone() for f in feats], "log_scale": ls.detach().clone(),
               "clip_loss": loss_k.detach().clone(), "clip_grads": [x.clone() for x in gk]}
        if bind_to is None and not no_it:
            loss_c = crit_c(feats[0], feats[1], feats[2], labels, ls.exp())
            rec["contrastive_loss"] = loss_c.detach().clone()
        cases.append(rec)
    out["loss_cases"] = cases
    # SURVEY §8c known answers (seed 0, randn(32,768) twice)
    torch.manual_seed(0)
    a, b = torch.randn(32, 768), torch.randn(32, 768)
    crit_c = lf.ContrastiveLoss(criterion=nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    out["known_answers"] = {
        "a": a, "b": b,
        "unique": crit_c(a, b, None, torch.arange(32), 1 / 0.07).detach().clone(),
        "dup": crit_c(a, b, None, torch.arange(32) // 2, 1 / 0.07).detach().clone(),
    }
    torch.save(out, os.path.join(HERE, "loss_golden.pt"))
    print("[make_golden] losses:", [float(c["clip_loss"]) for c in cases], float(out["known_answers"]["unique"]), float(out["known_answers"]["dup"]))

    # ---------------------------------------------------------------- G2: DNA tower (tiny BERT-MLM, dh = 64)
    tiny = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    gen = torch.Generator().manual_seed(1)
    hf = BertForMaskedLM(BertConfig(vocab_size=1027, output_hidden_states=True, **tiny))
    enc = ref["dna_encoder"].CLIBDDNAEncoder(model=hf, r=4, num_classes=128)
    randomize_(enc, gen)
    enc.eval()
    ids = torch.cat([torch.zeros(4, 1, dtype=torch.long), torch.randint(3, 1027, (4, 132), generator=gen)], dim=1)
    y = enc(ids)
    w = torch.randn(y.shape, generator=gen)
    dna = {"config": tiny, "state_dict": {k: v.detach().clone() for k, v in enc.state_dict().items()}, "ids": ids,
           "out": y.detach().clone(), "cot": w, "grads": grads_of(enc, (y * w).sum())}
    torch.save(dna, os.path.join(HERE, "dna_tiny_golden.pt"))
    print("[make_golden] dna out", tuple(y.shape), float(y.sum()), "trainable", len(dna["grads"]))

    # ---------------------------------------------------------------- G3: text tower (tiny BertModel with padding mask)
    gen = torch.Generator().manual_seed(2)
    hfb = BertModel(BertConfig(vocab_size=1000, **tiny))
    tenc = ref["language_encoder"].CLIBDLanguageEncoder(model=hfb, r=4, num_classes=128)
    randomize_(tenc, gen)
    tenc.eval()
    tids = torch.randint(0, 1000, (4, 20), generator=gen)
    lens = torch.tensor([20, 6, 13, 9])
    am = (torch.arange(20)[None, :] < lens[:, None]).long()
    tin = {"input_ids": tids, "token_type_ids": torch.zeros_like(tids), "attention_mask": am}
    ty = tenc(tin)
    tw = torch.randn(ty.shape, generator=gen)
    txt = {"config": tiny, "vocab": 1000, "state_dict": {k: v.detach().clone() for k, v in tenc.state_dict().items()}, "inputs": tin,
           "out": ty.detach().clone(), "cot": tw, "grads": grads_of(tenc, (ty * tw).sum())}
    torch.save(txt, os.path.join(HERE, "text_tiny_golden.pt"))
    print("[make_golden] text out", tuple(ty.shape), float(ty.abs().mean()))

    # ---------------------------------------------------------------- G4: image tower = reference wrapper around the restated ViT body
    gen = torch.Generator().manual_seed(3)
    vit = O.VisionTransformer(img_size=224, patch=16, dim=128, depth=2, heads=2, num_classes=10)
    ienc = ref["image_encoder"].CLIBDImageEncoder(vit_model=vit, r=4, num_classes=128)
    randomize_(ienc, gen)
    ienc.eval()
    img_u8 = torch.randint(0, 256, (3, 3, 224, 224), generator=gen, dtype=torch.uint8)
    img = img_u8.float() / 255.0
    iy = ienc(img)
    iw = torch.randn(iy.shape, generator=gen)
    image = {"config": dict(dim=128, depth=2, heads=2), "state_dict": {k: v.detach().clone() for k, v in ienc.state_dict().items()},
             "image_u8": img_u8, "out": iy.detach().clone(), "cot": iw, "grads": grads_of(ienc, (iy * iw).sum())}
    # independent cross-check of the ViT *body* against transformers.ViTModel (same weights, LoRA merged away)
    try:
        from transformers import ViTConfig, ViTModel

        hv = ViTModel(ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, image_size=224,
                                patch_size=16, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True), add_pooling_layer=False).eval()
        body = O.VisionTransformer(img_size=224, patch=16, dim=128, depth=2, heads=2, num_classes=0)
        randomize_(body, torch.Generator().manual_seed(33))
        sd = hv.state_dict()
        new = {}
        new["embeddings.cls_token"] = body.cls_token.data
        new["embeddings.position_embeddings"] = body.pos_embed.data
        new["embeddings.patch_embeddings.projection.weight"] = body.patch_embed.proj.weight.data
        new["embeddings.patch_embeddings.projection.bias"] = body.patch_embed.proj.bias.data
        keys = list(sd.keys())
        lay = "encoder.layer" if any(k.startswith("encoder.layer") for k in keys) else "layers"
        for i, blk in enumerate(body.blocks):
            qw, kw, vw = blk.attn.qkv.weight.data.chunk(3, 0)
            qb, kb, vb = blk.attn.qkv.bias.data.chunk(3, 0)
            cand = {
                "q": [f"{lay}.{i}.attention.attention.query", f"{lay}.{i}.attention.q_proj"],
                "k": [f"{lay}.{i}.attention.attention.key", f"{lay}.{i}.attention.k_proj"],
                "v": [f"{lay}.{i}.attention.attention.value", f"{lay}.{i}.attention.v_proj"],
                "o": [f"{lay}.{i}.attention.output.dense", f"{lay}.{i}.attention.o_proj"],
                "f1": [f"{lay}.{i}.intermediate.dense", f"{lay}.{i}.mlp.fc1", f"{lay}.{i}.mlp.up_proj"],
                "f2": [f"{lay}.{i}.output.dense", f"{lay}.{i}.mlp.fc2", f"{lay}.{i}.mlp.down_proj"],
                "n1": [f"{lay}.{i}.layernorm_before"], "n2": [f"{lay}.{i}.layernorm_after"],
            }
            src = {"q": (qw, qb), "k": (kw, kb), "v": (vw, vb), "o": (blk.attn.proj.weight.data, blk.attn.proj.bias.data),
                   "f1": (blk.mlp.fc1.weight.data, blk.mlp.fc1.bias.data), "f2": (blk.mlp.fc2.weight.data, blk.mlp.fc2.bias.data),
                   "n1": (blk.norm1.weight.data, blk.norm1.bias.data), "n2": (blk.norm2.weight.data, blk.norm2.bias.data)}
            for key, names in cand.items():
                hit = [n for n in names if n + ".weight" in sd]
                assert hit, (key, [k for k in keys if f".{i}." in k])
                new[hit[0] + ".weight"], new[hit[0] + ".bias"] = src[key]
        new["layernorm.weight"], new["layernorm.bias"] = body.norm.weight.data, body.norm.bias.data
        missing = hv.load_state_dict(new, strict=False)
        assert not missing.missing_keys, missing
        with torch.no_grad():
            hf_out = hv(pixel_values=img).last_hidden_state
            my_out = body.forward_features(img)
        err = (hf_out - my_out).abs().max().item()
        print(f"[make_golden] ViT body vs transformers.ViTModel: max abs diff {err:.3e}")
        assert err < 2e-4, err
        image["vit_body_crosscheck"] = {"max_abs_diff": err, "checked_against": f"transformers.ViTModel {transformers.__version__}"}
    except ImportError as e:  # pragma: no cover
        print("[make_golden] transformers.ViTModel unavailable:", e)
    torch.save(image, os.path.join(HERE, "image_tiny_golden.pt"))
    print("[make_golden] image out", tuple(iy.shape), float(iy.abs().mean()))

    # ---------------------------------------------------------------- G5: SimpleCLIP + ClipLoss full step (b=8)
    # the three towers above (same weights: no second copy of the state dicts is stored)
    gen = torch.Generator().manual_seed(4)
    SimpleCLIP = ref["simple_clip"].SimpleCLIP
    model = SimpleCLIP(image_encoder=ienc, dna_encoder=enc, language_encoder=tenc)
    with torch.no_grad():
        model.logit_scale.fill_(2.6592600)
    model.eval()
    B = 8
    img_u8 = torch.randint(0, 256, (B, 3, 224, 224), generator=gen, dtype=torch.uint8)
    img = img_u8.float() / 255.0
    ids = torch.cat([torch.zeros(B, 1, dtype=torch.long), torch.randint(3, 1027, (B, 132), generator=gen)], dim=1)
    tids = torch.randint(0, 1000, (B, 20), generator=gen)
    lens = torch.randint(6, 21, (B,), generator=gen)
    tin = {"input_ids": tids, "token_type_ids": torch.zeros_like(tids), "attention_mask": (torch.arange(20)[None, :] < lens[:, None]).long()}
    labels = torch.tensor([0, 1, 2, 3, 3, 5, 6, 0])
    step = {"logit_scale": model.logit_scale.detach().clone(), "image_u8": img_u8, "dna": ids, "text": tin, "labels": labels,
            "state_dict_keys": list(model.state_dict().keys())}
    for tag, use_text in (("id", False), ("idt", True)):
        io, do_, to, scale, _ = model(img, ids, tin)
        crit = lf.ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=nn.CrossEntropyLoss())
        loss = crit(io, do_, to if use_text else None, labels, scale)
        step[f"loss_{tag}"] = loss.detach().clone()
        step[f"grads_{tag}"] = grads_of(model, loss)
        step[f"features_{tag}"] = [io.detach().clone(), do_.detach().clone(), to.detach().clone()]
        print(f"[make_golden] step {tag}: loss {float(loss):.6f}")
    torch.save(step, os.path.join(HERE, "step_tiny_golden.pt"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f"  {f}: {os.path.getsize(os.path.join(HERE, f))/1e6:.2f} MB")


if __name__ == "__main__":
    main()
