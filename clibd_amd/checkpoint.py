"""Checkpoint I/O compatible with the reference ("next" row SURVEY §8f-3).

* `model.state_dict()` keys equal the reference's (asserted in the test-suite), so `last.pth` / `best.pth` written by
  `scripts/train_cl.py:292-318` load with a plain `load_state_dict` — after the legacy class-name renames the reference
  applies itself (`update_checkpoint_param_names`, util/util.py:924-948) and the DDP `module.` prefix strip.
* The reference saves the model only (no optimizer / scheduler state => no true resume); `save_training_state`
  / `load_training_state` add the fused AdamW moments, step count and scheduler state.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

# substring renames of checkpoints written by older versions of the reference (util/util.py:930-937)
_LEGACY_NAMES = {
    "LoRA_barcode_bert": "CLIBDDNAEncoder",
    "lora_barcode_bert": "base_dna_encoder",
    "LoRA_ViT_timm": "CLIBDImageEncoder",
    "lora_vit": "base_image_encoder",
    "LoRA_bert": "CLIBDLanguageEncoder",
    "lora_bert": "base_language_encoder",
}


def update_checkpoint_param_names(checkpoint: dict) -> dict:
    out = {}
    for name, tensor in checkpoint.items():
        new = name
        for old, repl in _LEGACY_NAMES.items():
            if old in new:
                new = new.replace(old, repl)
        out[new] = tensor
    return out


def remove_module_from_state_dict(state_dict: dict) -> dict:
    """strip the `module.` prefix DistributedDataParallel adds (train_cl.py:204 saves model.state_dict() of the DDP wrapper)"""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def drop_hf_buffer_keys(state_dict: dict) -> dict:
    """The reference pins transformers==4.29.2 (requirements.txt:6), whose BertEmbeddings registers `position_ids` as a
    PERSISTENT buffer (`token_type_ids` is registered persistent=False there and is not written): best.pth / last.pth of both
    BERT towers carry `...embeddings.position_ids` entries that are not parameters (an arange, recomputed here).  They are dropped
    before the strict load, as load_pre_trained_bioscan_bert does for the BarcodeBERT pretrain file; a `token_type_ids` entry —
    which other transformers versions do persist — is dropped as well if present."""
    return {k: v for k, v in state_dict.items() if not (k.endswith("embeddings.position_ids") or k.endswith("embeddings.token_type_ids"))}


def _normalise(state_dict: dict) -> dict:
    return drop_hf_buffer_keys(update_checkpoint_param_names(remove_module_from_state_dict(state_dict)))


def handle_local_ckpt_path(args) -> str:
    mc = args.model_config
    if hasattr(mc, "ckpt_path"):
        path = mc.ckpt_path
    else:
        path = f"{args.project_root_path}/ckpt/bioscan_clip/{args.version}/{mc.dataset}/{mc.model_output_name}/best.pth"
    for name in ("best.pth", "last.pth"):
        cand = os.path.join(path, name)
        if os.path.exists(cand):
            return cand
    return path


def load_reference_checkpoint(model: torch.nn.Module, path: str, strict: bool = True):
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return model.load_state_dict(_normalise(ckpt), strict=strict)


def initialize_model_and_load_from_checkpoint(args, device=None):
    """simple_clip.py:248-285 without the Hugging Face hub fallback (no network): local checkpoints only."""
    from .model.simple_clip import load_clip_model

    model = load_clip_model(args, device)
    if hasattr(args.model_config, "load_ckpt") and args.model_config.load_ckpt is False:
        return model
    path = handle_local_ckpt_path(args)
    if not os.path.exists(path):
        raise ValueError("Neither the local checkpoint nor the huggingface checkpoint was found. Please check the config file")
    load_reference_checkpoint(model, path)
    return model


def export_reference_state_dict(model: torch.nn.Module) -> dict:
    """`model.state_dict()` in the form the REFERENCE loads strictly (simple_clip.py:263/279 `model.load_state_dict(checkpoint)`):
    under its pinned transformers==4.29.2 `BertEmbeddings.position_ids` is a persistent buffer, so both BERT towers' state
    dicts carry `...embeddings.position_ids` = arange(max_position_embeddings)[None] (int64); the modules here recompute it and
    do not store it.  (`token_type_ids` is registered persistent=False there and is not part of a checkpoint.)"""
    sd = dict(model.state_dict())
    for key in list(sd.keys()):
        if key.endswith("embeddings.position_embeddings.weight"):
            n = sd[key].shape[0]
            sd[key[: -len("position_embeddings.weight")] + "position_ids"] = torch.arange(n, dtype=torch.int64).unsqueeze(0)
    return sd


def save_reference_checkpoint(model: torch.nn.Module, path: str) -> None:
    """What train_cl.py:292-318 writes (`torch.save(model.state_dict(), best.pth / last.pth)`), loadable by the reference."""
    torch.save({k: v.detach().cpu() for k, v in export_reference_state_dict(model).items()}, path)


def _optimizer_layout(model: torch.nn.Module, optimizer) -> list:
    """[(parameter name, flat offset, numel)] of the fused optimizer's bucket: its order depends on the towers present, on
    fix_temperature and on the backward order (train._backward_order), so the raw flat moments alone do not say which
    parameter a slice belongs to."""
    names = {id(p): n for n, p in model.named_parameters()}
    return [(names.get(id(p), f"<unnamed {i}>"), int(off), int(p.numel()))
            for i, (p, off) in enumerate(zip(optimizer.param_groups[0]["params"], optimizer._offsets))]


def save_training_state(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, epoch: Optional[int] = None):
    state = {"model": model.state_dict(), "epoch": epoch}
    if hasattr(model, "numerics"):
        state["numerics"] = model.numerics()   # which arithmetic produced these weights (engine.NUMERICS_CHOICES + fp8-forward flag)
    if optimizer is not None and hasattr(optimizer, "exp_avg"):
        state["optimizer"] = {"exp_avg": optimizer.exp_avg.detach().cpu(), "exp_avg_sq": optimizer.exp_avg_sq.detach().cpu(),
                              "step_count": optimizer.step_count, "layout": _optimizer_layout(model, optimizer),
                              "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups]}
    if scheduler is not None:
        state["scheduler"] = scheduler.state_dict()
    torch.save(state, path)


def load_training_state(path: str, model: torch.nn.Module, optimizer=None, scheduler=None):
    state = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(_normalise(state["model"]))
    if optimizer is not None and "optimizer" in state:
        o = state["optimizer"]
        cur = _optimizer_layout(model, optimizer)
        saved = o.get("layout")
        if saved is None:
            # a state written before the layout was recorded: only a bucket of the same size can be taken as is
            if o["exp_avg"].numel() != optimizer.exp_avg.numel():
                raise ValueError("training state has no optimizer layout and its moment buffers do not match this optimizer "
                                 f"({o['exp_avg'].numel()} vs {optimizer.exp_avg.numel()} values): it was saved with a different set of "
                                 "towers / fix_temperature; re-create it or start the moments from zero")
            optimizer.exp_avg.copy_(o["exp_avg"])
            optimizer.exp_avg_sq.copy_(o["exp_avg_sq"])
        elif [tuple(e) for e in saved] == cur:
            optimizer.exp_avg.copy_(o["exp_avg"])
            optimizer.exp_avg_sq.copy_(o["exp_avg_sq"])
        else:
            # same parameters in another order / a subset: move every moment slice to where its parameter lives now
            where = {n: (off, k) for n, off, k in (tuple(e) for e in saved)}
            missing = [n for n, _, k in cur if n not in where or where[n][1] != k]
            if missing:
                raise ValueError(f"training state lacks optimizer moments for {len(missing)} parameters (first: {missing[0]}); "
                                 "it was saved with a different set of trainable parameters")
            optimizer.exp_avg.zero_()
            optimizer.exp_avg_sq.zero_()
            for n, off, k in cur:
                s_off = where[n][0]
                optimizer.exp_avg[off : off + k].copy_(o["exp_avg"][s_off : s_off + k])
                optimizer.exp_avg_sq[off : off + k].copy_(o["exp_avg_sq"][s_off : s_off + k])
        optimizer.step_count = o["step_count"]
        for g, saved_g in zip(optimizer.param_groups, o["param_groups"]):
            g.update(saved_g)
        # the flat parameter bucket aliases the parameters, which load_state_dict just overwrote in place
    if scheduler is not None and "scheduler" in state:
        scheduler.load_state_dict(state["scheduler"])
    return state.get("epoch")
