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
    """The reference pins transformers==4.29.2 (requirements.txt:6), whose BertEmbeddings registers `position_ids` (and
    `token_type_ids`) as PERSISTENT buffers: best.pth / last.pth of both BERT towers carry
    `...embeddings.position_ids` / `...embeddings.token_type_ids` entries that are not parameters (arange / zeros, recomputed
    here).  They are dropped before the strict load, as load_pre_trained_bioscan_bert does for the BarcodeBERT pretrain file."""
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


def save_training_state(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, epoch: Optional[int] = None):
    state = {"model": model.state_dict(), "epoch": epoch}
    if optimizer is not None and hasattr(optimizer, "exp_avg"):
        state["optimizer"] = {"exp_avg": optimizer.exp_avg.detach().cpu(), "exp_avg_sq": optimizer.exp_avg_sq.detach().cpu(),
                              "step_count": optimizer.step_count, "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                                   for g in optimizer.param_groups]}
    if scheduler is not None:
        state["scheduler"] = scheduler.state_dict()
    torch.save(state, path)


def load_training_state(path: str, model: torch.nn.Module, optimizer=None, scheduler=None):
    state = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(_normalise(state["model"]))
    if optimizer is not None and "optimizer" in state:
        o = state["optimizer"]
        optimizer.exp_avg.copy_(o["exp_avg"])
        optimizer.exp_avg_sq.copy_(o["exp_avg_sq"])
        optimizer.step_count = o["step_count"]
        for g, saved in zip(optimizer.param_groups, o["param_groups"]):
            g.update(saved)
        # the flat parameter bucket aliases the parameters, which load_state_dict just overwrote in place
    if scheduler is not None and "scheduler" in state:
        scheduler.load_state_dict(state["scheduler"])
    return state.get("epoch")
