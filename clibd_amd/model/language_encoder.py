"""Text tower: drop-in for `bioscanclip.model.language_encoder` (reference model/language_encoder.py:12-89).

`CLIBDLanguageEncoder(model, r, num_classes=0, lora_layer=None)`: BERT-small (`prajjwal1/bert-small`: L=4, H=512,
A=8, FF=2048) with rank-4 adapters on query/value and `proj(last_hidden_state.mean(dim=1))` — the mean runs over
all positions including padding, exactly as the reference does.  `model` is the container from
`clibd_amd.model.dna_encoder.BertModel` or a real transformers.BertModel.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..towers import BertTower
from .dna_encoder import BertConfigLite, BertModel, add_bert_lora

BERT_SMALL = dict(vocab_size=30522, hidden_size=512, num_hidden_layers=4, num_attention_heads=8, intermediate_size=2048)


def load_pre_trained_bert(language_model_name: str = "prajjwal1/bert-small"):
    """(tokenizer, model).  Uses a locally cached HF checkpoint when there is one (no network here); otherwise a
    randomly initialised BERT-small-shaped container and no tokenizer."""
    try:  # pragma: no cover - needs local weights
        from transformers import AutoTokenizer
        from transformers import BertModel as HFBertModel

        tok = AutoTokenizer.from_pretrained(language_model_name, local_files_only=True)
        model = HFBertModel.from_pretrained(language_model_name, local_files_only=True)
    except Exception:
        tok, model = None, BertModel(BertConfigLite(**BERT_SMALL))
    for p in model.parameters():
        p.requires_grad = False
    return tok, model


class CLIBDLanguageEncoder(nn.Module):
    def __init__(self, model, r: int, num_classes: int = 0, lora_layer=None):
        super().__init__()
        assert r > 0
        self.lora_layer = lora_layer if lora_layer is not None else list(range(len(model.encoder.layer)))
        self.w_As, self.w_Bs = [], []
        for p in model.parameters():
            p.requires_grad = False
        add_bert_lora(model.encoder.layer, r, self.lora_layer, self.w_As, self.w_Bs)
        self.reset_parameters()
        self.base_language_encoder = model
        if num_classes > 0:
            dense = self.base_language_encoder.pooler.dense
            self.proj = nn.Linear(dense.out_features, num_classes).to(dense.weight.device)
        self._tower = None

    def reset_parameters(self) -> None:
        for w_A in self.w_As:
            nn.init.kaiming_uniform_(w_A.weight, a=math.sqrt(5))
        for w_B in self.w_Bs:
            nn.init.zeros_(w_B.weight)

    def tower(self) -> BertTower:
        if self._tower is None:
            self._tower = BertTower(self.base_language_encoder, "mean", dict(proj=self.proj))
        return self._tower

    def forward(self, x) -> torch.Tensor:
        tw = self.tower()
        tw.training = self.training
        return tw(x["input_ids"], x.get("token_type_ids"), x.get("attention_mask"))
