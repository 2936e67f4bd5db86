"""DNA tower: drop-in for `bioscanclip.model.dna_encoder` (reference model/dna_encoder.py:15-137).

`CLIBDDNAEncoder(model, r, num_classes=0, lora_layer=None)`: BarcodeBERT (a BertForMaskedLM over the 5-mer
vocabulary) with rank-4 adapters on query/value, the MLM decoder replaced by Linear(H, num_classes) and
forward = logits.softmax(-1).mean(1).  `model` is either the parameter container below or a real
transformers.BertForMaskedLM: only parameter attributes are read; the arithmetic runs on the HIP engine.
"""
from __future__ import annotations

import math
from itertools import product

import torch
import torch.nn as nn

from ..towers import BertTower
from .image_encoder import _ParamOnly


# ---------------------------------------------------------------------------------------------------------
# HF-shaped parameter containers (names as in transformers.models.bert.modeling_bert)
# ---------------------------------------------------------------------------------------------------------
class BertConfigLite:
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, **unused):
        self.vocab_size, self.hidden_size, self.num_hidden_layers = vocab_size, hidden_size, num_hidden_layers
        self.num_attention_heads, self.intermediate_size = num_attention_heads, intermediate_size
        self.max_position_embeddings, self.type_vocab_size, self.layer_norm_eps = max_position_embeddings, type_vocab_size, layer_norm_eps
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob


class _Embeddings(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _SelfAttention(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.num_attention_heads = c.num_attention_heads
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)


class _DenseLN(_ParamOnly):
    def __init__(self, din, dout, eps):
        super().__init__()
        self.dense = nn.Linear(din, dout)
        self.LayerNorm = nn.LayerNorm(dout, eps=eps)


class _AttentionBlock(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _DenseLN(c.hidden_size, c.hidden_size, c.layer_norm_eps)


class _Intermediate(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)


class _Layer(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.attention = _AttentionBlock(c)
        self.intermediate = _Intermediate(c)
        self.output = _DenseLN(c.intermediate_size, c.hidden_size, c.layer_norm_eps)


class _Encoder(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class _Pooler(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)


def _bert_init(module: nn.Module, std=0.02):
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=std)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, std=std)


class BertModel(_ParamOnly):
    def __init__(self, config: BertConfigLite, add_pooling_layer: bool = True):
        super().__init__()
        self.config = config
        self.embeddings = _Embeddings(config)
        self.encoder = _Encoder(config)
        if add_pooling_layer:
            self.pooler = _Pooler(config)
        _bert_init(self)


class _Transform(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _Predictions(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.transform = _Transform(c)
        self.decoder = nn.Linear(c.hidden_size, c.vocab_size)
        self.bias = nn.Parameter(torch.zeros(c.vocab_size))


class _Cls(_ParamOnly):
    def __init__(self, c):
        super().__init__()
        self.predictions = _Predictions(c)


class BertForMaskedLM(_ParamOnly):
    def __init__(self, config: BertConfigLite):
        super().__init__()
        self.config = config
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = _Cls(config)
        _bert_init(self.cls)


# ---------------------------------------------------------------------------------------------------------
# 5-mer vocabulary / tokenisation (model/dna_encoder.py:53-63, util/util.py:77-98) — no torchtext needed
# ---------------------------------------------------------------------------------------------------------
def kmer_vocab(k: int = 5) -> dict:
    """specials <MASK>=0, <CLS>=1, <UNK>=2, then the 4^k k-mers in product('ACGT', repeat=k) order."""
    vocab = {"<MASK>": 0, "<CLS>": 1, "<UNK>": 2}
    for i, kmer in enumerate(product("ACGT", repeat=k)):
        vocab["".join(kmer)] = 3 + i
    return vocab


def get_sequence_pipeline(k: int = 5):
    vocab = kmer_vocab(k)
    unk, max_len = vocab["<UNK>"], 660

    def pipeline(x: str):
        s = x[:max_len] if len(x) > max_len else x + "N" * (max_len - len(x))
        return [0, *[vocab.get(s[i : i + k], unk) for i in range(0, len(s) - k + 1, k)]]

    return pipeline


def load_pre_trained_bioscan_bert(bioscan_bert_checkpoint, k: int = 5, device=None):
    """Build BarcodeBERT and load a local checkpoint (same checkpoint conventions as the reference:
    optional 'model' / 'bert_config' entries, 'module.' prefixes, stale position_ids / classifier keys).
    `bioscan_bert_checkpoint=None` gives a randomly initialised model of the reference's shape."""
    vocab_size = len(kmer_vocab(k))
    ckpt, sd = None, None
    if bioscan_bert_checkpoint is not None:
        ckpt = torch.load(bioscan_bert_checkpoint, map_location="cpu", weights_only=False)
        sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
        sd = {(key[len("module."):] if key.startswith("module.") else key): v for key, v in sd.items()}
    cfg = dict(vocab_size=vocab_size)
    if isinstance(ckpt, dict) and "bert_config" in ckpt:
        cfg = dict(ckpt["bert_config"])
    cfg.pop("output_hidden_states", None)
    model = BertForMaskedLM(BertConfigLite(**cfg))
    if sd is not None:
        for key in ("bert.embeddings.position_ids", "classifier.weight", "classifier.bias"):
            sd.pop(key, None)
        model.load_state_dict(sd, strict=False)
    return model.to(device) if device is not None else model


class _LoRALayer(_ParamOnly):
    """Holder with the reference's names (model/dna_encoder.py:68-77): w(x) + w_b(w_a(x))."""

    def __init__(self, w: nn.Module, w_a: nn.Module, w_b: nn.Module):
        super().__init__()
        self.w = w
        self.w_a = w_a
        self.w_b = w_b
        self.in_features, self.out_features = w.in_features, w.out_features


def add_bert_lora(layers, r: int, lora_layer, w_As: list, w_Bs: list):
    for idx, layer in enumerate(layers):
        if idx not in lora_layer:
            continue
        sa = layer.attention.self
        q, v = sa.query, sa.value
        dim, dev = q.in_features, q.weight.device
        a_q, b_q = nn.Linear(dim, r, bias=False).to(dev), nn.Linear(r, dim, bias=False).to(dev)
        a_v, b_v = nn.Linear(dim, r, bias=False).to(dev), nn.Linear(r, dim, bias=False).to(dev)
        w_As += [a_q, a_v]
        w_Bs += [b_q, b_v]
        sa.query = _LoRALayer(q, a_q, b_q)
        sa.value = _LoRALayer(v, a_v, b_v)


class CLIBDDNAEncoder(nn.Module):
    def __init__(self, model, r: int, num_classes: int = 0, lora_layer=None):
        super().__init__()
        assert r > 0
        # reference: `is not None` — an empty list really disables LoRA here (dna_encoder.py:85-88)
        self.lora_layer = lora_layer if lora_layer is not None else list(range(len(model.bert.encoder.layer)))
        self.w_As, self.w_Bs = [], []
        for p in model.parameters():
            p.requires_grad = False
        add_bert_lora(model.bert.encoder.layer, r, self.lora_layer, self.w_As, self.w_Bs)
        self.reset_parameters()
        self.base_dna_encoder = model
        if num_classes > 0:
            dec = self.base_dna_encoder.cls.predictions.decoder
            self.base_dna_encoder.cls.predictions.decoder = nn.Linear(dec.in_features, num_classes).to(dec.weight.device)
        self._tower = None

    def reset_parameters(self) -> None:
        for w_A in self.w_As:
            nn.init.kaiming_uniform_(w_A.weight, a=math.sqrt(5))
        for w_B in self.w_Bs:
            nn.init.zeros_(w_B.weight)

    def tower(self) -> BertTower:
        if self._tower is None:
            pred = self.base_dna_encoder.cls.predictions
            self._tower = BertTower(self.base_dna_encoder.bert, "mlm",
                                    dict(transform_dense=pred.transform.dense, transform_ln=pred.transform.LayerNorm, decoder=pred.decoder))
        self._tower.hm["decoder"] = self.base_dna_encoder.cls.predictions.decoder
        return self._tower

    def forward(self, sequence) -> torch.Tensor:
        tw = self.tower()
        tw.training = self.training  # HF BERT applies dropout (p = 0.1) in train mode (model.train() in train_epoch.py:19)
        return tw(sequence, None, None)


class Freeze_DNA_Encoder(nn.Module):
    def forward(self, x):
        return x
