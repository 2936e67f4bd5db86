"""Synthetic batch producer with the reference's batch contract (util/dataset.py:294-302):
(processid, image f32[b,3,224,224] in [0,1), dna int64[b,133] = [0] + 132 x U{3..1026}, input_ids int64[b,20],
token_type_ids, attention_mask (first U{6..20} positions 1), label int64[b] = arange + rank*b)."""
from __future__ import annotations

import torch


def synthetic_batch(batch: int, device, seed: int = 42, rank: int = 0, with_text: bool = False, duplicate_labels: bool = False):
    g = torch.Generator(device="cpu").manual_seed(seed + rank)
    image = torch.rand((batch, 3, 224, 224), generator=g)
    dna = torch.cat([torch.zeros((batch, 1), dtype=torch.int64), torch.randint(3, 1027, (batch, 132), generator=g)], dim=1)
    labels = torch.arange(batch, dtype=torch.int64) + rank * batch
    if duplicate_labels:
        labels = labels // 2
    out = {"processid": [f"SYN{rank}_{i}" for i in range(batch)], "image": image.to(device), "dna": dna.to(device), "labels": labels.to(device),
           "text": None}
    if with_text:
        ids = torch.randint(0, 30522, (batch, 20), generator=g)
        lens = torch.randint(6, 21, (batch,), generator=g)
        out["text"] = {"input_ids": ids.to(device), "token_type_ids": torch.zeros_like(ids).to(device),
                       "attention_mask": (torch.arange(20)[None, :] < lens[:, None]).long().to(device)}
    return out
