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


class DevicePrefetcher:
    """Feeds host batches to the device one step ahead on a copy stream, so the PCIe transfer of batch i+1 (1.2 GB of fp32
    images at b = 2048: ~24 ms at ~52 GB/s) runs under the compute of batch i instead of in front of it — the reference's loop
    copies synchronously at the top of every step (epoch/train_epoch.py:26-32).

    `batches` yields dicts / tuples / lists of CPU tensors (pinned memory makes the copies asynchronous; other leaves are
    passed through).  Iterating yields the same structure with device tensors that are safe to use on the current stream."""

    def __init__(self, batches, device):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._next = None
        self._preload()

    def _to_device(self, obj):
        if torch.is_tensor(obj):
            return obj.to(self.device, non_blocking=True)
        if isinstance(obj, dict):
            return {k: self._to_device(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._to_device(v) for v in obj)
        return obj

    def _record(self, obj, stream):
        if torch.is_tensor(obj):
            if obj.is_cuda:
                obj.record_stream(stream)
        elif isinstance(obj, dict):
            for v in obj.values():
                self._record(v, stream)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                self._record(v, stream)

    def _preload(self):
        try:
            host = next(self.it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.copy_stream):
            self._next = self._to_device(host)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.copy_stream)     # batch i has landed
        batch = self._next
        self._record(batch, cur)              # allocated on the copy stream, consumed on the compute stream
        self._preload()                       # batch i+1 starts flying now, under the step that uses batch i
        return batch
