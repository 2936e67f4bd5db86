"""Eval-side counterparts on the device ("next" rows SURVEY §8f-1/2): k-mer tokenisation of raw barcodes and the top-k
retrieval of `make_prediction` (reference bioscanclip/util/util.py:521-553: sklearn L2-normalise -> faiss.IndexFlatIP ->
search(query, max_k) -> label lookup)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import ops

LEVELS = ["order", "family", "genus", "species"]


def tokenize_barcodes(sequences: Sequence[str], device, max_len: int = 660, k: int = 5) -> torch.Tensor:
    """[0] + 132 5-mer ids per barcode (truncate / 'N'-pad to 660 nt), as `get_sequence_pipeline` produces."""
    buf = b"".join(s[:max_len].ljust(max_len, "N").encode("ascii", "replace") for s in sequences)
    seq = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8).reshape(len(sequences), max_len).copy()).to(device)
    return ops.kmer_tokenize(seq, k)


def topk_search(query_feature: torch.Tensor, keys_feature: torch.Tensor, max_k: int = 5):
    """(similarities, indices) of IndexFlatIP.search on L2-normalised features."""
    q, _ = ops.l2norm_fwd(query_feature.detach().to(torch.float32).contiguous())
    kf, _ = ops.l2norm_fwd(keys_feature.detach().to(torch.float32).contiguous())
    return ops.topk_ip(q, kf, max_k)


def make_prediction(query_feature, keys_feature, keys_label: List[dict], with_similarity=False, with_indices=False, max_k=5, device=None):
    """Same return convention as the reference `make_prediction`: a list of {level: [k predicted labels]} per query,
    optionally followed by the similarity and index arrays (numpy, like faiss)."""
    dev = device or (query_feature.device if torch.is_tensor(query_feature) else torch.device("cuda"))
    qf = torch.as_tensor(np.asarray(query_feature) if not torch.is_tensor(query_feature) else query_feature).to(dev)
    kf = torch.as_tensor(np.asarray(keys_feature) if not torch.is_tensor(keys_feature) else keys_feature).to(dev)
    sim, idx = topk_search(qf, kf, max_k)
    idx_h, sim_h = idx.cpu().numpy(), sim.cpu().numpy()
    pred_list = [{level: [keys_label[i][level] for i in row] for level in LEVELS} for row in idx_h]
    out = [pred_list]
    if with_similarity:
        out.append(sim_h)
    if with_indices:
        out.append(idx_h)
    return out[0] if len(out) == 1 else out
