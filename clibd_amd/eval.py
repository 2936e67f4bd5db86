"""Eval-side counterparts on the device ("next" rows SURVEY §8f-1/2): k-mer tokenisation of raw barcodes, the embedding loop
`get_feature_and_label` (reference bioscanclip/epoch/inference_epoch.py:43-111) and the top-k retrieval of `make_prediction`
(reference bioscanclip/util/util.py:521-553: sklearn L2-normalise -> faiss.IndexFlatIP -> search(query, max_k) -> label lookup)."""
from __future__ import annotations

import weakref
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


def convert_label_dict_to_list_of_dict(label_batch: dict) -> List[dict]:
    """inference_epoch.py:8-20: {level: [labels of the batch]} -> [{level: label} per sample]"""
    return [{"order": o, "family": f, "genus": g, "species": s_}
            for o, f, g, s_ in zip(label_batch["order"], label_batch["family"], label_batch["genus"], label_batch["species"])]


def get_feature_and_label(dataloader, model, device, for_open_clip=False, multi_gpu=False, as_numpy=True):
    """The reference's embedding loop (inference_epoch.py:43-111) without its per-batch `.cpu().tolist()` round trips:
    every batch runs the towers under no_grad, the second `F.normalize` of the reference (:87-92) is the L2-norm kernel, and
    the embeddings stay on the device until ONE copy at the end.

    Batches are the reference's 7-tuples (processid, image, dna, input_ids, token_type_ids, attention_mask, label dict);
    `dna` may be a tensor of token ids or a list of raw barcode strings (tokenised by the 5-mer kernel; the remote
    BarcodeBERT tokenizer of the newer checkpoints is out of scope, SURVEY §8a-a5').
    Returns (file_name_list, image_features, dna_features, text_features, label_list) like the reference — features as
    float32 numpy arrays [N, D] (as_numpy=True; the reference builds float64 arrays of the same fp32 values) or as device
    tensors (as_numpy=False, e.g. to feed `make_prediction` without leaving the GPU); None for an absent tower."""
    if for_open_clip:
        raise NotImplementedError("open_clip towers are out of scope (DESIGN §8)")
    dev = torch.device(device)
    feats = ([], [], [])
    label_list, file_name_list = [], []
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for batch in dataloader:
                processid_batch, image_input_batch, dna_input_batch, input_ids, token_type_ids, attention_mask, label_batch = batch
                language_input = None
                if getattr(model, "language_encoder", None) is not None:
                    language_input = {"input_ids": input_ids.to(dev), "token_type_ids": token_type_ids.to(dev),
                                      "attention_mask": attention_mask.to(dev)}
                if torch.is_tensor(dna_input_batch):
                    dna_input_batch = dna_input_batch.to(dev)
                elif getattr(model, "dna_encoder", None) is not None:
                    dna_input_batch = tokenize_barcodes(list(dna_input_batch), dev)
                image_in = image_input_batch.to(dev) if getattr(model, "image_encoder", None) is not None else None
                outs = model(image_in, dna_input_batch, language_input)[:3]
                for store, out in zip(feats, outs):
                    if out is not None:
                        store.append(ops.l2norm_fwd(out.detach().to(torch.float32).contiguous())[0])
                label_list.extend(convert_label_dict_to_list_of_dict(label_batch))
                file_name_list.extend(list(processid_batch))
        if hasattr(model, "join_streams"):
            model.join_streams()
    finally:
        model.train(was_training)
    out = []
    for store in feats:
        if not store:
            out.append(None)
        else:
            t = torch.cat(store, dim=0)
            out.append(t.cpu().numpy() if as_numpy else t)
    return file_name_list, out[0], out[1], out[2], label_list


def prepare_key_bank(keys_feature: torch.Tensor) -> "ops.KeyBank":
    """L2-normalise a key set and prepare it ONCE for the pre-filtered search (what the reference's faiss.IndexFlatIP(dim) +
    index.add(keys) does once per key set, util/util.py:521-526)."""
    kf, _ = ops.l2norm_fwd(keys_feature.detach().to(torch.float32).contiguous())
    return ops.KeyBank(kf)


# One prepared bank per key TENSOR: `make_prediction` is called with the same keys for every query set of an evaluation
# (inference_epoch.py / util.py:521-553 build ONE faiss index per key set), and a bank costs a normalisation, a bf16 image of the
# keys (630 MB at 410 k x 768) and a conversion kernel.  The key is the tensor's storage, shape, strides and version counter, so an
# in-place update of the keys invalidates the entry; only the most recent bank is kept.
_bank_cache: dict = {}


def clear_key_bank_cache() -> None:
    """Drop the cached bank (normalised fp32 keys + their bf16 image: ~1.9 GB at 410 k x 768), e.g. before training resumes after an evaluation."""
    _bank_cache.clear()


def _cached_key_bank(keys_feature: torch.Tensor) -> "ops.KeyBank":
    # the entry must not outlive the caller's tensor (it would pin ~1.9 GB of device memory until the next call) and a tensor
    # whose version counter cannot be read (created under torch.inference_mode()) is simply not cached
    try:
        version = keys_feature._version
    except RuntimeError:
        return prepare_key_bank(keys_feature)
    key = (keys_feature.data_ptr(), tuple(keys_feature.shape), tuple(keys_feature.stride()), keys_feature.dtype, version, str(keys_feature.device))
    hit = _bank_cache.get("entry")
    if hit is not None and hit[0] == key and hit[1]() is keys_feature:
        return hit[2]
    bank = prepare_key_bank(keys_feature)
    _bank_cache["entry"] = (key, weakref.ref(keys_feature), bank)
    weakref.finalize(keys_feature, _drop_entry_of, key)    # the bank goes when the caller's tensor does
    return bank


def _drop_entry_of(key) -> None:
    hit = _bank_cache.get("entry")
    if hit is not None and hit[0] == key:
        _bank_cache.clear()


def topk_search(query_feature: torch.Tensor, keys_feature, max_k: int = 5, exact: bool = False, cache: bool = True):
    """(similarities, indices) of IndexFlatIP.search on L2-normalised features.  `keys_feature`: a tensor, or a bank from
    `prepare_key_bank` (re-used across query batches).  Large banks take the pre-filtered search (bf16 approximate scores, exact
    fp32 re-score of every key inside the rigorous error band: indices and similarities identical to the exact kernel's); queries
    it flags — candidate lists full inside the band, e.g. many duplicate keys — are re-run through the exact kernel.  exact=True
    forces the exact kernel for everything.  cache=False: the prepared bank of a key TENSOR is not kept (a temporary device copy of host keys
    could never hit and would only pin memory)."""
    q, _ = ops.l2norm_fwd(query_feature.detach().to(torch.float32).contiguous())
    bank = keys_feature if isinstance(keys_feature, ops.KeyBank) else None
    if bank is None:
        Nk, D = keys_feature.shape
        if exact or D % 64 != 0 or D > ops.KeyBank.MAX_D or not (4096 <= Nk < ops.KeyBank.MAX_KEYS):
            kf, _ = ops.l2norm_fwd(keys_feature.detach().to(torch.float32).contiguous())
            return ops.topk_ip(q, kf, max_k)
        bank = _cached_key_bank(keys_feature) if cache else prepare_key_bank(keys_feature)
    if exact:
        return ops.topk_ip(q, bank.keys, max_k)
    sim, idx, ovf = ops.topk_ip_fast(q, bank, max_k)
    bad = torch.nonzero(ovf).flatten()          # (host sync: the eval path hands numpy arrays back anyway)
    if bad.numel():
        s2, i2 = ops.topk_ip(q[bad].contiguous(), bank.keys, max_k)
        sim[bad], idx[bad] = s2, i2
    return sim, idx


def make_prediction(query_feature, keys_feature, keys_label: List[dict], with_similarity=False, with_indices=False, max_k=5, device=None):
    """Same return convention as the reference `make_prediction`: a list of {level: [k predicted labels]} per query,
    optionally followed by the similarity and index arrays (numpy, like faiss).
    `keys_feature`: a device tensor (its prepared bank is cached per tensor and re-used by later calls), a `KeyBank` from
    `prepare_key_bank` (evaluating several query sets against one key set: prepare it once, as the reference builds one faiss index),
    or host data (numpy / CPU tensor: copied and prepared on every call — there is no cheap way to know a host array is unchanged)."""
    dev = device or (query_feature.device if torch.is_tensor(query_feature) else torch.device("cuda"))
    qf = torch.as_tensor(np.asarray(query_feature) if not torch.is_tensor(query_feature) else query_feature).to(dev)
    kf = keys_feature if isinstance(keys_feature, ops.KeyBank) else \
        torch.as_tensor(np.asarray(keys_feature) if not torch.is_tensor(keys_feature) else keys_feature).to(dev)
    # only a tensor the CALLER holds on the device can hit the cache again; host keys (numpy, the reference's convention) become a fresh device copy per call
    on_device = torch.is_tensor(keys_feature) and keys_feature.device == kf.device if torch.is_tensor(kf) else False
    sim, idx = topk_search(qf, kf, max_k, cache=on_device)
    idx_h, sim_h = idx.cpu().numpy(), sim.cpu().numpy()
    pred_list = [{level: [keys_label[i][level] for i in row] for level in LEVELS} for row in idx_h]
    out = [pred_list]
    if with_similarity:
        out.append(sim_h)
    if with_indices:
        out.append(idx_h)
    return out[0] if len(out) == 1 else out
