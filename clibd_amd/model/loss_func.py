"""Contrastive losses: drop-in for `bioscanclip.model.loss_func` (reference model/loss_func.py:19-201).

Same classes, constructor arguments, `forward` signatures and values as the reference, evaluated by the fused
similarity + soft-target cross-entropy kernels (include/clibd_hip.h, K9).

What the reference computes, restated once: with D = the ordered modality pairs (a, b), a != b, that survive the
`bind_to` / `no_image_text_loss` filters, every directed term CE(scale * A_n B_n^T, T) appears twice in its
`loss_list` (as `sim_a_b` of (a,b) and as `sim_b_a` of (b,a)), so

    loss = (1 / |D|) * sum_{(a,b) in D}  mean_i CE_i(scale * a_i . B^T, T_i),      T_ij = [label_i == label_j].

Each directed term is evaluated ONCE here.  Data-parallel (`ClipLoss`, world_size > 1): every rank owns the row
block of its local samples against the all-gathered features.  Exchange steps: ONE packed RCCL all-gather forward
(the L2-normalised embeddings of every modality AND the labels travel in one fp32 message: an int64 label is two
fp32 slots, bit-copied), ONE reduce-scatter backward.  The summed loss equals the reference's full N x N loss on
every rank (one scalar all-reduce; the trainer folds that scalar into its gradient all-reduce instead, see
`reduce_loss_value`), and the feature gradients carry the same world_size factor that
`torch.distributed.nn.all_gather`'s backward gives the reference, so DDP's mean all-reduce yields identical
parameter gradients.  `gather_with_grad=False` (loss_func.py:99-105: only the local block of the rank's own full
loss carries a gradient) is the same schedule without the world_size factor on the feature gradients; with
`local_loss=True` on top, the reference's gathered buffers carry no gradient at all and only logit_scale learns.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops

try:
    import torch.distributed as dist

    has_distributed = True
except ImportError:  # pragma: no cover
    dist = None
    has_distributed = False

F32 = torch.float32


def construct_label_metrix(labels: torch.Tensor) -> torch.Tensor:
    """N x N float label-equality matrix (loss_func.py:19-22).  The fused kernels never materialise it; kept for API parity."""
    return (labels.unsqueeze(0) == labels.unsqueeze(1)).float()


def _directed_pairs(present_idx: List[int], bind_to_idx, no_image_text_loss: bool):
    """positions are indices into the FILTERED feature list, as in the reference loop (loss_func.py:176-198)."""
    pairs = []
    n = len(present_idx)
    for ia in range(n):
        for ib in range(n):
            if bind_to_idx is not None and ia != bind_to_idx and ib != bind_to_idx:
                continue
            if ia == ib:
                continue
            if no_image_text_loss and (ia == 0 or ib == 0) and (ia == 2 or ib == 2):
                continue
            pairs.append((ia, ib))
    return pairs


def _check_criterion(criterion):
    if criterion is None:
        return
    ok = isinstance(criterion, nn.CrossEntropyLoss) and criterion.reduction == "mean" and criterion.label_smoothing == 0.0 \
        and criterion.weight is None
    if not ok:
        raise NotImplementedError("the HIP loss implements nn.CrossEntropyLoss() (mean reduction, probability targets) only")


def collectives_forced() -> bool:
    """CLIBD_FORCE_COLLECTIVES=1 (tests): with a process group initialised, a world of ONE rank takes the data-parallel code path
    too — packed all-gather, reduce-scatter, gradient all-reduce, broadcast — so that path can run over RCCL on a 1-GPU box
    (tests/test_multi_gpu.py).  Values are those of the local path (every collective of one rank returns its input)."""
    return os.environ.get("CLIBD_FORCE_COLLECTIVES") == "1" and has_distributed and dist.is_available() and dist.is_initialized()


def _dist_on(world_size: int) -> bool:
    return (world_size > 1 and has_distributed and dist.is_available() and dist.is_initialized()) or collectives_forced()


class _SoftCEFn(torch.autograd.Function):
    """loss = mean over directed pairs of the row-block soft-target CE; see module docstring.
    feat_grad: 0 = world_size x dL/df (gather_with_grad=True), 1 = dL/df (gather_with_grad=False), 2 = none (+ local_loss)."""

    @staticmethod
    def forward(ctx, pairs, labels, scale, rank, world, reduce_value, feat_grad, use_dist, *feats):
        dev = feats[0].device
        b, D = feats[0].shape
        M = len(feats)
        ys, invs = [], []
        for f in feats:
            y, inv = ops.l2norm_fwd(f.detach().to(F32).contiguous())   # second normalisation, loss_func.py:186-187
            ys.append(y)
            invs.append(inv)
        labels = labels.detach().to(torch.int64).contiguous()
        if use_dist:   # world > 1, or a forced one-rank group (collectives_forced)
            nf = M * b * D
            packed = torch.empty((nf + 2 * b,), dtype=F32, device=dev)     # [M, b, D] embeddings | b int64 labels as 2b fp32 slots
            torch.stack(ys, dim=0, out=packed[:nf].view(M, b, D))
            packed[nf:].copy_(labels.view(F32))
            gathered = torch.empty((world, nf + 2 * b), dtype=F32, device=dev)
            dist.all_gather_into_tensor(gathered.view(-1), packed)         # the ONE forward collective (flat views: RCCL and gloo alike)
            all_y = [gathered[:, m * b * D : (m + 1) * b * D].reshape(world * b, D).contiguous() for m in range(M)]
            all_labels = gathered[:, nf:].contiguous().view(torch.int64).view(-1)
            row0 = rank * b
        else:
            all_y, all_labels, row0 = ys, labels, 0
        N = world * b
        scale_t = scale.detach().to(F32).reshape(1).contiguous()
        loss_sum = torch.zeros((1,), dtype=F32, device=dev)
        wss = []
        for ia, ib in pairs:
            ws = ops.softce_workspace(b, N, D, dev)
            ops.softce_rows_fwd(ys[ia], all_y[ib], all_labels, row0, scale_t, loss_sum, ws)
            wss.append(ws)
        loss = loss_sum / float(len(pairs) * N)
        if use_dist and reduce_value:
            dist.all_reduce(loss)
        ctx.pairs, ctx.rank, ctx.world, ctx.dims, ctx.feat_grad, ctx.use_dist = pairs, rank, world, (b, N, D, M, row0), feat_grad, use_dist
        ctx.saved = (ys, invs, all_y, all_labels, scale_t, wss)
        ctx.scale_needs_grad = scale.requires_grad
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        ys, invs, all_y, all_labels, scale_t, wss = ctx.saved
        b, N, D, M, row0 = ctx.dims
        pairs, world = ctx.pairs, ctx.world
        dev = ys[0].device
        # world factor: mirrors the reference, where reduce-scatter(SUM) of W identical full-loss gradients hands every
        # local feature W x dL/df and DDP's mean all-reduce divides it back (SURVEY §5 "scale semantics"); logit_scale:
        # every rank of the reference holds the FULL dL/dscale, here W x the rank's partial, equal under the mean
        weight = float(world) / float(len(pairs) * N)
        wscale = dloss.detach().to(F32).reshape(1).contiguous()
        dlocal = torch.zeros((M, b, D), dtype=F32, device=dev)
        dall = torch.zeros((M, N, D), dtype=F32, device=dev) if ctx.use_dist else dlocal
        dscale = torch.zeros((1,), dtype=F32, device=dev)
        for (ia, ib), ws in zip(pairs, wss):
            ops.softce_rows_bwd(all_labels, b, N, D, row0, scale_t, weight, dlocal[ia], dall[ib], dscale, ws, weight_scale=wscale)
        ctx.saved = None
        ds = dscale.reshape(()) if ctx.scale_needs_grad else None
        if ctx.feat_grad == 2:
            return (None, None, ds, None, None, None, None, None, *([None] * M))
        if ctx.use_dist:
            send = dall.view(M, world, b, D).permute(1, 0, 2, 3).contiguous()   # [W, M, b, D]
            recv = torch.empty((M, b, D), dtype=F32, device=dev)
            dist.reduce_scatter_tensor(recv.view(-1), send.view(-1))               # the ONE backward collective
            dlocal = dlocal + recv
            if ctx.feat_grad == 1:
                dlocal = dlocal / float(world)
        grads = [ops.l2norm_bwd(dlocal[m], ys[m], invs[m]) for m in range(M)]
        return (None, None, ds, None, None, None, None, None, *grads)


def _contrastive(features, labels, logit_scale, rank, world, bind_to=None, no_image_text_loss=False, reduce_value=True, feat_grad=0,
                 use_dist=None):
    present = [(i, f) for i, f in enumerate(features) if f is not None]
    if len(present) < 2:
        raise ValueError("Too less element for calculating the contrastive loss.")
    bind_to_idx = {"image": 0, "dna": 1, "text": 2}.get(bind_to) if bind_to is not None else None
    pairs = _directed_pairs([i for i, _ in present], bind_to_idx, no_image_text_loss)
    if not pairs:
        raise ValueError("no modality pair left after bind_to / no_image_text_loss filtering")
    feats = [f for _, f in present]
    dev = feats[0].device
    shape = feats[0].shape
    for f in feats:
        if f.shape != shape or f.dim() != 2:
            raise ValueError("all modality features must be [batch, dim] with equal shapes")
    if not torch.is_tensor(logit_scale):
        logit_scale = torch.tensor(float(logit_scale), dtype=F32, device=dev)
    if use_dist is None:
        use_dist = world > 1
    return _SoftCEFn.apply(pairs, labels.to(dev), logit_scale.to(dev), rank, world, reduce_value, feat_grad, bool(use_dist), *feats)


class ContrastiveLoss(nn.Module):
    """Local (non-gathered) loss, reference loss_func.py:25-69."""

    def __init__(self, criterion, logit_scale, local_loss=False, gather_with_grad=False, rank=0, world_size=1, use_horovod=False):
        super().__init__()
        _check_criterion(criterion)
        self.criterion = criterion
        self.logit_scale = logit_scale
        self.local_loss, self.gather_with_grad = local_loss, gather_with_grad
        self.rank, self.world_size, self.use_horovod = rank, world_size, use_horovod
        self.prev_num_logits = 0
        self.labels = {}

    def forward(self, image_features, dna_features, text_features, labels, logit_scale):
        scale = logit_scale if logit_scale is not None else self.logit_scale
        return _contrastive([image_features, dna_features, text_features], labels, scale, 0, 1)


def gather_features(features, local_loss=False, gather_with_grad=False, rank=0, world_size=1, use_horovod=False):
    """All-gather [b,D] features from every rank and concatenate on dim 0 (loss_func.py:73-106)."""
    assert has_distributed, "torch.distributed did not import correctly, please use a PyTorch version with support."
    if use_horovod:
        raise NotImplementedError("horovod is a dead branch in every shipped reference config (SURVEY §2b C6)")
    if gather_with_grad:
        import torch.distributed.nn

        return torch.cat(torch.distributed.nn.all_gather(features), dim=0)
    gathered = [torch.zeros_like(features) for _ in range(world_size)]
    dist.all_gather(gathered, features)
    if not local_loss:
        gathered[rank] = features  # keep the gradient path of the local block
    return torch.cat(gathered, dim=0)


class ClipLoss(nn.Module):
    """Gathered (CLIP negative sharing) loss, reference loss_func.py:110-201."""

    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0, world_size=1, use_horovod=False,
                 criterion=None, bind_to=None, no_image_text_loss=False):
        super().__init__()
        _check_criterion(criterion)
        if use_horovod:
            raise NotImplementedError("horovod is a dead branch in every shipped reference config (SURVEY §2b C6)")
        self.local_loss, self.gather_with_grad = local_loss, gather_with_grad
        self.rank, self.world_size, self.use_horovod = rank, world_size, use_horovod
        self.criterion = criterion if criterion is not None else nn.CrossEntropyLoss()
        self.prev_num_logits = 0
        self.labels = {}
        self.bind_to = bind_to
        self.no_image_text_loss = no_image_text_loss
        # True: every rank returns the full-batch loss value (one scalar all-reduce; the reference's contract).  A trainer
        # that all-reduces gradients anyway may set it False, take the rank's PARTIAL sum (partials add up to the full loss)
        # and fold it into that all-reduce (clibd_amd.train.Trainer does): gradients do not depend on the returned value.
        self.reduce_loss_value = True

    def forward(self, image_features, dna_features, text_features, labels, logit_scale, output_dict=False):
        use_dist = _dist_on(self.world_size)
        world = self.world_size if use_dist else 1
        if self.world_size > 1 and world == 1:
            raise RuntimeError("ClipLoss(world_size>1) needs an initialised torch.distributed process group")
        feat_grad = 0
        if world > 1 and not self.gather_with_grad:
            feat_grad = 2 if self.local_loss else 1   # loss_func.py:99-105
        total = _contrastive([image_features, dna_features, text_features], labels, logit_scale, self.rank, world, self.bind_to,
                             self.no_image_text_loss, reduce_value=self.reduce_loss_value, feat_grad=feat_grad, use_dist=use_dist)
        return {"contrastive_loss": total} if output_dict else total
