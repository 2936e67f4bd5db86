"""The contrastive training step: counterpart of the body of `train_epoch`
(reference epoch/train_epoch.py:21-63) and of the DDP/AdamW setup in scripts/train_cl.py:195-262.

One process per GPU.  Data parallel = the packed RCCL all-gather / reduce-scatter inside `ClipLoss`
(clibd_amd.model.loss_func) + ONE flat-bucket gradient all-reduce here; no DDP wrapper, no per-step host sync
(the reference's `loss.item()` / anomaly mode / per-epoch tokenizer download are not reproduced).
"""
from __future__ import annotations

from typing import Optional

import torch

from .model.loss_func import ClipLoss, ContrastiveLoss, collectives_forced
from .optim import FusedAdamW

try:
    import torch.distributed as dist
except ImportError:  # pragma: no cover
    dist = None


def scale_learning_rate(lr, batch_size, base_batch_size=500, world_size=1):
    """util/util.py:753-756"""
    return lr * batch_size * world_size / base_batch_size


def _gradient_reachable(model, fix_temperature) -> list:
    """Trainable parameters the step can actually produce a gradient for.  torch.optim.AdamW (the reference's optimizer,
    train_cl.py:221) skips parameters whose .grad is None — logit_scale under fix_temperature, BertModel.pooler.* and the
    replaced MLM decoder's orphaned cls.predictions.bias never receive one — while a flat-bucket update would weight-decay
    them; they are therefore kept out of the bucket and stay exactly as initialised / loaded, as in the reference."""
    reach = set()
    for enc in (model.image_encoder, model.dna_encoder, model.language_encoder):
        if enc is not None and hasattr(enc, "tower"):
            reach.update(id(p) for p in enc.tower().trainable_params())
    if fix_temperature is None:
        reach.add(id(model.logit_scale))
    return [p for p in model.parameters() if p.requires_grad and id(p) in reach]


def _towers(model):
    return [enc.tower() for enc in (model.image_encoder, model.dna_encoder, model.language_encoder)
            if enc is not None and hasattr(enc, "tower")]


def _backward_order(model, params):
    """`params` re-ordered so that every tower's gradient groups (towers._Tower.grad_groups: head, layers top -> bottom,
    embeddings — the order the backward completes them) are contiguous in the flat bucket; parameters outside every group
    (logit_scale) go last.  Returns (ordered params, [[group sizes in parameters] per tower])."""
    want = {id(p): p for p in params}
    seen, ordered, counts = set(), [], []
    for tw in _towers(model):
        groups = tw.grad_groups() if hasattr(tw, "grad_groups") else [tw.trainable_params()]
        c = []
        for ps in groups:
            k = 0
            for p in ps:
                if id(p) in want and id(p) not in seen:
                    seen.add(id(p))
                    ordered.append(p)
                    k += 1
            c.append(k)
        counts.append(c)
    ordered += [p for p in params if id(p) not in seen]
    return ordered, counts


class Trainer:
    def __init__(self, model, lr: float = 1e-3, world_size: int = 1, rank: int = 0, all_gather: bool = True,
                 fix_temperature: Optional[float] = None, bind_to=None, no_image_text_loss=False, weight_decay: float = 1e-2,
                 broadcast_parameters: bool = True, bucket_bytes: int = 64 << 20, fp8_recalibrate_every: int = 0,
                 numerics: Optional[dict] = None):
        """bucket_bytes: at world_size > 1, when the gradients are at least two buckets long (full fine-tune: 694 MB), the
        all-reduce is issued in pieces of about this size as the backward completes them, each on the stream that produced
        it, so RCCL runs under the rest of the backward; smaller gradient sets (LoRA: 6 MB) keep the single all-reduce.
        fp8_recalibrate_every = N > 0: fp8-forward mode (SimpleCLIP.enable_fp8_forward) re-measures its per-layer activation
        scales on the incoming batch before steps 0, N, 2N, ... (one extra bf16 forward each time), so the static scales of
        a long run follow the activations as the adapters train.
        numerics: backward arithmetic switches for every tower of `model` (clibd_amd.engine.NUMERICS_CHOICES); None keeps what the
        towers were constructed with (environment defaults)."""
        self.model, self.world_size, self.rank = model, world_size, rank
        if numerics:    # e.g. dict(residual_grad="fp32"): the reference's fp32 residual-gradient stream (engine.NUMERICS_CHOICES)
            model.set_numerics(**numerics)
        # the data-parallel path: world_size > 1, or a one-rank group under CLIBD_FORCE_COLLECTIVES=1 (tests: RCCL on a 1-GPU box)
        self._dist = world_size > 1 or collectives_forced()
        self.fix_temperature = fix_temperature
        self.fp8_recalibrate_every, self._steps_done = int(fp8_recalibrate_every), 0
        ordered, counts = _backward_order(model, _gradient_reachable(model, fix_temperature))
        self.optimizer = FusedAdamW(ordered, lr=lr, weight_decay=weight_decay)
        self._plan_buckets(counts, bucket_bytes)
        if all_gather:
            self.criterion = ClipLoss(local_loss=False, gather_with_grad=True, rank=rank, world_size=world_size,
                                      criterion=torch.nn.CrossEntropyLoss(), bind_to=bind_to, no_image_text_loss=no_image_text_loss)
            # the rank's partial loss rides in a spare slot of the gradient all-reduce: no separate scalar collective
            self.criterion.reduce_loss_value = not self._dist
        else:
            self.criterion = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
        self.optimizer.grad_scale = 1.0 / world_size  # SUM all-reduce then mean: what DDP does (train_cl.py:204)
        if self._dist and broadcast_parameters:
            # DDP(model) broadcasts rank 0's parameters and buffers at construction (train_cl.py:204): adapters, heads and the
            # replaced decoder are randomly initialised per process.  Trainable values live in the flat bucket (one message);
            # frozen weights and buffers follow tensor by tensor (once, at start-up).
            dist.broadcast(self.optimizer.flat_p, src=0)
            own = {id(p) for p in self.optimizer.param_groups[0]["params"]}
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    if id(t) not in own:
                        dist.broadcast(t.data, src=0)
            # `.data` writes do not bump Parameter._version, which is all TransformerStack._key() watches: a forward that ran
            # before this constructor (eval, fp8 calibration) would leave bf16 / fp8 weight images of the pre-broadcast values
            for tw in _towers(model):
                if hasattr(tw, "invalidate_weight_images"):
                    tw.invalidate_weight_images()
        # let the towers accumulate parameter gradients straight into the optimizer's flat bucket
        sink = {id(p): p.grad for p in self.optimizer.param_groups[0]["params"]}
        for ti, tw in enumerate(_towers(model)):
            tw.grad_sink = sink
            if self._bucketed:
                tw.on_grads_ready = (lambda k, ti=ti: self._grads_ready(ti, k))

    # ---- gradient all-reduce in backward order (world_size > 1, large gradient sets) ------------------------------------------
    def _plan_buckets(self, counts, bucket_bytes):
        opt = self.optimizer
        n = opt.flat_g.numel()
        offs = list(opt._offsets) + [n]
        self._group_end, k = [], 0           # per tower: flat offset at which each group ends (groups are contiguous, in order)
        self._tower_start = []
        for c in counts:
            self._tower_start.append(offs[k])
            ends = []
            for cnt in c:
                k += cnt
                ends.append(offs[k])
            self._group_end.append(ends)
        self._tail_start = offs[k]            # logit_scale & co, then the AUX slots
        self._bucket_elems = max(1, bucket_bytes // 4)
        self._bucketed = self._dist and n >= 2 * self._bucket_elems
        self._done = list(self._tower_start)  # per tower: flat offset up to which the all-reduce has been issued this step
        self._works = []

    def _issue(self, a, b):
        if b > a:
            self._works.append(dist.all_reduce(self.optimizer.flat_comm[a:b], async_op=True))

    def _grads_ready(self, ti, k):
        """Tower ti finished gradient group k (None: all of them).  Called from inside that tower's backward, i.e. with its
        stream current: the collective is ordered after the kernels that produced the gradients and runs beside what follows."""
        ends = self._group_end[ti]
        upto = ends[-1] if k is None else ends[k]
        if upto - self._done[ti] >= self._bucket_elems or (k is None and upto > self._done[ti]):
            self._issue(self._done[ti], upto)
            self._done[ti] = upto

    def step(self, image, dna, text, labels) -> torch.Tensor:
        """forward (all towers) -> loss -> backward -> gradient all-reduce -> AdamW.  Returns the (device) loss.
        Collectives per step at world_size > 1: one packed all-gather (embeddings + labels), one reduce-scatter (feature
        gradients), one all-reduce (flat gradient bucket + the loss value)."""
        if self.fp8_recalibrate_every > 0 and self._steps_done % self.fp8_recalibrate_every == 0:
            self.model.enable_fp8_forward(calibration_inputs=(image, dna, text))
        self._steps_done += 1
        self.optimizer.zero_grad()
        image_out, dna_out, text_out, logit_scale, _ = self.model(image, dna, text)
        if self.fix_temperature is not None:
            logit_scale = 1.0 / 0.07
        loss = self.criterion(image_out, dna_out, text_out, labels, logit_scale)
        loss.backward()
        if hasattr(self.model, "join_streams"):
            self.model.join_streams()  # tower backward passes ran on the towers' own streams
        loss = loss.detach()
        if self._dist:
            fold = not getattr(self.criterion, "reduce_loss_value", True)
            if fold:
                self.optimizer.aux[0:1].copy_(loss.reshape(1))     # partial sums add up to the full-batch loss
            if self._bucketed:
                # what the towers have not handed over yet (a tower without hooks, a group below the bucket size), then the
                # tail (logit_scale, the loss slot); the pieces issued during the backward are already in flight
                for ti, ends in enumerate(self._group_end):
                    if ends:
                        self._issue(self._done[ti], ends[-1])
                        self._done[ti] = self._tower_start[ti]
                self._issue(self._tail_start, self.optimizer.flat_comm.numel())
                for w in self._works:
                    w.wait()
                self._works = []
            else:
                dist.all_reduce(self.optimizer.flat_comm)
            if fold:
                loss = self.optimizer.aux[0].clone()
        self.optimizer.step()
        return loss


def train_epoch(total_epochs, epoch, dataloader, trainer: Trainer, device, scheduler=None, log_every: int = 0):
    """Epoch loop over the reference's 7-tuple batches (util/dataset.py:294-302)."""
    trainer.model.train()
    running = torch.zeros((), device=device)
    n = 0
    for step, batch in enumerate(dataloader):
        _pid, image, dna, input_ids, token_type_ids, attention_mask, label = batch
        text = None
        if trainer.model.language_encoder is not None:
            text = {"input_ids": input_ids.to(device), "token_type_ids": token_type_ids.to(device), "attention_mask": attention_mask.to(device)}
        loss = trainer.step(image.to(device), dna.to(device), text, label.to(device))
        if scheduler is not None:
            scheduler.step()
        running += loss
        n += 1
        if log_every and (step + 1) % log_every == 0 and trainer.rank == 0:
            print(f"Epoch {epoch}/{total_epochs} step {step + 1}: loss {loss.item():.4f}")
    return (running / max(n, 1)).item()
