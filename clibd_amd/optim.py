"""Fused AdamW over one flat fp32 bucket (torch.optim.AdamW semantics; reference scripts/train_cl.py:221).

All trainable parameters (LoRA adapters, heads, logit_scale: 1.48 M values for I+D) are re-homed as views of ONE
contiguous buffer, and so are their gradients: the data-parallel all-reduce is a single RCCL call on the flat
gradient bucket and the update is a single kernel launch.  It is a real torch.optim.Optimizer, so
the reference's LR schedulers (OneCycleLR, CosineAnnealingLR, ... train_cl.py:222-246) drive `param_groups[0]['lr']`.
"""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    AUX_SLOTS = 64  # spare fp32 slots behind the gradients in the same allocation: scalars that ride along the gradient all-reduce

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FusedAdamW: no trainable parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamW supports a single parameter group")
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        self._check_params(ps, dev)
        align = 64  # every parameter view starts on a 256-byte boundary (16-byte vector loads in the cast / pack kernels)
        pad = lambda k: (k + align - 1) // align * align
        n = sum(pad(p.numel()) for p in ps)
        self._offsets = []
        self.flat_p = torch.zeros((n,), dtype=torch.float32, device=dev)
        # gradients + AUX_SLOTS trailing scalars (e.g. the rank's partial loss value) = ONE all-reduce message (`flat_comm`);
        # the optimizer kernel only ever sees the first n elements (`flat_g`)
        self.flat_comm = torch.zeros((n + self.AUX_SLOTS,), dtype=torch.float32, device=dev)
        self.flat_g = self.flat_comm[:n]
        self.aux = self.flat_comm[n:]
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat_p[off : off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off : off + k].view(p.shape)
                p.grad = self.flat_g[off : off + k].view(p.shape)
                self._offsets.append(off)
                off += pad(k)
        self.step_count = 0
        self.grad_scale = 1.0  # e.g. 1/world_size after a SUM all-reduce (DDP's mean)

    @staticmethod
    def _check_params(ps, dev):
        if not all(p.is_cuda and p.dtype == torch.float32 and p.device == dev for p in ps):
            raise ValueError("FusedAdamW: parameters must be fp32 tensors on one GPU")

    def zero_grad(self, set_to_none: bool = False):
        # gradients stay resident as views of the flat bucket (autograd accumulates in place)
        self.flat_comm.zero_()
        for p, off in zip(self.param_groups[0]["params"], self._offsets):
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off : off + k].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        ops.adamw_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, g["lr"], b1, b2, g["eps"], g["weight_decay"],
                       self.step_count, self.grad_scale)
        return loss
