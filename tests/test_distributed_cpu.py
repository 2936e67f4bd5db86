"""world_size=2 gloo test of the data-parallel loss path (host logic only, no GPU): the packed all-gather, the
row-block ownership, the reduce-scatter of feature gradients and the world_size factor of ClipLoss.

The product has no CPU compute path, so the HIP ops the loss calls are replaced INSIDE THIS TEST by plain-torch
stand-ins with the same contracts (the kernels themselves are verified on the GPU in test_ops_gpu.py).  Expected
values come from the oracle's full N x N loss (reference semantics, loss_func.py:138-201): every rank must report
the same full-batch loss, and each rank's local feature gradient must be world_size x dL/d(local features) — what
torch.distributed.nn.all_gather's backward gives the reference (grad-norm ratio sqrt(2) at W=2)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeOps:
    """CPU stand-ins honouring the C-ABI contracts of l2norm / softce_rows (accumulating outputs, device-scalar scale)."""

    F32 = torch.float32

    @staticmethod
    def l2norm_fwd(x):
        inv = 1.0 / x.norm(dim=1).clamp_min(1e-12)
        return x * inv[:, None], inv

    @staticmethod
    def l2norm_bwd(dy, y, inv):
        return inv[:, None] * (dy - y * (dy * y).sum(1, keepdim=True))

    @staticmethod
    def softce_workspace(Nx, N, D, device):
        return {}

    @staticmethod
    def softce_rows_fwd(x, y, labels, row0, scale, loss_sum, ws):
        S = scale * (x @ y.T)
        T = (labels[row0 : row0 + x.shape[0], None] == labels[None, :]).float()
        loss_sum += -(T * torch.log_softmax(S, dim=1)).sum()
        ws.update(x=x, y=y, S=S, T=T)

    @staticmethod
    def softce_rows_bwd(labels, Nx, N, D, row0, scale, weight, dx, dy, dscale, ws, weight_scale=None):
        w = weight * (float(weight_scale) if weight_scale is not None else 1.0)
        g = w * (ws["T"].sum(1, keepdim=True) * torch.softmax(ws["S"], dim=1) - ws["T"])
        dx += scale * (g @ ws["y"])
        dy += scale * (g.T @ ws["x"])
        dscale += (g * (ws["x"] @ ws["y"].T)).sum()


def _worker(rank, world, port, nmod, bind_to, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clibd_amd.model import loss_func
        from oracle import clibd_oracle as O

        loss_func.ops = _FakeOps  # the only substitution: device kernels -> torch statements of the same contracts
        b, D = 6, 32
        g = torch.Generator().manual_seed(7)
        full = [torch.randn(world * b, D, generator=g) for _ in range(nmod)] + [None] * (3 - nmod)
        labels = torch.tensor([0, 1, 2, 2, 4, 5, 6, 0, 8, 9, 9, 11])
        log_scale = torch.tensor(2.0, requires_grad=True)
        local = [None if f is None else f[rank * b : (rank + 1) * b].clone().requires_grad_(True) for f in full]
        crit = loss_func.ClipLoss(local_loss=False, gather_with_grad=True, rank=rank, world_size=world, criterion=torch.nn.CrossEntropyLoss(),
                                  bind_to=bind_to)
        loss = crit(local[0], local[1], local[2], labels[rank * b : (rank + 1) * b], log_scale.exp())
        present = [f for f in local if f is not None]
        grads = torch.autograd.grad(loss, present + [log_scale])
        # reference semantics on the full batch
        fullv = [None if f is None else f.clone().requires_grad_(True) for f in full]
        ls2 = torch.tensor(2.0, requires_grad=True)
        ref = O.contrastive_loss(fullv, labels, ls2.exp(), bind_to=bind_to)
        rg = torch.autograd.grad(ref, [f for f in fullv if f is not None] + [ls2])
        ok = abs(float(loss) - float(ref)) < 1e-5
        for gl, gf in zip(grads[:-1], rg[:-1]):
            ok = ok and torch.allclose(gl, world * gf[rank * b : (rank + 1) * b], rtol=1e-4, atol=1e-6)
        # logit_scale: the per-rank partials (already x world) average to the full gradient under DDP's mean
        gs = grads[-1].clone()
        dist.all_reduce(gs)
        ok = ok and abs(float(gs) / world - float(rg[-1])) < 1e-4 * max(1.0, abs(float(rg[-1])))
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nmod,bind_to,port", [(2, None, 29611), (3, None, 29612), (3, "dna", 29613)])
def test_cliploss_world2_gloo_matches_full_batch_reference(nmod, bind_to, port):
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, nmod, bind_to, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


def test_cliploss_world_size_needs_process_group():
    sys.path.insert(0, ROOT)
    from clibd_amd.model import loss_func

    crit = loss_func.ClipLoss(rank=0, world_size=2, gather_with_grad=True)
    with pytest.raises(RuntimeError):
        crit(torch.zeros(2, 4), torch.zeros(2, 4), None, torch.arange(2), 1.0)


def test_directed_pairs_follow_reference_filters():
    from clibd_amd.model.loss_func import _directed_pairs

    assert _directed_pairs([0, 1], None, False) == [(0, 1), (1, 0)]
    assert len(_directed_pairs([0, 1, 2], None, False)) == 6
    assert _directed_pairs([0, 1, 2], 1, False) == [(0, 1), (1, 0), (1, 2), (2, 1)]
    assert (0, 2) not in _directed_pairs([0, 1, 2], None, True) and len(_directed_pairs([0, 1, 2], None, True)) == 4
