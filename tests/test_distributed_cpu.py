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


# ------------------------------------------------------------------------------- reference goldens at world_size 2
def _golden_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clibd_amd.model import loss_func

        loss_func.ops = _FakeOps
        gold = torch.load(os.path.join(ROOT, "tests", "golden", "loss_w2_golden.pt"), map_location="cpu", weights_only=False)
        ok = gold["world_size"] == world
        worst = 0.0
        for c in gold["cases"]:
            N = c["labels"].numel()
            b = N // world
            sl = slice(rank * b, (rank + 1) * b)
            feats = [None if f is None else f[sl].clone().requires_grad_(True) for f in c["features"]]
            ls = c["log_scale"].clone().requires_grad_(True)
            crit = loss_func.ClipLoss(local_loss=False, gather_with_grad=c["gather_with_grad"], rank=rank, world_size=world,
                                      criterion=torch.nn.CrossEntropyLoss(), bind_to=c["bind_to"], no_image_text_loss=c["no_image_text_loss"])
            loss = crit(feats[0], feats[1], feats[2], c["labels"][sl], ls.exp())
            present = [f for f in feats if f is not None]
            grads = torch.autograd.grad(loss, present + [ls])
            ref = c["per_rank"][rank]
            ok = ok and abs(float(loss) - float(ref["loss"])) < 5e-6
            for g_, r_ in zip(grads[:-1], ref["grads"][:-1]):      # local feature gradients: identical per rank
                err = float((g_ - r_).abs().max() / (r_.abs().max() + 1e-30))
                worst = max(worst, err)
                ok = ok and err < 1e-4
            # logit_scale: the reference holds the full gradient on every rank; here W x the rank's partial, equal under the mean
            gs = grads[-1].clone()
            dist.all_reduce(gs)
            ok = ok and abs(float(gs) / world - float(ref["grads"][-1])) < 1e-4 * max(1.0, abs(float(ref["grads"][-1])))
        out[rank] = (bool(ok), worst)
    finally:
        dist.destroy_process_group()


def test_cliploss_world2_matches_reference_goldens():
    """tests/golden/loss_w2_golden.pt = the reference's own ClipLoss on two gloo ranks (make_golden_r2.py): loss value, local
    feature gradients (gather_with_grad True and False) and the logit-scale gradient, per rank."""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_golden_worker, args=(2, 29621, out), nprocs=2, join=True)
    assert all(v[0] for v in dict(out).values()), dict(out)


# ------------------------------------------------------------------------------- Trainer: flat bucket + 1/W + broadcast
class _ToyTower:
    grad_sink = None

    def __init__(self, lin):
        self.lin = lin

    def trainable_params(self):
        return list(self.lin.parameters())


class _ToyEnc(torch.nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.lin = torch.nn.Linear(din, dout)
        self.unused = torch.nn.Parameter(torch.ones(3))   # never receives a gradient: must not be weight-decayed
        self._tower = _ToyTower(self.lin)

    def tower(self):
        return self._tower

    def forward(self, x):
        return self.lin(x)


class _ToyCLIP(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.image_encoder, self.dna_encoder, self.language_encoder = _ToyEnc(12, 16), _ToyEnc(9, 16), None
        self.logit_scale = torch.nn.Parameter(torch.tensor(1.5))

    def forward(self, image, dna, text):
        n = torch.nn.functional.normalize
        return n(self.image_encoder(image), dim=-1), n(self.dna_encoder(dna), dim=-1), None, self.logit_scale.exp(), None


def _cpu_adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """torch.optim.AdamW arithmetic on the flat bucket (the contract of clibd_adamw_step, include/clibd_hip.h)."""
    with torch.no_grad():
        gg = g * grad_scale
        p.mul_(1 - lr * weight_decay)
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        p.addcdiv_(m / (1 - beta1 ** step), (v / (1 - beta2 ** step)).sqrt().add_(eps), value=-lr)


def _trainer_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clibd_amd import optim, train
        from clibd_amd.model import loss_func
        from oracle import clibd_oracle as O

        loss_func.ops = _FakeOps
        optim.ops.adamw_step = _cpu_adamw_step
        optim.FusedAdamW._check_params = staticmethod(lambda ps, dev: None)
        torch.manual_seed(100 + rank)            # different initialisation per rank: the trainer must broadcast rank 0's
        model = _ToyCLIP()
        torch.manual_seed(100)
        ref_model = _ToyCLIP()                    # == rank 0's initialisation
        b = 5
        g = torch.Generator().manual_seed(3)
        image, dna = torch.randn(world * b, 12, generator=g), torch.randn(world * b, 9, generator=g)
        labels = torch.tensor([0, 1, 2, 2, 4, 5, 6, 0, 8, 9])
        tr = train.Trainer(model, lr=1e-2, world_size=world, rank=rank, all_gather=True)
        ropt = torch.optim.AdamW([p for n, p in ref_model.named_parameters() if "unused" not in n], lr=1e-2, weight_decay=1e-2)
        sl = slice(rank * b, (rank + 1) * b)
        ok = True
        for _ in range(3):
            loss = tr.step(image[sl], dna[sl], None, labels[sl])
            ropt.zero_grad()
            i, d, _, sc, _ = ref_model(image, dna, None)
            rl = O.contrastive_loss([i, d, None], labels, sc)
            rl.backward()
            ropt.step()
            ok = ok and abs(float(loss) - float(rl)) < 1e-5           # every rank reports the full-batch loss
        for (n, p), (_, q) in zip(model.named_parameters(), ref_model.named_parameters()):
            ok = ok and torch.allclose(p, q, rtol=1e-4, atol=1e-6)
        ok = ok and torch.equal(model.image_encoder.unused, torch.ones(3))   # no gradient -> untouched (torch.optim.AdamW skips it)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_trainer_world2_gloo_matches_full_batch_adamw():
    """Trainer.step at world_size 2 (gloo, device ops replaced by torch statements of their contracts): parameter broadcast
    from rank 0, flat-bucket SUM all-reduce + 1/W scale, the loss value folded into that all-reduce, AdamW — against one
    process training the same toy towers on the full batch with torch.optim.AdamW and the oracle's loss
    (reference: DDP + AdamW, scripts/train_cl.py:204,221)."""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_worker, args=(2, 29622, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


# ------------------------------------------------------------------------------- Trainer: all-reduce in backward order
class _SinkLinearFn(torch.autograd.Function):
    """A tower in miniature: writes its parameter gradients straight into the trainer's flat bucket and reports each
    gradient group as it completes (the protocol of clibd_amd.towers._Tower / _TowerFn)."""

    @staticmethod
    def forward(ctx, tower, x, w, b):
        ctx.tower = tower
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        tw = ctx.tower
        tw.grad_sink[id(tw.lin.bias)].add_(dy.sum(0))
        tw._ready(0)                                   # group 0 = [bias] ("head"), then group 1 = [weight] ("layer")
        tw.grad_sink[id(tw.lin.weight)].add_(dy.t() @ x)
        tw._ready(1)
        tw._ready(None)
        return None, None, None, None


class _HookedTower(_ToyTower):
    on_grads_ready = None

    def grad_groups(self):
        return [[self.lin.bias], [self.lin.weight]]

    def _ready(self, k):
        if self.on_grads_ready is not None:
            self.on_grads_ready(k)


class _HookedEnc(_ToyEnc):
    def __init__(self, din, dout):
        super().__init__(din, dout)
        self._tower = _HookedTower(self.lin)

    def forward(self, x):
        return _SinkLinearFn.apply(self._tower, x, self.lin.weight, self.lin.bias)


class _HookedCLIP(_ToyCLIP):
    def __init__(self):
        super().__init__()
        self.image_encoder, self.dna_encoder = _HookedEnc(12, 16), _HookedEnc(9, 16)


def _bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clibd_amd import optim, train
        from clibd_amd.model import loss_func
        from oracle import clibd_oracle as O

        loss_func.ops = _FakeOps
        optim.ops.adamw_step = _cpu_adamw_step
        optim.FusedAdamW._check_params = staticmethod(lambda ps, dev: None)
        torch.manual_seed(200 + rank)
        model = _HookedCLIP()
        torch.manual_seed(200)
        ref_model = _ToyCLIP()                    # plain autograd, rank 0's initialisation
        torch.manual_seed(200)
        ref_model.load_state_dict(_HookedCLIP().state_dict())
        b = 5
        g = torch.Generator().manual_seed(4)
        image, dna = torch.randn(world * b, 12, generator=g), torch.randn(world * b, 9, generator=g)
        labels = torch.tensor([0, 1, 2, 2, 4, 5, 6, 0, 8, 9])
        issued = []
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(t, *a, **k):
            issued.append(t.numel())
            return real_all_reduce(t, *a, **k)

        train.dist.all_reduce = counting_all_reduce
        tr = train.Trainer(model, lr=1e-2, world_size=world, rank=rank, all_gather=True, bucket_bytes=256)   # 64 elements: every group is a bucket
        why = []
        ok = tr._bucketed
        # flat order follows the backward: per tower bias (group 0) then weight (group 1); logit_scale last
        names = {id(p): n for n, p in model.named_parameters()}
        order = [names[id(p)] for p in tr.optimizer.param_groups[0]["params"]]
        if order != ["image_encoder.lin.bias", "image_encoder.lin.weight", "dna_encoder.lin.bias", "dna_encoder.lin.weight", "logit_scale"]:
            why.append(("order", order))
        ropt = torch.optim.AdamW([p for n, p in ref_model.named_parameters() if "unused" not in n], lr=1e-2, weight_decay=1e-2)
        sl = slice(rank * b, (rank + 1) * b)
        for _ in range(3):
            issued.clear()
            loss = tr.step(image[sl], dna[sl], None, labels[sl])
            if not (len(issued) == 5 and sum(issued) == tr.optimizer.flat_comm.numel()):   # 2 towers x 2 groups during the backward + the tail; disjoint cover
                why.append(("issued", list(issued), tr.optimizer.flat_comm.numel()))
            ropt.zero_grad()
            i, d, _, sc, _ = ref_model(image, dna, None)
            rl = O.contrastive_loss([i, d, None], labels, sc)
            rl.backward()
            ropt.step()
            if abs(float(loss.detach()) - float(rl.detach())) >= 1e-5:
                why.append(("loss", float(loss.detach()), float(rl.detach())))
        ref = dict(ref_model.named_parameters())
        for n, p in model.named_parameters():
            if not torch.allclose(p, ref[n], rtol=1e-4, atol=1e-6):
                why.append(("param", n))
        out[rank] = True if (ok and not why) else repr(why)
    finally:
        dist.destroy_process_group()


def test_trainer_world2_bucketed_allreduce_in_backward_order():
    """Large gradient sets (full fine-tune) are all-reduced in pieces, each issued by the tower's backward as soon as a
    gradient group is complete; here with toy towers that speak the same protocol and a 256-byte bucket: the flat bucket is
    laid out in backward order, five disjoint collectives cover it, and three steps match single-process AdamW on the full
    batch (reference: DDP's bucketed gradient all-reduce, scripts/train_cl.py:204)."""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker, args=(2, 29633, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}
