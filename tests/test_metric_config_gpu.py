"""BASELINE.json's metric configuration on ONE GPU: global batch 2048 = per-GPU batch 2048 (M = 403 456 ViT token rows,
272 384 DNA token rows).  The CPU oracle cannot run this size in seconds, so parity is carried by size-independent
properties:

  * batch-split invariance — every token row's arithmetic is independent of the batch it sits in, so the embeddings of the
    b=2048 forward equal those of eight b=256 forwards (same kernels, different tile grids / 32-bit offset ranges);
  * the loss of the b=2048 embeddings against a CPU statement of the reference's loss on those embeddings (1e-3), and
    ln(2048) +- 0.5 at random init;
  * linearity of the tower backward over samples — parameter gradients of the b=2048 backward equal the sum over eight
    b=256 backward passes fed the same upstream rows;
  * the 256x256 GEMM's 32-bit operand-offset guard (csrc/gemm256.hip: operands >= 4 GiB go to the 64-bit-addressed kernel).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32
NB, CH = 2048, 256


@pytest.fixture(scope="module")
def big(dev):
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

    free, _ = torch.cuda.mem_get_info()
    if free < 200 * 2 ** 30:
        pytest.skip("needs ~170 GB of free HBM (MI355X: 288 GB)")
    torch.manual_seed(2048)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
    with torch.no_grad():
        for enc in (model.image_encoder, model.dna_encoder):
            for wb in enc.w_Bs:
                wb.weight.normal_(0, 0.02)
    model = model.to(dev).eval()  # dropout off: deterministic rows
    batch = synthetic_batch(NB, dev, seed=42, rank=0, with_text=False)
    return model, batch


def _params(model):
    return [p for p in model.parameters() if p.requires_grad and p is not model.logit_scale]


def test_b2048_embeddings_equal_chunked_and_loss_matches_cpu(dev, big):
    from clibd_amd.model import ClipLoss
    from oracle import clibd_oracle as O

    model, batch = big
    with torch.no_grad():
        i_full, d_full, _, scale, _ = model(batch["image"], batch["dna"], None)
        chunks = [model(batch["image"][s : s + CH], batch["dna"][s : s + CH], None) for s in range(0, NB, CH)]
    torch.cuda.synchronize()
    i_ch = torch.cat([c[0] for c in chunks])
    d_ch = torch.cat([c[1] for c in chunks])
    for full, ch in ((i_full, i_ch), (d_full, d_ch)):
        assert torch.isfinite(full).all()
        assert (full - ch).abs().max().item() <= 1e-6, (full - ch).abs().max().item()   # same per-row arithmetic (bit-exact in practice)
        assert torch.allclose(full.norm(dim=1), torch.ones(NB, device=dev), atol=1e-4)
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    loss = float(crit(i_full, d_full, None, batch["labels"], scale))
    ref = float(O.contrastive_loss([i_full.cpu(), d_full.cpu(), None], batch["labels"].cpu(), scale.detach().cpu()))
    assert abs(loss - ref) < 1e-3, (loss, ref)                      # north_star: loss within 1e-3
    assert abs(loss - math.log(NB)) < 0.5, loss                     # near-uniform similarities at random init


def test_b2048_backward_is_the_sum_of_chunk_backwards(dev, big):
    """Parameter gradients are sums over samples: tower backward at M = 403 456 / 272 384 token rows against eight b=256
    passes with the same upstream gradient rows (float-atomic / split order only: 2e-3 relative, cosine 0.99999)."""
    model, batch = big
    ps = _params(model)
    g = torch.Generator(device="cpu").manual_seed(5)
    up_i = (torch.randn(NB, 768, generator=g) * 1e-2).to(dev)
    up_d = (torch.randn(NB, 768, generator=g) * 1e-2).to(dev)

    def grads(sl):
        i, d, _, _, _ = model(batch["image"][sl], batch["dna"][sl], None)
        obj = (i * up_i[sl]).sum() + (d * up_d[sl]).sum()
        gs = torch.autograd.grad(obj, ps, allow_unused=True)
        model.join_streams()
        return [torch.zeros_like(p) if g_ is None else g_.detach().clone() for p, g_ in zip(ps, gs)]

    full = grads(slice(0, NB))
    acc = [torch.zeros_like(p) for p in ps]
    for s in range(0, NB, CH):
        for a, g_ in zip(acc, grads(slice(s, s + CH))):
            a += g_
    torch.cuda.synchronize()
    fa = torch.cat([t.flatten() for t in full]).double()
    ca = torch.cat([t.flatten() for t in acc]).double()
    assert torch.isfinite(fa).all() and fa.abs().max() > 0
    rel = ((fa - ca).norm() / ca.norm()).item()
    cosv = (fa @ ca / (fa.norm() * ca.norm())).item()
    assert rel < 2e-3 and cosv > 0.99999, (rel, cosv)


def test_b2048_training_step_runs_and_learns(dev, big):
    from clibd_amd.train import Trainer

    model, batch = big
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
    losses = [float(tr.step(batch["image"], batch["dna"], None, batch["labels"])) for _ in range(3)]
    assert all(math.isfinite(l) for l in losses)
    assert abs(losses[0] - math.log(NB)) < 0.5 and losses[-1] < losses[0], losses
    assert torch.isfinite(tr.optimizer.flat_p).all()


def test_b2048_train_mode_step_is_finite_and_bit_reproducible(dev, big):
    """The path `bench.py` actually times: the model in TRAIN mode (`train_epoch.py:19`; HF BERT dropout p = 0.1 puts the DNA
    tower on attention_*<10,...,DROP=true> and the dropout epilogues of the 256x256 GEMM at M = 272 384 rows) at the metric's
    batch.  No oracle runs this size, so: finite, |loss - ln 2048| < 0.5 at random init, and — the masks being a pure function
    of (seed, element index) — the embeddings, the loss and (round 6: every sum on the LoRA path has a fixed order) the gradients repeat bit for bit when
    the same base seed is drawn again, and change when another one is drawn."""
    from clibd_amd.model import ClipLoss

    model, batch = big
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    ps = _params(model)
    model.train()
    try:
        def run(seed):
            torch.manual_seed(seed)        # the towers draw their dropout base seed from the CPU generator
            i, d, _, scale, _ = model(batch["image"], batch["dna"], None)
            loss = crit(i, d, None, batch["labels"], scale)
            gs = torch.autograd.grad(loss, ps, allow_unused=True)
            model.join_streams()
            torch.cuda.synchronize()
            g = torch.cat([(torch.zeros_like(p) if g_ is None else g_).flatten() for p, g_ in zip(ps, gs)])
            return i.detach().clone(), d.detach().clone(), float(loss.detach()), g.detach().clone()

        i1, d1, l1, g1 = run(77)
        i2, d2, l2, g2 = run(77)
        i3, d3, l3, _ = run(78)
    finally:
        model.eval()
    assert math.isfinite(l1) and torch.isfinite(i1).all() and torch.isfinite(d1).all() and torch.isfinite(g1).all()
    assert abs(l1 - math.log(NB)) < 0.5, l1
    assert torch.equal(i1, i2) and torch.equal(d1, d2)                       # same seed: the same masks, the same bits
    assert l1 == l2                                                          # round 6: the loss is a fixed-order sum (csrc/loss.hip)
    assert torch.equal(g1, g2)                                               # ... and so is every adapter / head gradient (no float atomics on the LoRA path)
    assert torch.equal(i1, i3)                                               # timm ViT has no dropout: the image rows do not move
    assert not torch.equal(d1, d3) and (d1 - d3).abs().max().item() > 1e-5   # BERT dropout with fresh masks does


# ------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: BIOSCAN-5M-shaped run, per-GPU batch 1024 of a global batch 8192, fp8 MFMA GEMM path
# (reference shape: config/model_config/for_bioscan_5m/final_experiments/image_dna_seed_42.yaml:1-2).  Same property style:
# the oracle cannot run b=1024 in seconds, so the rows are tied to b=256 forwards (which ARE oracle-checked at small batch,
# tests/test_fp8_gpu.py) bit for bit, the rank-local 1024 x 8192 loss block to a CPU statement, and a step must learn.
B5, N5 = 1024, 8192


# fp8 modes of this size: None = bf16; "pooled_ffn" = configs[4]'s RECOMMENDED mode (fp8 forward on the MLP pair of the mean-pooled tower + its 8-bit
# dgrad: training-grade, tests/test_fp8_gpu.py); "all" = every tower, every site (embedding-grade: every fp8 forward kernel at this size)
FP8_MODES = [None, "pooled_ffn", "all"]
FP8_IDS = ["bf16", "fp8-recommended", "fp8-all"]


@pytest.mark.parametrize("fp8", FP8_MODES, ids=FP8_IDS)
def test_b1024_rows_equal_chunked_and_block_loss_matches_cpu(dev, big, fp8):
    from clibd_amd import ops
    from clibd_amd.data import synthetic_batch

    model, batch = big
    img, dna = batch["image"][:B5], batch["dna"][:B5]
    if fp8:
        model.enable_fp8_forward(calibration_inputs=(img, dna, None), towers=fp8)   # per-layer scales from this batch; fixed for every call below
    try:
        with torch.no_grad():
            i_full, d_full, _, scale, _ = model(img, dna, None)
            chunks = [model(img[s : s + CH], dna[s : s + CH], None) for s in range(0, B5, CH)]
            # the other seven ranks' rows of the global batch 8192 (their own synthetic batches)
            others = []
            for r in range(1, N5 // B5):
                ob = synthetic_batch(B5, dev, seed=42, rank=r, with_text=False)
                oi, od, _, _, _ = model(ob["image"], ob["dna"], None)
                others.append((oi, od, ob["labels"]))
        model.join_streams()
        torch.cuda.synchronize()
    finally:
        if fp8:
            model.enable_fp8_forward(enabled=False)
    for full, k in ((i_full, 0), (d_full, 1)):
        ch = torch.cat([c[k] for c in chunks])
        assert torch.isfinite(full).all()
        assert (full - ch).abs().max().item() <= 1e-6, (k, (full - ch).abs().max().item())     # 1024 = 4 x 256, same per-row arithmetic
        assert torch.allclose(full.norm(dim=1), torch.ones(B5, device=dev), atol=1e-4)
    # rank 0's row block of the directed terms image->dna and dna->image over the global batch (what ClipLoss evaluates per rank)
    all_i = torch.cat([i_full] + [o[0] for o in others]).contiguous()
    all_d = torch.cat([d_full] + [o[1] for o in others]).contiguous()
    labels = torch.cat([batch["labels"][:B5]] + [o[2] for o in others]).contiguous()
    assert labels.numel() == N5 and labels.unique().numel() == N5
    labels[5] = labels[4]                      # one duplicate pair inside the block, one across ranks: multi-hot target rows
    labels[B5 + 7] = labels[9]
    scale_t = scale.detach().to(F32).reshape(1).contiguous()
    loss_sum = torch.zeros((1,), dtype=F32, device=dev)
    for x, y in ((i_full, all_d), (d_full, all_i)):
        ws = ops.softce_workspace(B5, N5, 768, dev)
        ops.softce_rows_fwd(x.contiguous(), y, labels, 0, scale_t, loss_sum, ws)
    torch.cuda.synchronize()
    got = float(loss_sum) / (2 * N5)
    xi, xd, ai, ad, lab, sc = i_full.double().cpu(), d_full.double().cpu(), all_i.double().cpu(), all_d.double().cpu(), labels.cpu(), float(scale)
    T = (lab[:B5, None] == lab[None, :]).double()
    ref = 0.0
    for x, y in ((xi, ad), (xd, ai)):
        ref += float(-(T * torch.log_softmax(sc * x @ y.T, dim=1)).sum())
    ref /= 2 * N5
    assert abs(got - ref) < 1e-3 * max(1.0, abs(ref)), (got, ref)          # this rank's share of the N = 8192 loss
    assert 0.0 < got * (N5 // B5) < math.log(N5) + 1.0, got                 # a share of a loss that starts at ln 8192


@pytest.mark.parametrize("fp8", FP8_MODES, ids=FP8_IDS)
def test_b1024_training_step_runs_and_learns(dev, big, fp8):
    from clibd_amd.train import Trainer

    model, batch = big
    img, dna, labels = batch["image"][:B5], batch["dna"][:B5], batch["labels"][:B5]
    if fp8:
        model.enable_fp8_forward(calibration_inputs=(img, dna, None), towers=fp8)    # the selection the trainer's re-calibration keeps
        if fp8 == "pooled_ffn":
            model.enable_fp8_dgrad(towers="pooled")                                  # M = 136 192 DNA token rows: the two-row e4m3 LayerNorm backward, the 8-bit dgrad GEMMs
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True, fp8_recalibrate_every=2 if fp8 else 0)
    try:
        losses = [float(tr.step(img, dna, None, labels)) for _ in range(3)]
        if fp8 == "pooled_ffn":
            num = model.numerics()
            assert num["dna_encoder"]["dgrad"] == "fp8" and num["dna_encoder"]["forward"].startswith("fp8") and num["image_encoder"]["forward"] == "bf16"
    finally:
        model.enable_fp8_dgrad(enabled=False)
        model.enable_fp8_forward(enabled=False)
        for enc in (model.image_encoder, model.dna_encoder):
            enc.tower().grad_sink = None
    assert all(math.isfinite(l) for l in losses)
    assert abs(losses[0] - math.log(B5)) < 0.6, losses
    if fp8 != "all":    # the all-tower mode is embedding-grade (its gradient is not the bf16 step's: tests/test_fp8_gpu.py): it must run, not learn in three steps
        assert losses[-1] < losses[0], losses
    assert torch.isfinite(tr.optimizer.flat_p).all()


def test_gemm_operand_beyond_4gib_takes_the_64bit_kernel(dev):
    """A [M,K] bf16 operand of 4.3 GB: the 256x256 kernel addresses operands with 32-bit byte offsets and must decline
    (csrc/gemm256.hip gemm256_try_launch); the 128x128 kernel takes it.  Exact on integer data."""
    from clibd_amd import ops

    M, N, K = 700_000, 256, 3072
    assert M * K * 2 >= 2 ** 32
    g = torch.Generator().manual_seed(9)
    w = torch.randint(-2, 3, (N, K), generator=g).to(BF16).to(dev)
    a = torch.zeros((M, K), dtype=BF16, device=dev)
    rows = torch.tensor([0, 1, 255, 256, 349_999, 349_525, 349_526, 699_998, 699_999])   # 349 525.33 rows = 2^31 bytes
    vals = torch.randint(-2, 3, (rows.numel(), K), generator=g).to(BF16)
    a[rows.to(dev)] = vals.to(dev)
    out = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a, w, out_bf16=out)
    torch.cuda.synchronize()
    ref = vals.double() @ w.cpu().double().T          # |values| <= 4 * 3072: exact in fp32, rounded once to bf16
    got = out[rows.to(dev)].cpu().double()
    assert torch.equal(got, ref.to(BF16).double())
    assert float(out.double().abs().sum()) == float(got.abs().sum())   # every other row is exactly zero
