"""Module-level parity (GPU): the drop-in `clibd_amd.model` classes on the HIP kernels against
  (1) the committed golden vectors generated from the reference's own classes (fp32 CPU path), and
  (2) the CPU oracle run with the kernels' bf16 rounding points (tight tolerance).
Tolerances: tower outputs / loss within 1e-3 of the bf16-emulating oracle (north-star bf16 tolerance);
gradients rel-L2 <= 2e-2 and cosine >= 0.999 (SURVEY §8d parity gates)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return torch.load(os.path.join(G, name), map_location="cpu", weights_only=False)


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def grads_named(module, loss):
    ps = {n: p for n, p in module.named_parameters() if p.requires_grad}
    gs = torch.autograd.grad(loss, list(ps.values()), allow_unused=True)
    return {n: (torch.zeros_like(p) if g is None else g).detach().cpu() for (n, p), g in zip(ps.items(), gs)}


def assert_grads(got, ref, rel_tol=2e-2, cos_tol=0.999, what="", zero_tol=1e-9, zero_rel=0.0):
    assert sorted(got) == sorted(ref)
    gmax = max(float(r.abs().max()) for r in ref.values())
    for n in ref:
        r = ref[n]
        # analytically zero gradients (e.g. BERT key bias: softmax ignores a per-query constant) are rounding noise on both sides
        if r.abs().max() < max(zero_tol, zero_rel * gmax):
            assert got[n].abs().max() < max(1e-6, 10 * zero_tol, 10 * zero_rel * gmax), (what, n)
            continue
        assert rel(got[n], r) < rel_tol, (what, n, rel(got[n], r))
        assert cos(got[n], r) > cos_tol, (what, n, cos(got[n], r))


# ------------------------------------------------------------------------------------------ builders
def hip_dna(gd, dev):
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder

    c = gd["config"]
    m = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, **c)), r=4, num_classes=128)
    m.load_state_dict(gd["state_dict"], strict=True)
    return m.to(dev).eval()  # goldens: dropout off (eval); train-mode dropout has its own tests below


def hip_text(gt, dev):
    from clibd_amd.model import BertConfigLite, BertModel, CLIBDLanguageEncoder

    m = CLIBDLanguageEncoder(BertModel(BertConfigLite(vocab_size=gt["vocab"], **gt["config"])), r=4, num_classes=128)
    m.load_state_dict(gt["state_dict"], strict=True)
    return m.to(dev).eval()


def hip_image(gi, dev):
    from clibd_amd.model import CLIBDImageEncoder, VisionTransformer

    c = gi["config"]
    m = CLIBDImageEncoder(VisionTransformer(embed_dim=c["dim"], depth=c["depth"], num_heads=c["heads"], num_classes=10), r=4, num_classes=128)
    m.load_state_dict(gi["state_dict"], strict=True)
    return m.to(dev).eval()


def oracle_models():
    from tests.test_oracle import build_dna, build_image, build_text

    return build_dna, build_text, build_image


# ------------------------------------------------------------------------------------------ towers
def test_dna_tower_parity(dev):
    from oracle import clibd_oracle as O

    gd = load("dna_tiny_golden.pt")
    m = hip_dna(gd, dev)
    y = m(gd["ids"].to(dev))
    assert y.shape == (4, 128) and y.dtype == torch.float32
    assert torch.allclose(y.sum(1).cpu(), torch.ones(4), atol=1e-4)
    got = grads_named(m, (y * gd["cot"].to(dev)).sum())
    build_dna, _, _ = oracle_models()
    om = build_dna(gd)
    with O.precision("bf16"):
        yo = om(gd["ids"])
        go = {n: g for n, g in zip([n for n, p in om.named_parameters() if p.requires_grad],
                                   torch.autograd.grad((yo * gd["cot"]).sum(), [p for p in om.parameters() if p.requires_grad]))}
    # (2) bf16-emulating oracle: tight
    assert (y.cpu() - yo.detach()).abs().max().item() < 1e-3 * yo.abs().max().item() + 1e-5
    assert rel(y.cpu(), yo.detach()) < 3e-3
    assert_grads(got, go, what="dna vs oracle-bf16")
    # (1) reference fp32 golden: bf16 tolerance
    assert rel(y.cpu(), gd["out"]) < 2e-2
    assert_grads(got, gd["grads"], rel_tol=5e-2, cos_tol=0.998, what="dna vs reference fp32")


def test_text_tower_parity_with_padding_mask(dev):
    from oracle import clibd_oracle as O

    gt = load("text_tiny_golden.pt")
    m = hip_text(gt, dev)
    x = {k: v.to(dev) for k, v in gt["inputs"].items()}
    y = m(x)
    got = grads_named(m, (y * gt["cot"].to(dev)).sum())
    _, build_text, _ = oracle_models()
    om = build_text(gt)
    with O.precision("bf16"):
        yo = om(gt["inputs"])
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * gt["cot"]).sum(), [p for _, p in ps])))
    assert rel(y.cpu(), yo.detach()) < 3e-3
    assert_grads(got, go, what="text vs oracle-bf16")
    assert rel(y.cpu(), gt["out"]) < 2e-2
    assert_grads(got, gt["grads"], rel_tol=5e-2, cos_tol=0.998, what="text vs reference fp32")


def test_image_tower_parity(dev):
    from oracle import clibd_oracle as O

    gi = load("image_tiny_golden.pt")
    m = hip_image(gi, dev)
    img = gi["image_u8"].float() / 255.0
    y = m(img.to(dev))
    got = grads_named(m, (y * gi["cot"].to(dev)).sum())
    _, _, build_image = oracle_models()
    om = build_image(gi)
    with O.precision("bf16"):
        yo = om(img)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * gi["cot"]).sum(), [p for _, p in ps])))
    assert rel(y.cpu(), yo.detach()) < 3e-3
    assert_grads(got, go, what="image vs oracle-bf16")
    assert rel(y.cpu(), gi["out"]) < 2e-2
    assert_grads(got, gi["grads"], rel_tol=5e-2, cos_tol=0.998, what="image vs reference fp32")


def test_towers_eval_mode_and_no_grad(dev):
    gd = load("dna_tiny_golden.pt")
    m = hip_dna(gd, dev)
    with torch.no_grad():
        y = m(gd["ids"].to(dev))
    assert not y.requires_grad and rel(y.cpu(), gd["out"]) < 2e-2


# ------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("i", range(7))
def test_loss_modules_match_reference_goldens(dev, i):
    from clibd_amd.model import ClipLoss, ContrastiveLoss

    c = load("loss_golden.pt")["loss_cases"][i]
    feats = [None if f is None else f.clone().to(dev).requires_grad_(True) for f in c["features"]]
    ls = c["log_scale"].clone().to(dev).requires_grad_(True)
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss(),
                    bind_to=c["bind_to"], no_image_text_loss=c["no_image_text_loss"])
    loss = crit(feats[0], feats[1], feats[2], c["labels"].to(dev), ls.exp())
    assert abs(loss.item() - float(c["clip_loss"])) < 1e-3
    present = [f for f in feats if f is not None]
    gs = torch.autograd.grad(loss, present + [ls])
    for g, r in zip(gs, c["clip_grads"]):
        assert rel(g.cpu(), r) < 1e-2
        assert cos(g.cpu(), r) > 0.9999
    if "contrastive_loss" in c:
        l2 = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)(feats[0], feats[1], feats[2], c["labels"].to(dev),
                                                                                          ls.exp())
        assert abs(l2.item() - float(c["contrastive_loss"])) < 1e-3


def test_loss_known_answers_and_errors(dev):
    from clibd_amd.model import ContrastiveLoss

    ka = load("loss_golden.pt")["known_answers"]
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    a, b = ka["a"].to(dev), ka["b"].to(dev)
    assert abs(crit(a, b, None, torch.arange(32, device=dev), 1 / 0.07).item() - 3.676485061645508) < 5e-4
    assert abs(crit(a, b, None, (torch.arange(32) // 2).to(dev), None).item() - 7.247870445251465) < 5e-4
    with pytest.raises(ValueError):
        crit(a, None, None, torch.arange(32, device=dev), 1.0)


# ------------------------------------------------------------------------------------------ full step
@pytest.mark.parametrize("tag,use_text", [("id", False), ("idt", True)])
def test_full_step_matches_reference(dev, tag, use_text):
    from clibd_amd.model import ClipLoss, SimpleCLIP
    from oracle import clibd_oracle as O

    gs, gd, gt, gi = load("step_tiny_golden.pt"), load("dna_tiny_golden.pt"), load("text_tiny_golden.pt"), load("image_tiny_golden.pt")
    model = SimpleCLIP(hip_image(gi, dev), hip_dna(gd, dev), hip_text(gt, dev)).to(dev)
    with torch.no_grad():
        model.logit_scale.copy_(gs["logit_scale"])
    assert list(model.state_dict().keys()) == gs["state_dict_keys"]
    img = (gs["image_u8"].float() / 255.0).to(dev)
    text = {k: v.to(dev) for k, v in gs["text"].items()}
    io, do_, to, scale, bias = model(img, gs["dna"].to(dev), text)
    assert bias is None
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    loss = crit(io, do_, to if use_text else None, gs["labels"].to(dev), scale)
    got = grads_named(model, loss)
    # oracle with the kernels' rounding points
    build_dna, build_text, build_image = oracle_models()
    om = O.SimpleCLIP(build_image(gi), build_dna(gd), build_text(gt))
    with torch.no_grad():
        om.logit_scale.copy_(gs["logit_scale"])
    with O.precision("bf16"):
        oi, od, ot, osc, _ = om(gs["image_u8"].float() / 255.0, gs["dna"], gs["text"])
        lo = O.contrastive_loss([oi, od, ot if use_text else None], gs["labels"], osc)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], [torch.zeros_like(p) if g is None else g
                                           for (_, p), g in zip(ps, torch.autograd.grad(lo, [p for _, p in ps], allow_unused=True))]))
    for f, r in zip((io, do_, to), (oi, od, ot)):
        assert (f.cpu() - r.detach()).abs().max().item() < 1e-3          # unit-norm embeddings: abs tolerance
    assert abs(loss.item() - lo.item()) < 1e-3                           # north-star: loss within 1e-3 (bf16 tolerance)
    # End-to-end gradients of the contrastive loss are ill-conditioned under bf16: the logits amplify a 2^-9 embedding
    # perturbation by the temperature (x14.3), so two correct bf16 evaluations that round in a different order (or the
    # oracle's own bf16 vs fp32 modes, 3-8 % per tensor on this fixture) differ by several percent per tensor.  The tight
    # 2e-2 gradient gates are applied where the upstream is identical (tower tests above, loss tests); here: direction.
    assert_grads(got, go, rel_tol=0.2, cos_tol=0.98, what="step vs oracle-bf16")
    allg = torch.cat([got[n].flatten() for n in sorted(go)])
    allo = torch.cat([go[n].flatten() for n in sorted(go)])
    assert rel(allg, allo) < 0.08 and cos(allg, allo) > 0.997
    # reference fp32 goldens
    for f, r in zip((io, do_, to), gs[f"features_{tag}"]):
        assert (f.cpu() - r).abs().max().item() < 5e-3
    assert abs(loss.item() - float(gs[f"loss_{tag}"])) < 2e-2
    assert_grads(got, gs[f"grads_{tag}"], rel_tol=0.25, cos_tol=0.97, what="step vs reference fp32")
    allr = torch.cat([gs[f"grads_{tag}"][n].flatten() for n in sorted(go)])
    assert rel(allg, allr) < 0.1 and cos(allg, allr) > 0.995


# ------------------------------------------------------------------------------------------ BASELINE configs[3]: real-size text tower
def _bert_small_pair(dev, seed=21):
    """prajjwal1/bert-small shapes (language_encoder.py:12-20: L=4, H=512, A=8, FF=2048, vocab 30522), random weights, LoRA B
    non-zero, identical state dicts in the oracle and the HIP module."""
    from oracle import clibd_oracle as O
    from clibd_amd.model import BertConfigLite, BertModel, CLIBDLanguageEncoder
    from clibd_amd.model.language_encoder import BERT_SMALL

    torch.manual_seed(seed)
    om = O.LanguageEncoder(O.BertModel(vocab=30522, hidden=512, layers=4, heads=8, ff=2048), r=4, num_classes=768)
    with torch.no_grad():
        for n, p in om.named_parameters():
            if p.dim() >= 2 and "embeddings" not in n:
                p.normal_(0, 0.03)
            if ".w_b." in n:
                p.normal_(0, 0.02)
    m = CLIBDLanguageEncoder(BertModel(BertConfigLite(**BERT_SMALL)), r=4, num_classes=768)
    m.load_state_dict(om.state_dict(), strict=True)
    return om.eval(), m.to(dev).eval()


def _text_batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 30522, (B, 20), generator=g)
    lens = torch.randint(6, 21, (B,), generator=g)
    return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": (torch.arange(20)[None, :] < lens[:, None]).long()}


def test_bert_small_text_tower_matches_oracle(dev):
    """The real text tower of the tri-modal config (H=512 -> 8 heads of 64, S=20 padded to 32 keys with a key mask, mean over
    all 20 positions including pads, proj 512 -> 768) against the oracle with the kernels' rounding points."""
    from oracle import clibd_oracle as O

    om, m = _bert_small_pair(dev)
    x = _text_batch(64, 5)
    cot = torch.randn(64, 768, generator=torch.Generator().manual_seed(6))
    y = m({k: v.to(dev) for k, v in x.items()})
    got = grads_named(m, (y * cot.to(dev)).sum())
    with O.precision("bf16"):
        yo = om(x)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * cot).sum(), [p for _, p in ps])))
    assert y.shape == (64, 768)
    # 4 layers of H=512 bf16 GEMM operands + the 512 -> 768 projection: 2.2e-3 relative overall, worst element 2.5e-3 of the range
    assert rel(y.cpu(), yo.detach()) < 3e-3 and (y.cpu() - yo.detach()).abs().max().item() < 4e-3 * yo.abs().max().item()
    assert_grads(got, go, what="bert-small vs oracle-bf16")


def test_trimodal_full_size_step_matches_oracle(dev):
    """Image + DNA + Text at the real tower sizes (ViT-B/16, BERT-base 133 tokens, BERT-small 20 tokens), batch 8, 3-way loss
    (six directed terms): embeddings, loss (1e-3) and all trainable gradients against the oracle's bf16 mode."""
    from oracle import clibd_oracle as O
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import ClipLoss, CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

    torch.manual_seed(31)
    base = O.build_image_dna_model()
    with torch.no_grad():
        for n, p in base.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    ot_enc, ht_enc = _bert_small_pair(dev, seed=32)
    om = O.SimpleCLIP(base.image_encoder, base.dna_encoder, ot_enc)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), ht_enc.cpu())
    model.load_state_dict(om.state_dict(), strict=True)
    model = model.to(dev).eval()
    B = 8
    batch = synthetic_batch(B, torch.device("cpu"), seed=9, rank=0, with_text=True)
    labels = torch.tensor([0, 1, 2, 3, 3, 5, 6, 0])
    with O.precision("bf16"):
        oi, od, ot, osc, _ = om(batch["image"], batch["dna"], batch["text"])
        lo = O.contrastive_loss([oi, od, ot], labels, osc)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad(lo, [p for _, p in ps], allow_unused=True)))
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    hi, hd, ht, scale, _ = model(batch["image"].to(dev), batch["dna"].to(dev), {k: v.to(dev) for k, v in batch["text"].items()})
    loss = crit(hi, hd, ht, labels.to(dev), scale)
    got = grads_named(model, loss)
    model.join_streams()
    for a, b_ in ((hi, oi), (hd, od), (ht, ot)):
        assert (a.cpu() - b_.detach()).abs().max().item() < 1e-3, (a.cpu() - b_.detach()).abs().max().item()   # unit-norm rows
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-3
    keys = sorted(n for n, _ in ps if n in got)
    allg = torch.cat([got[n].flatten() for n in keys])
    allo = torch.cat([(torch.zeros_like(dict(ps)[n]) if go[n] is None else go[n]).flatten() for n in keys])
    assert cos(allg, allo) > 0.99 and rel(allg, allo) < 0.15


# ------------------------------------------------------------------------------------------ anchored to the reference's bf16 mode
@pytest.mark.parametrize("name", ["dna", "text", "image"])
def test_towers_as_close_to_fp32_reference_as_reference_autocast(dev, name):
    """tests/golden/autocast_golden.pt = the reference's own towers under torch.autocast(bfloat16) (its optional bf16 mode,
    train_epoch.py:42-46), same weights and inputs as the fp32 goldens.  The HIP towers (bf16 MFMA operands, fp32 residual
    stream and statistics) must be at least as close to the reference's fp32 outputs and gradients as the reference's bf16 mode
    is (x1.25 on outputs, x1.6 on the worst per-tensor gradient error: equivalent but different rounding points), and agree with
    that bf16 mode itself within the sum of the two distances."""
    ac = load("autocast_golden.pt")[name]
    if name == "dna":
        g = load("dna_tiny_golden.pt"); m = hip_dna(g, dev); x = g["ids"].to(dev)
    elif name == "text":
        g = load("text_tiny_golden.pt"); m = hip_text(g, dev); x = {k: v.to(dev) for k, v in g["inputs"].items()}
    else:
        g = load("image_tiny_golden.pt"); m = hip_image(g, dev); x = (g["image_u8"].float() / 255.0).to(dev)
    y = m(x)
    got = grads_named(m, (y * g["cot"].to(dev)).sum())
    err = float((y.detach().cpu() - g["out"]).abs().max())
    assert err <= 1.25 * ac["err_vs_fp32"] + 1e-6, (err, ac["err_vs_fp32"])
    assert float((y.detach().cpu() - ac["out"]).abs().max()) <= err + ac["err_vs_fp32"] + 1e-6
    ours = theirs = 0.0
    for n, r in g["grads"].items():
        if r.abs().max() < 1e-9:
            continue
        ours, theirs = max(ours, rel(got[n], r)), max(theirs, rel(ac["grads"][n], r))
    assert ours <= 1.6 * theirs + 1e-6, (ours, theirs)


@pytest.mark.parametrize("tag,use_text", [("id", False), ("idt", True)])
def test_full_step_loss_as_close_as_reference_autocast(dev, tag, use_text):
    """Loss of the b=8 step fixture: |HIP - reference fp32| <= max(1e-3, 1.5 x |reference bf16-autocast - reference fp32|)."""
    from clibd_amd.model import ClipLoss, SimpleCLIP

    gs, gd, gt, gi, ac = load("step_tiny_golden.pt"), load("dna_tiny_golden.pt"), load("text_tiny_golden.pt"), load("image_tiny_golden.pt"), load("autocast_golden.pt")
    model = SimpleCLIP(hip_image(gi, dev), hip_dna(gd, dev), hip_text(gt, dev)).to(dev)
    with torch.no_grad():
        model.logit_scale.copy_(gs["logit_scale"])
        io, do_, to, scale, _ = model((gs["image_u8"].float() / 255.0).to(dev), gs["dna"].to(dev), {k: v.to(dev) for k, v in gs["text"].items()})
        crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
        loss = float(crit(io, do_, to if use_text else None, gs["labels"].to(dev), scale))
    ref32, ref16 = float(gs[f"loss_{tag}"]), float(ac[f"loss_{tag}"])
    assert abs(loss - ref32) <= max(1e-3, 1.5 * abs(ref16 - ref32)), (loss, ref32, ref16)


# ------------------------------------------------------------------------------------------ dropout (train mode)
def test_bert_train_mode_dropout_matches_oracle_masks(dev):
    """HF BERT applies dropout (p = 0.1) in train mode (train_epoch.py:19 model.train()).  The HIP masks are a pure function
    of (seed, element index); the oracle evaluates the same function, so train-mode parity is exact up to bf16 rounding."""
    from oracle import clibd_oracle as O

    gd = load("dna_tiny_golden.pt")
    m = hip_dna(gd, dev).train()
    torch.manual_seed(1234)
    base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    torch.manual_seed(1234)  # the tower draws the same base seed from the CPU generator
    y = m(gd["ids"].to(dev))
    got = grads_named(m, (y * gd["cot"].to(dev)).sum())
    build_dna, _, _ = oracle_models()
    om = build_dna(gd)
    with O.precision("bf16"), O.dropout(0.1, 0.1, base):
        yo = om(gd["ids"])
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * gd["cot"]).sum(), [p for _, p in ps])))
    assert rel(y.cpu(), yo.detach()) < 4e-3
    assert_grads(got, go, rel_tol=3e-2, cos_tol=0.999, what="dna train-mode vs oracle with identical masks")
    # and it really is different from eval mode
    with torch.no_grad():
        ye = m.eval()(gd["ids"].to(dev))
    assert rel(ye.cpu(), y.detach().cpu()) > 1e-4  # (the DNA head output is a near-uniform softmax mean: small but real change)
    # different seed -> different masks; same seed -> bit-identical output
    m.train()
    torch.manual_seed(1234)
    with torch.no_grad():
        y2 = m(gd["ids"].to(dev))
        y3 = m(gd["ids"].to(dev))
    # (the no_grad forward evaluates GELU through a different but equivalent expression: last-bit differences only)
    # flips an occasional bf16 rounding downstream): same masks -> ~1e-4 apart; fresh masks -> percent-level change
    assert rel(y2.cpu(), y.detach().cpu()) < 2e-3 and rel(y3.cpu(), y2.cpu()) > 5e-3


def test_bert_base_width_train_mode_matches_oracle_masks(dev):
    """The DNA tower at the width the bench times it — 12 layers, H = 768, 12 heads, FF = 3072, S = 133 — in TRAIN mode (HF BERT
    dropout p = 0.1 on the embeddings, the attention probabilities and both dense outputs of every layer: what
    `train_epoch.py:19` model.train() switches on and what `bench.py` measures), batch 16, against the oracle evaluating the
    same counter-based masks.  This is the kernel set of the timed step — attention_*<10,...,DROP=true>, the dropout epilogue
    of the 256x256 GEMM at K = 768 / 3072, the dropout LayerNorm — at full width; the tiny fixture (H = 128) reaches them
    only through the 128x128 kernel.  Embeddings 1e-3 on the unit-scale head output, adapter / decoder
    gradients rel 2e-2, cosine 0.999."""
    from oracle import clibd_oracle as O
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import CLIBDDNAEncoder, load_pre_trained_bioscan_bert

    torch.manual_seed(23)
    om = O.build_image_dna_model().dna_encoder
    with torch.no_grad():
        for n, p in om.named_parameters():
            if ".w_b." in n:
                p.normal_(0, 0.02)      # adapters that do something (B = 0 at initialisation)
    m = CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768)
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.to(dev).train()
    B = 16
    ids = synthetic_batch(B, torch.device("cpu"), seed=9, rank=0, with_text=False)["dna"]
    cot = torch.randn(B, 768, generator=torch.Generator().manual_seed(5))
    torch.manual_seed(4321)
    base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    torch.manual_seed(4321)       # the tower draws the same base seed from the CPU generator
    y = m(ids.to(dev))
    got = grads_named(m, (y * cot.to(dev)).sum())
    with O.precision("bf16"), O.dropout(0.1, 0.1, base):
        yo = om(ids)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * cot).sum(), [p for _, p in ps], allow_unused=True)))
    go = {n: (torch.zeros_like(p) if go[n] is None else go[n]) for n, p in ps}
    # rows of the head output are probability vectors averaged over 133 tokens (sum 1, entries ~ 1/768): compare on the rows'
    # own scale, i.e. after the L2 normalisation SimpleCLIP applies (unit-norm rows, north_star's 1e-3)
    yn, yon = torch.nn.functional.normalize(y.detach().float().cpu(), dim=1), torch.nn.functional.normalize(yo.detach(), dim=1)
    err = (yn - yon).abs().max().item()
    print(f"[BERT-base train mode vs oracle with identical masks] unit-norm embedding error {err:.2e}, rel {rel(y.cpu(), yo.detach()):.2e}")
    assert err < 1e-3
    got = {n: g for n, g in got.items() if n in go}
    assert_grads(got, go, rel_tol=2e-2, cos_tol=0.999, what="BERT-base train mode", zero_rel=1e-6)
    with torch.no_grad():
        ye = m.eval()(ids.to(dev))
    assert rel(ye.cpu(), y.detach().cpu()) > 1e-4     # and the masks did something


def test_text_tower_train_mode_dropout_with_mask(dev):
    from oracle import clibd_oracle as O

    gt = load("text_tiny_golden.pt")
    m = hip_text(gt, dev).train()
    x = {k: v.to(dev) for k, v in gt["inputs"].items()}
    torch.manual_seed(77)
    base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    torch.manual_seed(77)
    y = m(x)
    got = grads_named(m, (y * gt["cot"].to(dev)).sum())
    _, build_text, _ = oracle_models()
    om = build_text(gt)
    with O.precision("bf16"), O.dropout(0.1, 0.1, base):
        yo = om(gt["inputs"])
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * gt["cot"]).sum(), [p for _, p in ps])))
    assert rel(y.cpu(), yo.detach()) < 4e-3
    assert_grads(got, go, rel_tol=3e-2, cos_tol=0.999, what="text train-mode vs oracle with identical masks")


def test_dropout_keep_rate(dev):
    from clibd_amd import ops

    M, N, K = 512, 256, 64
    a = torch.zeros((M, K), dtype=torch.bfloat16, device=dev)
    w = torch.zeros((N, K), dtype=torch.bfloat16, device=dev)
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    ops.gemm_nt(a, w, bias=torch.ones(N, device=dev), out_f32=out, drop=ops.Drop(0.1, 12345))
    torch.cuda.synchronize()
    o = out.cpu()
    keep = (o != 0).float().mean().item()
    assert abs(keep - 0.9) < 5e-3
    assert torch.allclose(o[o != 0], torch.tensor(1.0 / (1.0 - round(0.1 * 65536) / 65536)))


# ------------------------------------------------------------------------------------------ full fine-tune
def _full_grads(m, om, run_hip, run_oracle, cot, dev, ctx):
    for p in m.parameters():
        p.requires_grad_(True)
    for p in om.parameters():
        p.requires_grad_(True)
    y = run_hip(m)
    got = grads_named(m, (y * cot.to(dev)).sum())
    with ctx:
        yo = run_oracle(om)
        ps = [(n, p) for n, p in om.named_parameters()]
        gs = torch.autograd.grad((yo * cot).sum(), [p for _, p in ps], allow_unused=True)
        go = {n: (torch.zeros_like(p) if g is None else g) for (n, p), g in zip(ps, gs)}
    return y, yo, got, go


@pytest.mark.parametrize("train_mode", [False, True])
def test_dna_tower_full_finetune_gradients(dev, train_mode):
    """model_config.disable_lora: every base weight, bias, LayerNorm and embedding parameter of the BERT tower gets a gradient
    (weight gradients through the NT GEMM on transposed operands, the rest through the paramgrad kernels); with the HF
    train-mode dropout masks in the second case (embedding dropout sits in front of the embedding-table gradients)."""
    import contextlib
    from oracle import clibd_oracle as O

    gd = load("dna_tiny_golden.pt")
    m = hip_dna(gd, dev)
    build_dna, _, _ = oracle_models()
    om = build_dna(gd)
    stack = contextlib.ExitStack()
    stack.enter_context(O.precision("bf16"))
    if train_mode:
        m.train()
        torch.manual_seed(4321)
        base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        torch.manual_seed(4321)
        stack.enter_context(O.dropout(0.1, 0.1, base))
    y, yo, got, go = _full_grads(m, om, lambda mm: mm(gd["ids"].to(dev)), lambda oo: oo(gd["ids"]), gd["cot"], dev, stack)
    assert len(got) > 40 and rel(y.cpu(), yo.detach()) < 4e-3
    assert_grads(got, go, rel_tol=3e-2, cos_tol=0.999, what=f"dna full fine-tune (train={train_mode})", zero_tol=2e-6, zero_rel=1e-4)


def test_text_tower_full_finetune_gradients(dev):
    from oracle import clibd_oracle as O

    gt = load("text_tiny_golden.pt")
    m = hip_text(gt, dev)
    _, build_text, _ = oracle_models()
    om = build_text(gt)
    x = {k: v.to(dev) for k, v in gt["inputs"].items()}
    y, yo, got, go = _full_grads(m, om, lambda mm: mm(x), lambda oo: oo(gt["inputs"]), gt["cot"], dev, O.precision("bf16"))
    assert rel(y.cpu(), yo.detach()) < 4e-3
    assert_grads(got, go, rel_tol=3e-2, cos_tol=0.999, what="text full fine-tune", zero_tol=2e-6, zero_rel=1e-4)


def test_image_tower_full_finetune_gradients(dev):
    """ViT: LoRA adapters AND base weights trainable (the reference's image tower keeps its adapters under disable_lora,
    image_encoder.py:54-57), plus patch embedding, class token, position embedding and the final norm."""
    from oracle import clibd_oracle as O

    gi = load("image_tiny_golden.pt")
    m = hip_image(gi, dev)
    _, _, build_image = oracle_models()
    om = build_image(gi)
    img = gi["image_u8"].float() / 255.0
    y, yo, got, go = _full_grads(m, om, lambda mm: mm(img.to(dev)), lambda oo: oo(img), gi["cot"], dev, O.precision("bf16"))
    assert rel(y.cpu(), yo.detach()) < 4e-3
    assert_grads(got, go, rel_tol=3e-2, cos_tol=0.999, what="image full fine-tune", zero_tol=2e-6, zero_rel=1e-4)


def test_full_finetune_gradients_through_the_tn_kernel(dev):
    """Token counts that are multiples of 128 with widths that are multiples of 256 take the rows-contracting weight-gradient
    kernel (gemm256_tn.hip: dy and x read in place, bias gradient from the all-ones MFMA) instead of the transposing path the
    tiny fixtures exercise: a 2-block ViT (width 256, 4 heads, batch 128 -> 25 216 token rows) against the oracle, every
    parameter."""
    from oracle import clibd_oracle as O
    from clibd_amd import ops
    from clibd_amd.model import CLIBDImageEncoder, VisionTransformer

    torch.manual_seed(9)
    om = O.ImageEncoder(O.VisionTransformer(img_size=224, patch=16, dim=256, depth=2, heads=4, num_classes=0), 4, 256)
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n:
                p.normal_(0, 0.02)
    m = CLIBDImageEncoder(VisionTransformer(embed_dim=256, depth=2, num_heads=4, num_classes=0), r=4, num_classes=256)
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.to(dev).eval()
    calls = []
    inner = ops.gemm_tn_splitk

    def spy(a, b, out, accumulate=True, colsum=None):
        ok = inner(a, b, out, accumulate=accumulate, colsum=colsum)
        calls.append((tuple(a.shape), tuple(b.shape), colsum is not None, ok))
        return ok

    ops.gemm_tn_splitk = spy
    try:
        g = torch.Generator().manual_seed(10)
        img = torch.rand(128, 3, 224, 224, generator=g)
        cot = torch.randn(128, 256, generator=g)
        y, yo, got, go = _full_grads(m, om, lambda mm: mm(img.to(dev)), lambda oo: oo(img), cot, dev, O.precision("bf16"))
    finally:
        ops.gemm_tn_splitk = inner
    big = [c for c in calls if c[0][0] == 128 * 197]
    assert len(big) == 5 and all(c[3] for c in calls) and all(c[2] for c in big)     # qkv, proj, fc1, fc2 of block 0 + qkv of the class-row-only block 1, each with its bias
    assert rel(y.cpu(), yo.detach()) < 4e-3
    # 4e-2: the q adapter of the class-row-only block sees 128 query rows (its gradient is small and bf16-noisy: 3.2e-2 here)
    assert_grads(got, go, rel_tol=4e-2, cos_tol=0.999, what="image full fine-tune (TN weight gradients)", zero_tol=2e-6, zero_rel=1e-4)
    base = [n for n in got if ".attn.qkv.qkv.weight" in n or ".mlp.fc" in n or ".attn.proj." in n or "patch_embed" in n]
    assert len(base) >= 14
    for n in base:                                   # the matrices the TN kernel produced (and their biases): well inside the gate
        assert rel(got[n], go[n]) < 1.5e-2, (n, rel(got[n], go[n]))


def test_full_finetune_training_step_updates_base_weights(dev):
    """One fused-optimizer step in full fine-tune mode: base weights move, the next forward sees the new weights (the bf16
    weight images are rebuilt every step because the in-place optimizer does not bump tensor versions)."""
    from clibd_amd.model import SimpleCLIP
    from clibd_amd.train import Trainer

    gd, gi = load("dna_tiny_golden.pt"), load("image_tiny_golden.pt")
    model = SimpleCLIP(hip_image(gi, dev), hip_dna(gd, dev), None).to(dev)
    for p in model.parameters():
        p.requires_grad_(True)
    tr = Trainer(model, lr=1e-4, world_size=1, rank=0, all_gather=True)
    img = (gi["image_u8"].float() / 255.0).to(dev)
    ids = gd["ids"].to(dev)
    B = min(img.shape[0], ids.shape[0])
    img, ids = img[:B], ids[:B]
    labels = torch.arange(B, device=dev)
    w = model.dna_encoder.base_dna_encoder.bert.encoder.layer[0].intermediate.dense.weight
    before = w.detach().clone()
    l0 = float(tr.step(img, ids, None, labels))
    assert (w.detach() - before).abs().max().item() > 0
    losses = [l0] + [float(tr.step(img, ids, None, labels)) for _ in range(7)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]
    # gradient-ready protocol of the towers (the trainer's hook for the bucketed all-reduce at world_size > 1): head, then the
    # layers from the top down, then "everything" — and at each call the gradients of that group are already in the bucket
    seen = {}
    for name, enc in (("image", model.image_encoder), ("dna", model.dna_encoder)):
        tw = enc.tower()
        groups = tw.grad_groups()
        log = seen.setdefault(name, [])

        def hook(k, tw=tw, groups=groups, log=log):
            if k is not None and k > 0:   # layer groups: their weight gradients must be non-zero when the group is reported
                g = groups[k][2]          # a weight matrix of that layer (ViT: proj_w; BERT: the value projection)
                log.append((k, float(tw.grad_sink[id(g)].abs().sum()) > 0))
            else:
                log.append((k, True))
        tw.on_grads_ready = hook
    tr.step(img, ids, None, labels)
    for name, enc in (("image", model.image_encoder), ("dna", model.dna_encoder)):
        nl = len(enc.tower().stack.layers)
        assert [k for k, _ in seen[name]] == list(range(nl + 1)) + [None], seen[name]
        assert all(ok for _, ok in seen[name])
        enc.tower().on_grads_ready = None


# ------------------------------------------------------------------------------------------ streams / full-size properties
def test_tower_streams_do_not_change_results(dev):
    """The towers run on separate HIP streams by default (forward and, through autograd, backward); serialized on one stream
    they must give the same embeddings, loss and gradients (only float atomics reorder: 1e-5)."""
    from clibd_amd.model import ClipLoss, SimpleCLIP

    gd, gi, gt = load("dna_tiny_golden.pt"), load("image_tiny_golden.pt"), load("text_tiny_golden.pt")
    img = (gi["image_u8"].float() / 255.0).to(dev)
    ids = gd["ids"].to(dev)
    txt = {k: v.to(dev) for k, v in gt["inputs"].items()}
    B = min(img.shape[0], ids.shape[0], txt["input_ids"].shape[0])
    img, ids, txt = img[:B], ids[:B], {k: v[:B] for k, v in txt.items()}
    labels = torch.arange(B, device=dev) % 3
    res = []
    for overlap in (True, False):
        model = SimpleCLIP(hip_image(gi, dev), hip_dna(gd, dev), hip_text(gt, dev)).to(dev)
        model.overlap_towers = overlap
        crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
        i, d, t, scale, _ = model(img, ids, txt)
        loss = crit(i, d, t, labels, scale)
        grads = grads_named(model, loss)
        model.join_streams()
        torch.cuda.synchronize()
        res.append((i.detach().cpu(), d.detach().cpu(), t.detach().cpu(), float(loss.detach()), grads))
    a, b = res
    for k in range(3):
        assert torch.equal(a[k], b[k])
    assert abs(a[3] - b[3]) <= 1e-6 * abs(b[3])   # the loss sum is a float-atomic reduction
    for n in a[4]:
        assert (a[4][n] - b[4][n]).abs().max().item() <= 1e-5 * (b[4][n].abs().max().item() + 1e-12), n


def test_full_size_step_properties(dev):
    """BASELINE configs[1] shapes (ViT-B/16 + BERT-base, 197 / 133 tokens) at a small batch: the step runs through the
    full-size kernels (256x256 GEMM, long-sequence attention).  Size-independent properties: unit-norm embeddings, DNA head
    rows sum to 1 before normalisation is lost (finite, positive), loss = log(B) +- small at random init with unit
    temperature scaling, identical loss when the step is repeated from the same state, loss decreases over a few steps."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert
    from clibd_amd.train import Trainer

    torch.manual_seed(7)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None).to(dev).eval()
    B = 48
    batch = synthetic_batch(B, dev, seed=3, rank=0, with_text=False)
    with torch.no_grad():
        i1, d1, _, scale, _ = model(batch["image"], batch["dna"], None)
        i2, d2, _, _, _ = model(batch["image"], batch["dna"], None)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)                       # forward is deterministic
    for e in (i1, d1):
        assert torch.isfinite(e).all() and torch.allclose(e.norm(dim=1), torch.ones(B, device=dev), atol=1e-4)
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
    losses = [float(tr.step(batch["image"], batch["dna"], None, batch["labels"])) for _ in range(6)]
    assert all(l == l and l < 1e4 for l in losses)
    assert abs(losses[0] - float(torch.log(torch.tensor(float(B))))) < 0.5   # near-uniform similarities at init
    assert losses[-1] < losses[0]


def test_full_size_model_matches_oracle(dev):
    """ViT-B/16 + BERT-base(133 tokens) at batch 16, random weights, LoRA B matrices non-zero: the full-size HIP path (long-
    sequence attention, both GEMM kernels: 3152 token rows put the fc1 / fc2-dgrad products on the 256x256 kernel) against
    the CPU oracle with the kernels' rounding points: embeddings, loss and the adapter / head gradients."""
    from oracle import clibd_oracle as O
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import ClipLoss, CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

    torch.manual_seed(11)
    om = O.build_image_dna_model()
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
    model.load_state_dict(om.state_dict(), strict=True)
    model = model.to(dev).eval()
    B = 16
    batch = synthetic_batch(B, torch.device("cpu"), seed=5, rank=0, with_text=False)
    labels = torch.arange(B) % 11
    with O.precision("bf16"):
        oi, od, _, osc, _ = om(batch["image"], batch["dna"], None)
        lo = O.contrastive_loss([oi, od, None], labels, osc)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad(lo, [p for _, p in ps], allow_unused=True)))
    with torch.no_grad():   # the oracle's fp32 mode = the reference's default (non-autocast) arithmetic
        fi, fd, _, fsc, _ = om(batch["image"], batch["dna"], None)
        lf = O.contrastive_loss([fi, fd, None], labels, fsc)
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    hi, hd, _, scale, _ = model(batch["image"].to(dev), batch["dna"].to(dev), None)
    loss = crit(hi, hd, None, labels.to(dev), scale)
    got = grads_named(model, loss)
    model.join_streams()
    # north_star: "logits/loss within 1e-3 (bf16 tolerance)".  Measured at this size (tools/full_size_errors.py): embeddings
    # 5.7e-4 / 1.3e-4 from the bf16 oracle and 9.6e-4 / 3.1e-4 from the FP32 path; loss 6e-5 from either.
    for a, b_, f_ in ((hi, oi, fi), (hd, od, fd)):
        assert (a.cpu() - b_.detach()).abs().max().item() < 1e-3           # unit-norm rows; vs the kernels' rounding points
        assert (a.cpu() - f_).abs().max().item() < 1.5e-3                  # vs fp32 (the oracle's own bf16 mode: 1.0e-3)
    sim_h, sim_o = (hi.detach().cpu() @ hd.detach().cpu().T), (oi.detach() @ od.detach().T)
    assert (sim_h - sim_o).abs().max().item() < 1e-3                       # cosine logits (before the temperature)
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-3 and abs(float(loss.detach()) - float(lf)) < 1e-3
    allg = torch.cat([got[n].flatten() for n in sorted(got)])
    allo = torch.cat([(torch.zeros_like(p) if go[n] is None else go[n]).flatten() for n, p in sorted(ps)])
    assert sorted(got) == sorted(n for n, _ in ps)
    assert cos(allg, allo) > 0.99 and rel(allg, allo) < 0.15   # 24 layers of bf16 rounding under the x14.3 temperature


def test_device_prefetcher_delivers_the_batches_in_order(dev):
    """clibd_amd.data.DevicePrefetcher: batch i+1 is copied on a side stream while batch i is in use; contents and order must
    be exactly the host iterator's (nested dict / tuple structure preserved, non-tensor leaves passed through)."""
    from clibd_amd.data import DevicePrefetcher

    g = torch.Generator().manual_seed(3)
    host = [{"image": torch.rand(8, 3, 32, 32, generator=g).pin_memory(), "dna": torch.randint(0, 1027, (8, 133), generator=g).pin_memory(),
             "text": {"input_ids": torch.randint(0, 100, (8, 20), generator=g)}, "ids": [f"s{i}"], "pair": (torch.full((4,), float(i)), i)}
            for i in range(5)]
    seen = 0
    for i, bt in enumerate(DevicePrefetcher(iter(host), dev)):
        y = bt["image"] * 2.0       # use it on the compute stream right away
        assert bt["image"].is_cuda and torch.equal(bt["image"].cpu(), host[i]["image"]) and torch.equal(bt["dna"].cpu(), host[i]["dna"])
        assert torch.equal(bt["text"]["input_ids"].cpu(), host[i]["text"]["input_ids"]) and bt["ids"] == [f"s{i}"]
        assert torch.equal(bt["pair"][0].cpu(), host[i]["pair"][0]) and bt["pair"][1] == i and torch.equal(y.cpu(), host[i]["image"] * 2.0)
        seen += 1
    assert seen == 5


def test_image_tower_takes_the_datasets_image_bytes(dev):
    """uint8 images through CLIBDImageEncoder.forward (what DevicePrefetcher hands over when the loader yields the HDF5 bytes):
    embeddings AND adapter / head gradients equal those of the fp32 u8 / 255 tensor the reference's ToTensor would have produced,
    bit for bit — only the patch gather differs, and it reads the same values."""
    gi = load("image_tiny_golden.pt")
    hm = hip_image(gi, dev)
    img8 = gi["image_u8"]
    assert img8.dtype == torch.uint8
    cot = torch.randn(img8.shape[0], hm(img8.to(dev)).shape[1], generator=torch.Generator().manual_seed(2)).to(dev)
    y8 = hm(img8.to(dev))
    g8 = grads_named(hm, (y8 * cot).sum())
    y32 = hm((img8.float() / 255.0).to(dev))
    g32 = grads_named(hm, (y32 * cot).sum())
    assert torch.equal(y8, y32) and len(g8) == len(g32) > 4
    for n in g32:
        assert rel(g8[n], g32[n]) < 2e-5, n   # float-atomic order of the adapter sums only


@pytest.mark.parametrize("name,dim,depth,heads", [("vit_large_patch16_224", 1024, 24, 16), ("vit_small_patch16_224", 384, 12, 6)])
def test_other_vit_sizes_match_oracle(dev, name, dim, depth, heads):
    """The reference also ships configs on other timm ViTs (`pre_train_model`, simple_clip.py:148-153; e.g.
    without_open_clip_vit_large_patch16_224.yaml: H = 1024, 16 heads, 24 blocks, FF = 4096).  Same kernels, other shapes:
    embeddings within 1e-3 of the oracle's bf16 mode at batch 4 (ViT-L: batch 2), adapter / head gradients within the tower gates."""
    from oracle import clibd_oracle as O
    from clibd_amd.model import CLIBDImageEncoder, create_vit

    torch.manual_seed(5)
    om = O.ImageEncoder(O.VisionTransformer(img_size=224, patch=16, dim=dim, depth=depth, heads=heads, num_classes=0), 4, 512)
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n:
                p.normal_(0, 0.02)
    m = CLIBDImageEncoder(create_vit(name), r=4, num_classes=512)
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(6)
    nb = 4 if depth <= 12 else 2      # (the 24-block oracle is the suite's second slowest test: batch 2 there)
    img = torch.rand(nb, 3, 224, 224, generator=g)
    cot = torch.randn(nb, 512, generator=g)
    y = m(img.to(dev))
    got = grads_named(m, (y * cot.to(dev)).sum())
    with O.precision("bf16"):
        yo = om(img)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * cot).sum(), [p for _, p in ps])))
    # un-normalised head outputs; 24 blocks of bf16 operand rounding put ViT-L at 3.2e-3 relative (ViT-B: 2e-3)
    assert rel(y.cpu(), yo.detach()) < 6e-3 and (y.cpu() - yo.detach()).abs().max().item() < 8e-3 * yo.abs().max().item()
    # q-adapter gradients are the ill-conditioned ones (near-uniform attention at random init: dP - delta cancels):
    # the small ones of the late blocks differ by 5-20 % of THEIR norm between two correct bf16 evaluations at batch 4.  Gates:
    # all trainable gradients together (2e-2, cos 0.9995), and every tensor's error against the gradient's overall scale.
    keys = sorted(go)
    assert sorted(got) == keys
    allg, allo = torch.cat([got[n].flatten() for n in keys]), torch.cat([go[n].flatten() for n in keys])
    assert rel(allg, allo) < 2e-2 and cos(allg, allo) > 0.9995, (rel(allg, allo), cos(allg, allo))
    per_tensor_scale = allo.double().norm().item() / len(keys) ** 0.5
    worst = max(((got[n].double() - go[n].double()).norm().item() / per_tensor_scale, n) for n in keys)
    assert worst[0] < 5e-2, worst


@pytest.mark.parametrize("r", [1, 2, 3, 5, 6, 8, 9, 16, 19])
def test_lora_ranks_other_than_four(dev, r):
    """The reference accepts any r > 0 (image_encoder.py:53, dna_encoder.py:84); its configs use 4.  Ranks 1-3 run zero-padded on
    the rank-4 kernels, ranks above 4 as ceil(r / 4) rank-(4+4) slots (engine._rank_slots: every slot after the first takes passes
    of its own for its down-projection, rank update and gradients; round 4: no upper limit — 9, 16 and 19 here): outputs and all
    adapter gradients (shapes [r,H] / [H,r]) against the oracle, both tower kinds."""
    from oracle import clibd_oracle as O
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder, CLIBDImageEncoder, VisionTransformer

    torch.manual_seed(40 + r)
    g = torch.Generator().manual_seed(r)
    od = O.DNAEncoder(O.BertForMaskedLM(vocab=1027, hidden=128, layers=2, heads=2, ff=256), r=r, num_classes=128)
    oi = O.ImageEncoder(O.VisionTransformer(img_size=224, patch=16, dim=128, depth=2, heads=2, num_classes=10), r, 128)
    with torch.no_grad():
        for om in (od, oi):
            for n, p in om.named_parameters():
                if p.dim() >= 2:
                    p.normal_(0, 0.05)
    hd = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                                        intermediate_size=256)), r=r, num_classes=128)
    hi = CLIBDImageEncoder(VisionTransformer(embed_dim=128, depth=2, num_heads=2, num_classes=10), r=r, num_classes=128)
    hd.load_state_dict(od.state_dict(), strict=True)
    hi.load_state_dict(oi.state_dict(), strict=True)
    ids = torch.cat([torch.zeros(3, 1, dtype=torch.long), torch.randint(3, 1027, (3, 132), generator=g)], dim=1)
    img = torch.rand(3, 3, 224, 224, generator=g)
    for hm, om, x in ((hd.to(dev).eval(), od.eval(), ids), (hi.to(dev).eval(), oi.eval(), img)):
        cot = torch.randn(3, 128, generator=g)
        y = hm(x.to(dev))
        got = grads_named(hm, (y * cot.to(dev)).sum())
        with O.precision("bf16"):
            yo = om(x)
            ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
            go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * cot).sum(), [p for _, p in ps])))
        assert rel(y.cpu(), yo.detach()) < 3e-3
        assert any(tuple(v.shape) in ((r, 128), (128, r)) for v in got.values())
        assert_grads(got, go, rel_tol=5e-2, cos_tol=0.998, what=f"r={r}")   # random N(0, 0.05) weights, batch 3: single tensors reach 3 %


def test_training_trajectory_matches_oracle(dev):
    """Ten optimizer steps of the b=8 tri-modal fixture: `Trainer.step` on the HIP path (fused AdamW on the flat bucket) against
    the oracle's bf16 mode trained with torch.optim.AdamW on the CPU from the same weights — the whole loop of train_epoch.py:21-63
    (forward, loss, backward, AdamW), not a single evaluation: per-step losses within 2e-3, final adapter / head parameters on
    the oracle's direction (update cosine > 0.98)."""
    from clibd_amd.model import SimpleCLIP
    from clibd_amd.train import Trainer
    from oracle import clibd_oracle as O

    gs, gd, gt, gi = load("step_tiny_golden.pt"), load("dna_tiny_golden.pt"), load("text_tiny_golden.pt"), load("image_tiny_golden.pt")
    model = SimpleCLIP(hip_image(gi, dev), hip_dna(gd, dev), hip_text(gt, dev)).to(dev)   # eval(): dropout off on both sides
    build_dna, build_text, build_image = oracle_models()
    om = O.SimpleCLIP(build_image(gi), build_dna(gd), build_text(gt))
    with torch.no_grad():
        model.logit_scale.copy_(gs["logit_scale"])
        om.logit_scale.copy_(gs["logit_scale"])
    start = {n: p.detach().clone() for n, p in om.named_parameters() if p.requires_grad}
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
    oopt = torch.optim.AdamW([p for p in om.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-2)
    img = gs["image_u8"].float() / 255.0
    text_d = {k: v.to(dev) for k, v in gs["text"].items()}
    hl, ol = [], []
    for _ in range(10):
        hl.append(float(tr.step(img.to(dev), gs["dna"].to(dev), text_d, gs["labels"].to(dev))))
        oopt.zero_grad()
        with O.precision("bf16"):
            oi, od, ot, osc, _ = om(img, gs["dna"], gs["text"])
            lo = O.contrastive_loss([oi, od, ot], gs["labels"], osc)
        lo.backward()
        oopt.step()
        ol.append(float(lo.detach()))
    assert ol[-1] < ol[0] - 0.05                                         # it trains
    assert max(abs(a - b) for a, b in zip(hl, ol)) < 2e-3, list(zip(hl, ol))
    hp = {n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad}
    du_h = torch.cat([(hp[n] - start[n]).flatten() for n in sorted(start)])
    du_o = torch.cat([(dict(om.named_parameters())[n].detach() - start[n]).flatten() for n in sorted(start)])
    assert cos(du_h, du_o) > 0.98 and rel(du_h, du_o) < 0.2, (cos(du_h, du_o), rel(du_h, du_o))


def test_get_feature_and_label_matches_the_reference_loop(dev):
    """clibd_amd.eval.get_feature_and_label against the oracle running the reference's loop body
    (epoch/inference_epoch.py:56-96: eval mode, no_grad, model(...), F.normalize, extend the label / file-name lists) on the
    reference's 7-tuple batches — including a batch whose barcodes arrive as raw strings, a ragged last batch, and the
    device-tensor return feeding make_prediction without a host round trip."""
    import random

    from clibd_amd.eval import LEVELS, get_feature_and_label, make_prediction
    from clibd_amd.model import SimpleCLIP
    from oracle import clibd_oracle as O
    from tests.test_oracle import build_dna, build_image, build_text

    gd, gt, gi = load("dna_tiny_golden.pt"), load("text_tiny_golden.pt"), load("image_tiny_golden.pt")
    model = SimpleCLIP(hip_image(gi, dev), hip_dna(gd, dev), hip_text(gt, dev)).to(dev)
    om = O.SimpleCLIP(build_image(gi), build_dna(gd), build_text(gt))
    rnd = random.Random(4)
    g = torch.Generator().manual_seed(8)
    batches, sizes = [], (5, 5, 3)
    for bi, n in enumerate(sizes):
        seqs = ["".join(rnd.choice("ACGT") for _ in range(rnd.choice((400, 660, 700)))) for _ in range(n)]
        dna = seqs if bi == 1 else torch.tensor([O.kmer_tokenize(s) for s in seqs])
        ids = torch.randint(0, gt["vocab"], (n, 20), generator=g)
        lens = torch.randint(6, 21, (n,), generator=g)
        labels = {lv: [f"{lv}_{bi}_{i}" for i in range(n)] for lv in LEVELS}
        batches.append(([f"P{bi}_{i}" for i in range(n)], torch.rand(n, 3, 224, 224, generator=g), dna, ids, torch.zeros_like(ids),
                        (torch.arange(20)[None] < lens[:, None]).long(), labels, seqs))
    loader = [b[:7] for b in batches]
    model.train()     # the loop must switch to eval (dropout off) and restore the mode afterwards
    names, fi, fd, ft, lab = get_feature_and_label(loader, model, dev)
    assert model.training
    assert names == [f"P{bi}_{i}" for bi, n in enumerate(sizes) for i in range(n)]
    assert lab[6] == {lv: f"{lv}_1_1" for lv in LEVELS} and len(lab) == sum(sizes)
    ref = ([], [], [])
    with torch.no_grad(), O.precision("bf16"):
        for b in batches:
            ids = torch.tensor([O.kmer_tokenize(s) for s in b[7]])
            outs = om(b[1], ids, {"input_ids": b[3], "token_type_ids": b[4], "attention_mask": b[5]})[:3]
            for store, o in zip(ref, outs):
                store.append(torch.nn.functional.normalize(o, dim=-1))
    for got, want in zip((fi, fd, ft), ref):
        want = torch.cat(want)
        assert got.dtype.name == "float32" and got.shape == tuple(want.shape)
        assert (torch.from_numpy(got) - want).abs().max().item() < 2e-3
    _, ti, td, _, _ = get_feature_and_label(loader, model, dev, as_numpy=False)
    assert ti.is_cuda and torch.equal(ti.cpu(), torch.from_numpy(fi))
    pred, idx = make_prediction(ti, td, lab, with_indices=True, max_k=3)      # image queries against the DNA keys, on the device
    _, oidx = O.topk_inner_product(torch.from_numpy(fi), torch.from_numpy(fd), k=3)
    assert torch.equal(torch.from_numpy(idx), oidx) and pred[0]["genus"][0] == lab[int(oidx[0, 0])]["genus"]


def test_single_pass_attention_backward_in_the_towers(dev, monkeypatch):
    """numerics attn_bwd="sp" (opt-in; CLIBD_ATTN_BWD=sp as the construction default): the towers' training forward saves lse / output residual and the backward takes the single-pass
    attention kernel (the class-row-only last ViT block and masked sequences keep the two-phase one).  Same embeddings bit for
    bit, gradients equal to the default path's up to the kernels' rounding points, and still within the oracle gates."""
    from oracle import clibd_oracle as O

    gd, gi = load("dna_tiny_golden.pt"), load("image_tiny_golden.pt")
    from tests.test_oracle import build_dna, build_image

    for hm, om, x in ((hip_image(gi, dev), build_image(gi), gi["image_u8"].float() / 255.0), (hip_dna(gd, dev), build_dna(gd), gd["ids"])):
        g = torch.Generator().manual_seed(3)
        res = {}
        for mode in ("2phase", "sp"):
            hm.tower().stack.set_numerics(attn_bwd=mode)
            y = hm(x.to(dev))
            if mode == "2phase":
                cot = torch.randn(y.shape, generator=g)
            res[mode] = (y.detach().cpu(), grads_named(hm, (y * cot.to(dev)).sum()))
        hm.tower().stack.set_numerics(attn_bwd="2phase")
        assert torch.equal(res["sp"][0], res["2phase"][0])
        assert_grads(res["sp"][1], res["2phase"][1], rel_tol=2e-2, cos_tol=0.9995, what="sp vs two-phase")
        with O.precision("bf16"):
            yo = om(x)
            ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
            go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * cot).sum(), [p for _, p in ps])))
        assert_grads(res["sp"][1], go, rel_tol=2e-2, cos_tol=0.999, what="sp vs oracle")


def test_gelu_grad_kept_as_one_byte_in_the_towers(dev, monkeypatch):
    """numerics gelu_grad="u8" (opt-in; CLIBD_GELU_GRAD=u8 as the construction default): the fc1 epilogue keeps gelu' as one byte per element and the fc2 dgrad decodes it.
    Forward untouched (embeddings bit for bit), gradients equal to the default path's within the stated quantisation budget and still
    within the oracle gates, both tower kinds; the full fine-tune walk uses the same two epilogues."""
    from oracle import clibd_oracle as O

    gd, gi = load("dna_tiny_golden.pt"), load("image_tiny_golden.pt")
    from tests.test_oracle import build_dna, build_image

    for hm, om, x in ((hip_image(gi, dev), build_image(gi), gi["image_u8"].float() / 255.0), (hip_dna(gd, dev), build_dna(gd), gd["ids"])):
        g = torch.Generator().manual_seed(5)
        res = {}
        for mode in ("bf16", "u8"):
            hm.tower().stack.set_numerics(gelu_grad=mode)
            y = hm(x.to(dev))
            if mode == "bf16":
                cot = torch.randn(y.shape, generator=g)
            res[mode] = (y.detach().cpu(), grads_named(hm, (y * cot.to(dev)).sum()))
        hm.tower().stack.set_numerics(gelu_grad="bf16")
        assert torch.equal(res["u8"][0], res["bf16"][0])
        assert any(not torch.equal(res["u8"][1][n], res["bf16"][1][n]) for n in res["bf16"][1]), "the knob did not reach the kernels"
        assert_grads(res["u8"][1], res["bf16"][1], rel_tol=2e-2, cos_tol=0.9995, what="u8 vs bf16 gelu'")
        with O.precision("bf16"):
            yo = om(x)
            ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
            go = dict(zip([n for n, _ in ps], torch.autograd.grad((yo * cot).sum(), [p for _, p in ps])))
        # (the most sensitive tensor of the tiny ViT, the block-0 q adapter, sits at 1.7-2.5 % with either form depending on the other knobs)
        assert_grads(res["u8"][1], go, rel_tol=3e-2, cos_tol=0.999, what="u8 gelu' vs oracle")


def test_gelu_grad_e4m7_towers_equal_the_bf16_form(dev):
    """numerics gelu_grad="e4m7" (round 6; opt-in: free numerically, 3 % slower — engine.NUMERICS_CHOICES): gelu' saved as the twelve-bit form of its bf16 value.  Every |gelu'| >= 2^-14 is reproduced
    bit for bit, so the tower gradients are those of the bf16 form up to the flushed tail (|gelu'| < 6.1e-5, pre-activations below -4.55): here,
    both tower kinds, LoRA and full fine-tune walks — identical embeddings, gradients within 1e-5 of the largest element (bit-identical in practice)."""
    gd, gi = load("dna_tiny_golden.pt"), load("image_tiny_golden.pt")
    for hm, x in ((hip_image(gi, dev), gi["image_u8"].float() / 255.0), (hip_dna(gd, dev), gd["ids"])):
        assert hm.tower().stack.numerics["gelu_grad"] == "bf16"     # the default
        for full in (False, True):
            if full:
                for p_ in hm.parameters():
                    p_.requires_grad_(True)
            g = torch.Generator().manual_seed(6)
            res = {}
            for mode in ("bf16", "e4m7"):
                hm.tower().stack.set_numerics(gelu_grad=mode)
                y = hm(x.to(dev))
                if mode == "bf16":
                    cot = torch.randn(y.shape, generator=g)
                res[mode] = (y.detach().cpu(), grads_named(hm, (y * cot.to(dev)).sum()))
            assert torch.equal(res["e4m7"][0], res["bf16"][0])
            same = 0
            for n, gb in res["bf16"][1].items():
                ge = res["e4m7"][1][n]
                assert float((ge - gb).abs().max()) <= 1e-5 * max(float(gb.abs().max()), 1e-30), (n, full)
                same += int(torch.equal(ge, gb))
            print(f"[e4m7 gelu'] {type(hm).__name__} full={full}: {same} of {len(res['bf16'][1])} gradients bit-identical to the bf16 form")


@pytest.mark.parametrize("name", ["image", "dna"])
def test_full_finetune_residual_grad_streams_agree(dev, name):
    """since round 4 the full fine-tune walk (model_config.disable_lora) follows the residual_grad switch too, i.e. by
    default the gradient of the residual stream travels between block halves in bf16 where the reference's autograd keeps fp32.
    Both settings on the SAME model and cotangent: identical forward, every base-weight / bias / LayerNorm / embedding gradient of
    the bf16 stream within 2e-2 (cosine 0.9995) of the fp32 stream's — and the fp32 stream, the reference's semantics, still
    within the oracle gate."""
    from oracle import clibd_oracle as O
    from tests.test_oracle import build_dna, build_image

    if name == "image":
        gi = load("image_tiny_golden.pt")
        hm, om, x = hip_image(gi, dev), build_image(gi), gi["image_u8"].float() / 255.0
    else:
        gd = load("dna_tiny_golden.pt")
        hm, om, x = hip_dna(gd, dev), build_dna(gd), gd["ids"]
    for p in hm.parameters():
        p.requires_grad_(True)
    for p in om.parameters():
        p.requires_grad_(True)
    g = torch.Generator().manual_seed(11)
    res, cot = {}, None
    for mode in ("fp32", "bf16"):
        hm.tower().stack.set_numerics(residual_grad=mode)
        y = hm(x.to(dev))
        if cot is None:
            cot = torch.randn(y.shape, generator=g)
        res[mode] = (y.detach().cpu(), grads_named(hm, (y * cot.to(dev)).sum()))
    hm.tower().stack.set_numerics(residual_grad="bf16")
    assert torch.equal(res["fp32"][0], res["bf16"][0])
    assert len(res["bf16"][1]) > 30
    assert any(not torch.equal(res["bf16"][1][n], res["fp32"][1][n]) for n in res["fp32"][1]), "the switch did not reach the full fine-tune walk"
    assert_grads(res["bf16"][1], res["fp32"][1], rel_tol=2e-2, cos_tol=0.9995, what=f"{name} full fine-tune: bf16 vs fp32 residual-gradient stream",
                 zero_tol=2e-6, zero_rel=1e-4)
    with O.precision("bf16"):
        yo = om(x)
        ps = [(n, p) for n, p in om.named_parameters()]
        gs = torch.autograd.grad((yo * cot).sum(), [p for _, p in ps], allow_unused=True)
        go = {n: (torch.zeros_like(p) if g_ is None else g_) for (n, p), g_ in zip(ps, gs)}
    assert_grads(res["fp32"][1], go, rel_tol=3e-2, cos_tol=0.999, what=f"{name} full fine-tune, fp32 stream vs oracle", zero_tol=2e-6, zero_rel=1e-4)


def test_ln_fold_tower_matches_oracle(dev):
    """numerics ln_fold="on" (round 5): norm2 -> mlp.fc1 of every full ViT block as the algebraic fold (projection epilogue writes the bf16
    copy of the residual stream and its row sums, fc1's epilogue applies (mean, rstd)): a width-768 ViT of four blocks (three folded, the
    last on the class row as always) at batch 64 — the smallest batch whose projection GEMM the 256x256 kernel takes — against the oracle
    evaluating THE SAME fold (value of the fold, gradient of the unfolded pair) at the tower gates, and against the oracle's STANDARD bf16
    mode at the same gates: the fold moves a rounding point, not the error.  "off" and "on" must differ (the switch reaches the kernels)."""
    from oracle import clibd_oracle as O
    from clibd_amd.model import CLIBDImageEncoder, VisionTransformer

    torch.manual_seed(23)
    om = O.ImageEncoder(O.VisionTransformer(img_size=224, patch=16, dim=768, depth=4, heads=12, num_classes=0), 4, 768)
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n:
                p.normal_(0, 0.02)
            if "norm2" in n:   # a LayerNorm that is not the identity: gamma, beta enter the folded weight image and bias
                p.add_(0.2 * torch.randn_like(p))
    m = CLIBDImageEncoder(VisionTransformer(embed_dim=768, depth=4, num_heads=12, num_classes=0), r=4, num_classes=768)
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(24)
    img = torch.rand(64, 3, 224, 224, generator=g)
    cot = torch.randn(64, 768, generator=g)
    res = {}
    for mode in ("off", "on"):
        m.tower().stack.set_numerics(ln_fold=mode)
        y = m(img.to(dev))
        res[mode] = (y.detach().cpu(), grads_named(m, (y * cot.to(dev)).sum()))
    m.tower().stack.set_numerics(ln_fold="off")
    assert not torch.equal(res["on"][0], res["off"][0]), "the switch did not reach the kernels"
    assert rel(res["on"][0], res["off"][0]) < 4e-3

    with O.precision("bf16"), O.ln_fold(True):
        yo = om(img)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        yo_f, go_f = yo.detach(), dict(zip([n for n, _ in ps], torch.autograd.grad((yo * cot).sum(), [p for _, p in ps])))
    with O.precision("bf16"), torch.no_grad():    # (the standard mode's gradients are test_image_tower_parity's business)
        yo_s = om(img)
    e_ff, e_fs, e_ss = rel(res["on"][0], yo_f), rel(res["on"][0], yo_s), rel(res["off"][0], yo_s)
    print(f"[ln_fold] embeddings: fold vs oracle-fold {e_ff:.2e}, fold vs oracle-standard {e_fs:.2e}, standard vs oracle-standard {e_ss:.2e}")
    assert e_ff < 3e-3 and e_ss < 3e-3 and e_fs < 4e-3
    assert_grads(res["on"][1], go_f, what="ln_fold on vs oracle fold")
    assert_grads(res["on"][1], res["off"][1], rel_tol=3e-2, cos_tol=0.999, what="ln_fold on vs off")   # the fold moves the gradients by bf16 noise only
