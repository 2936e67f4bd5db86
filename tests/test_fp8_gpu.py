"""fp8-forward mode (BASELINE.json configs[4], "fp8 MFMA attention/GEMM path"): the fp8 GEMM is exact on integer operands
(bit-exact product, so any error of the mode is quantisation of the operands, not of the kernel), the producers' e4m3
outputs match a torch quantisation of their bf16-path outputs, and the full-size model in fp8-forward mode stays within the
stated distance of the bf16 path (the reference has no fp8 path: its bf16 autocast numbers are the anchor, via the bf16 path's
own parity tests)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FP8 = torch.float8_e4m3fn


def ints(shape, lo, hi, g):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (2000, 2304, 768), (1111, 768, 3072), (300, 256, 512), (1, 512, 2048)])
def test_gemm_fp8_exact_on_integers(dev, M, N, K):
    from clibd_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    a, w = ints((M, K), -3, 3, g), ints((N, K), -2, 2, g)
    cs = 2.0 ** torch.randint(-3, 2, (N,), generator=g).float()
    bias, res = ints((N,), -4, 4, g), ints((M, N), -8, 8, g)
    a8, w8 = a.to(FP8).to(dev), w.to(FP8).to(dev)
    ref = (a.double() @ w.double().T) * cs.double() + bias.double()
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    ops.gemm_fp8_nt(a8, w8, cs.to(dev), bias=bias.to(dev), residual=res.to(dev), out_f32=out)
    assert torch.equal(out.cpu().double(), ref + res.double())                      # residual form: exact in fp32
    u = torch.zeros((M, 8)); u[:, :4] = ints((M, 4), -1, 1, g); u[:, 4:] = ints((M, 4), -1, 1, g)
    v = ints((N, 8), -1, 1, g)
    outb = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8_nt(a8, w8, cs.to(dev), bias=bias.to(dev), rank_u=u.bfloat16().to(dev), rank_v=v.bfloat16().to(dev), out_bf16=outb)
    assert torch.equal(outb.cpu(), (ref + u.double() @ v.double().T).float().bfloat16())   # rank update added unscaled
    # gelu form: x exact, gelu(x) * 4 quantised to e4m3 like torch does; gelu' saved as bf16
    a2, cs2, b2 = a * 0.25, cs * 0.125, bias * 0.25
    x = ((a2.double() @ w.double().T) * cs2.double() + b2.double()).float().bfloat16().float()
    xg = x.clone().requires_grad_(True)
    gel = torch.nn.functional.gelu(xg)
    (dgel,) = torch.autograd.grad(gel.sum(), xg)
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    go = torch.empty((M, N), dtype=torch.uint8, device=dev).view(FP8)
    ops.gemm_fp8_nt(a2.to(FP8).to(dev), w8, cs2.to(dev), bias=b2.to(dev), gelu_out_fp8=go, gelu_out_scale=4.0, out_pre=pre)
    want = (gel.detach() * 4.0).clamp(-448, 448).to(FP8).float()
    got = go.cpu().float()
    # the kernel's erf polynomial differs from torch's in the last bits: allow one e4m3 step (2^-3 relative) on a few elements
    step = torch.maximum(want.abs() * 0.125, torch.tensor(2.0 ** -9))
    assert ((got - want).abs() <= step).all() and (got != want).float().mean() < 0.02
    assert (pre.cpu().float() - dgel).abs().max() < 1e-2


def test_gemm_fp8_rejects_unsupported(dev):
    from clibd_amd import ops

    a = torch.zeros((512, 384), dtype=torch.uint8, device=dev).view(FP8)      # K = 384: not a multiple of 256
    w = torch.zeros((256, 384), dtype=torch.uint8, device=dev).view(FP8)
    cs, bias = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    out = torch.empty((512, 256), dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="gemm_fp8"):
        ops.gemm_fp8_nt(a, w, cs, bias=bias, out_bf16=out)


def test_quantize_rows_and_producers(dev):
    from clibd_amd import ops

    g = torch.Generator().manual_seed(3)
    w = torch.randn((768, 3072), generator=g) * 0.05
    w[5] = 0.0                                                                    # an all-zero row keeps scale 1
    w8, cs = ops.quantize_rows_fp8(w.to(dev), 8.0)
    amax = w.abs().amax(dim=1)
    s = torch.where(amax > 0, 448.0 / amax, torch.ones_like(amax))
    assert torch.allclose(cs.cpu(), 1.0 / (s * 8.0), rtol=1e-6)
    want = (w * s[:, None]).clamp(-448, 448).to(FP8)
    assert torch.equal(w8.cpu().view(torch.uint8), want.view(torch.uint8))
    # LayerNorm: fp8 image of the same y
    M, H = 333, 768
    x = torch.randn((M, H), generator=g) * 3 + 1
    gam, bet = torch.randn(H, generator=g), torch.randn(H, generator=g)
    y16 = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
    y32 = torch.empty((M, H), dtype=torch.float32, device=dev)
    y8 = torch.empty((M, H), dtype=torch.uint8, device=dev).view(FP8)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), 1e-6, y_bf16=y16, y_f32=y32, y_fp8=y8, fp8_scale=8.0)
    want = (y32.cpu() * 8.0).clamp(-448, 448).to(FP8)
    assert torch.equal(y8.cpu().view(torch.uint8), want.view(torch.uint8))
    only8 = torch.empty((M, H), dtype=torch.uint8, device=dev).view(FP8)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), 1e-6, y_fp8=only8, fp8_scale=8.0)      # fp8 as the only output
    assert torch.equal(only8.view(torch.uint8), y8.view(torch.uint8))
    # attention: fp8 output = quantised bf16-path output up to the bf16 rounding the latter carries
    B, S, nh = 3, 197, 12
    qkv = (torch.randn((B * S, 3 * 64 * nh), generator=g) * 0.7).bfloat16().to(dev)
    o16 = torch.empty((B * S, 64 * nh), dtype=torch.bfloat16, device=dev)
    o8 = torch.empty((B * S, 64 * nh), dtype=torch.uint8, device=dev).view(FP8)
    ops.attention_fwd(qkv, B, S, nh, None, o16)
    ops.attention_fwd(qkv, B, S, nh, None, o8, out_fp8_scale=32.0)
    got, ref = o8.cpu().float() / 32.0, o16.cpu().float()
    assert ((got - ref).abs() <= ref.abs() * 0.0725 + 2.0 ** -9 / 32 + 1e-6).all()       # half an e4m3 step (2^-4) + bf16 rounding


def _full_size_pair(dev):
    from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

    torch.manual_seed(11)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    return model.to(dev).eval()


def test_full_size_fp8_forward_close_to_bf16_path(dev):
    """ViT-B/16 + BERT-base at batch 16: embeddings, loss and adapter / head gradients of the fp8-forward mode against the
    bf16 path of the same model.  Tolerances are the mode's own: e4m3 carries 3 mantissa bits (relative rounding error up
    to 2^-4 per operand element), every one of the 4 x 24 forward GEMMs adds ~2 % of independent noise to what it writes into
    the residual stream (tools/fp8_layer_drift.py: 3.8 % on the first qkv, 8.5 % on the stream after 12 ViT blocks).  Measured
    here: per-row cosine 0.995 (image) / 0.9996 (DNA), max |delta| 1.4e-2 on unit-norm rows, loss 1e-3.  Gates: cosine > 0.99
    per row, |delta| < 2e-2, loss within 2e-2.  Gradients: at random init the rows of a tower's output are nearly parallel
    (mutual cosine 0.997), so the part of an embedding that tells samples apart (norm 0.04) is smaller than the fp8 noise on
    the image side and the gradient direction moves accordingly (tools/fp8_errors.py: per-group cosines 0.77 - 0.93, 0.81
    overall); the gate (> 0.7 overall, > 0.85 on the DNA adapters) only catches a broken backward, e.g. a wrong gelu'."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import ClipLoss

    model = _full_size_pair(dev)
    B = 16
    batch = synthetic_batch(B, dev, seed=5, rank=0, with_text=False)
    labels = (torch.arange(B) % 11).to(dev)
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())

    def run():
        hi, hd, _, scale, _ = model(batch["image"], batch["dna"], None)
        loss = crit(hi, hd, None, labels, scale)
        ps = {n: p for n, p in model.named_parameters() if p.requires_grad}
        gs = torch.autograd.grad(loss, list(ps.values()), allow_unused=True)
        model.join_streams()
        torch.cuda.synchronize()
        gd = {n: (torch.zeros_like(p) if g is None else g).flatten().float().cpu() for (n, p), g in zip(ps.items(), gs)}
        return hi.detach().float().cpu(), hd.detach().float().cpu(), float(loss.detach()), gd

    i16, d16, l16, g16 = run()
    model.enable_fp8_forward(towers="all")     # the embedding-grade form: every tower, the ViT included
    i8, d8, l8, g8 = run()
    model.enable_fp8_forward(enabled=False)
    i16b, d16b, l16b, _ = run()
    assert torch.equal(i16, i16b) and torch.equal(d16, d16b) and l16 == l16b   # the mode switches off cleanly (round 6: the loss is a fixed-order sum)
    for a, b in ((i8, i16), (d8, d16)):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() < 2e-2
        assert ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).min().item() > 0.99
    assert abs(l8 - l16) < 2e-2
    def gcos(names):
        a, b = torch.cat([g8[n] for n in names]).double(), torch.cat([g16[n] for n in names]).double()
        return float(a @ b / (a.norm() * b.norm()))

    assert gcos(sorted(g16)) > 0.7
    assert gcos([n for n in sorted(g16) if n.startswith("dna") and ("w_a" in n or "w_b" in n)]) > 0.85


def test_fp8_calibration_sets_per_layer_scales(dev):
    """Calibrated per-layer scales are powers of two with headroom under 448 / amax and keep the embeddings within the mode's
    tolerance.  Then one LayerNorm of the ViT gets a gain of 96: its output (max ~ 400) saturates e4m3 under the static
    scale 8, while the calibrated scale of that layer drops and the result stays closer to the bf16 path."""
    from clibd_amd.data import synthetic_batch

    model = _full_size_pair(dev)
    B = 8
    batch = synthetic_batch(B, dev, seed=5, rank=0, with_text=False)
    cal = (batch["image"], batch["dna"], None)

    def emb():
        with torch.no_grad():
            hi, hd, _, _, _ = model(batch["image"], batch["dna"], None)
        model.join_streams()
        torch.cuda.synchronize()
        return hi.float().cpu(), hd.float().cpu()

    cosr = lambda a, b: ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).min().item()
    i16, d16 = emb()
    model.enable_fp8_forward(calibration_inputs=cal, towers="all")
    i_cal, d_cal = emb()
    st = model.image_encoder.tower().stack
    assert len(st.fp8) == 12 and all(set(d) == set(st.FP8_SITES) for d in st.fp8)
    for d in st.fp8:
        for v in d.values():
            assert v > 0 and abs(torch.log2(torch.tensor(v)).item() - round(torch.log2(torch.tensor(v)).item())) < 1e-6
    assert cosr(i_cal, i16) > 0.99 and cosr(d_cal, d16) > 0.99
    scale2 = st.fp8[2]["qkv_in"]
    model.enable_fp8_forward(enabled=False)
    with torch.no_grad():
        model.image_encoder.base_image_encoder.blocks[3].norm1.weight.mul_(96.0)
    i16, _ = emb()
    model.enable_fp8_forward(towers="all")
    i_static, _ = emb()
    model.enable_fp8_forward(calibration_inputs=cal)      # towers=None keeps the selection
    i_cal, _ = emb()
    assert st.fp8[3]["qkv_in"] * 16 <= scale2                                      # the hot layer got a much smaller scale
    assert (i_cal - i16).abs().max() < (i_static - i16).abs().max()               # saturation hurts the static scales


def test_fp8_forward_needs_frozen_base_and_supported_width(dev):
    from clibd_amd.engine import NotSupportedYet
    from clibd_amd.model import CLIBDImageEncoder, create_vit

    m = CLIBDImageEncoder(create_vit("vit_small_patch16_224"), r=4, num_classes=64).to(dev)
    with pytest.raises(NotSupportedYet):
        m.tower().stack.enable_fp8()                                               # hidden 384: K not a multiple of 256
    m = CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=64).to(dev)
    for p in m.base_image_encoder.blocks[0].mlp.fc1.parameters():
        p.requires_grad = True
    with pytest.raises(NotSupportedYet):
        m.tower().stack.enable_fp8()                                               # trainable base weights


def test_fp8_training_steps_reduce_loss(dev):
    from clibd_amd.data import synthetic_batch
    from clibd_amd.train import Trainer

    model = _full_size_pair(dev)
    model.enable_fp8_forward()
    B = 32
    batch = synthetic_batch(B, dev, seed=3, rank=0, with_text=False)
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True, fp8_recalibrate_every=3)   # scales re-measured before steps 0 and 3
    losses = [float(tr.step(batch["image"], batch["dna"], None, batch["labels"])) for _ in range(6)]
    assert all(l == l and l < 1e4 for l in losses) and losses[-1] < losses[0]
    st = model.dna_encoder.tower().stack
    assert st.fp8 is not None and len(st.fp8) == 12 and all(v > 0 for d in st.fp8 for v in d.values())
    assert model.image_encoder.tower().stack.fp8 is None     # default selection: the mean-pooled towers only (training-grade)


# ----------------------------------------------------------------------------------------------- round 3: the mode against ITS oracle
def _cosv(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def _oracle_pair(dev, seed=11):
    """Full-size HIP model + the oracle carrying the same weights (state-dict keys are the reference's on both sides)."""
    from oracle import clibd_oracle as O
    from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

    torch.manual_seed(seed)
    om = O.build_image_dna_model()
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
    model.load_state_dict(om.state_dict(), strict=True)
    return model.to(dev).eval(), om


def _hand_scales_to_oracle(model, om):
    """The oracle quantises with the scales the HIP towers hold (TransformerStack.fp8: [{site: scale}] per layer)."""
    from oracle import clibd_oracle as O

    blocks, layers = om.image_encoder.base_image_encoder.blocks, om.dna_encoder.base_dna_encoder.bert.encoder.layer
    f_img, f_dna = model.image_encoder.tower().stack.fp8, model.dna_encoder.tower().stack.fp8
    # a tower outside the selection evaluates in the oracle's bf16 arithmetic: no site carries a scale
    O.set_fp8_scales(blocks, f_img if f_img is not None else [{} for _ in blocks], last_block_qkv_only=f_img is not None)
    O.set_fp8_scales(layers, f_dna if f_dna is not None else [{} for _ in layers])


def _named_grads(module, loss):
    ps = {n: p for n, p in module.named_parameters() if p.requires_grad}
    gs = torch.autograd.grad(loss, list(ps.values()), allow_unused=True)
    return {n: (torch.zeros_like(p) if g is None else g).detach().float().cpu() for (n, p), g in zip(ps.items(), gs)}


@pytest.mark.parametrize("calibrated,towers", [(False, "all"), (True, "all"), (True, "pooled"), (True, "pooled_mlp"), (True, "pooled_ffn+dgrad8")])   # (round 6: the calibrated all-tower / all-site cases of round 4 are back)
def test_fp8_forward_matches_the_fp8_oracle(dev, calibrated, towers):
    """configs[4]'s mode against a CPU statement of the SAME arithmetic (oracle precision("fp8"): e4m3 operands with the towers'
    scales, fp32 accumulation, bf16 backward), ViT-B/16 + BERT-base at batch 16 — no longer HIP against HIP.
    What separates the two sides is summation order and the places where a last-bit difference of a producer lands on the
    other side of an e4m3 rounding boundary (one 2^-3 relative step on that operand element; a difference d upstream flips a
    fraction d / ulp of the elements by one ulp each, i.e. rms sqrt(d ulp) >> d: quantisation amplifies last-bit noise).
    Measured (MI355X, round 3): embeddings 5.0e-3 / 9.5e-4 (image / DNA, unit-norm rows) — against 1.4e-2 for fp8-vs-bf16 on
    the same weights, i.e. the quantisation error itself is reproduced to a third; loss 1.8e-3 with the static scales, 1.0e-4
    with calibrated ones; gradient cosine 0.88-0.89 over all trainable tensors, 0.92-0.94 on the DNA adapters: two correct
    implementations of this mode do not agree better than that on the gradient, which is why DESIGN.md §3.1b calls the mode
    embedding-grade, not gradient-faithful.  Gates: embeddings 1e-2, loss 3e-3, cosines 0.8 / 0.85.
    Round 4, towers="pooled" (the default selection: fp8 only where the head averages its tokens, the ViT in bf16): the image side is
    the bf16 path, the DNA side's quantisation noise is averaged by its head, and HIP and oracle agree on the gradient like two bf16
    implementations do: gate cosine >= 0.97 over all trainable tensors.
    Round 5, towers="pooled_mlp": "pooled" plus fp8 on the MLP pair (fc1, fc2) of every ViT block, the attention half of the block on
    bf16 operands — the oracle is told through the same per-layer dicts (a site without a scale is a bf16 site); gates as for "all".
    Round 5, later, "pooled_ffn+dgrad8": configs[4]'s fastest training-grade mode as a whole — fp8 forward on the MLP pair of the mean-pooled
    tower only, the 8-bit dgrad (numerics dgrad = "fp8") on BOTH towers — against the oracle evaluating both rules (precision("fp8") with
    the same site dicts + dgrad8()); gates as for "pooled"."""
    from oracle import clibd_oracle as O
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import ClipLoss

    model, om = _oracle_pair(dev)
    dg8 = towers.endswith("+dgrad8")
    # batch 16 for the static-scale case and for configs[4]'s whole mode; the three calibrated selections (restored in round 6) at batch 8: the CPU side of
    # this test is a full-size fp8-emulated forward + backward, ~1.5 s per sample on the GPU box's host share, and the suite has a time limit
    B = 16 if (not calibrated or dg8) else 8
    batch = synthetic_batch(B, torch.device("cpu"), seed=5, rank=0, with_text=False)
    labels = torch.arange(B) % (11 if B == 16 else 5)
    img, dna = batch["image"].to(dev), batch["dna"].to(dev)
    towers = towers.split("+")[0]
    if dg8:
        model.enable_fp8_dgrad(towers="all")
    model.enable_fp8_forward(calibration_inputs=(img, dna, None) if calibrated else None, towers=towers)
    _hand_scales_to_oracle(model, om)
    if calibrated:   # per-layer powers of two, not all equal to the static defaults
        sc = (model.image_encoder if towers == "all" else model.dna_encoder).tower().stack.fp8
        assert any(d != sc[0] for d in sc[1:]) or sc[0] != dict(model.image_encoder.tower().stack.FP8_SCALES)
    with O.precision("fp8"), O.dgrad8(dg8):
        oi, od, _, osc, _ = om(batch["image"], batch["dna"], None)
        lo = O.contrastive_loss([oi, od, None], labels, osc)
        ps = [(n, p) for n, p in om.named_parameters() if p.requires_grad]
        go = dict(zip([n for n, _ in ps], torch.autograd.grad(lo, [p for _, p in ps], allow_unused=True)))
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    hi, hd, _, scale, _ = model(img, dna, None)
    loss = crit(hi, hd, None, labels.to(dev), scale)
    got = _named_grads(model, loss)
    model.join_streams()
    torch.cuda.synchronize()
    errs = [(a.detach().float().cpu() - b_.detach()).abs().max().item() for a, b_ in ((hi, oi), (hd, od))]
    dl = abs(float(loss.detach()) - float(lo.detach()))
    names = sorted(got)
    assert names == sorted(n for n, _ in ps)
    allg = torch.cat([got[n].flatten() for n in names])
    allo = torch.cat([(torch.zeros_like(p) if go[n] is None else go[n]).flatten() for n, p in sorted(ps)])
    dna_ad = [n for n in names if n.startswith("dna") and (".w_a." in n or ".w_b." in n)]
    c_all = _cosv(allg, allo)
    c_dna = _cosv(torch.cat([got[n].flatten() for n in dna_ad]), torch.cat([go[n].flatten() for n in dna_ad]))
    print(f"[fp8 vs fp8 oracle, calibrated={calibrated}, towers={towers}{'+dgrad8' if dg8 else ''}] emb err image {errs[0]:.2e} dna {errs[1]:.2e} loss {dl:.2e} grad cos all {c_all:.4f} dna adapters {c_dna:.4f}")
    assert errs[0] < 1e-2 and errs[1] < 3e-3, errs
    assert dl < 3e-3, dl
    assert c_all > (0.97 if towers in ("pooled", "pooled_ffn") else 0.8) and c_dna > 0.85, (c_all, c_dna)
    if towers in ("pooled", "pooled_ffn"):
        assert errs[0] < 1e-3 and errs[1] < 3e-3, errs      # the image side IS the bf16 path


def test_fp8_gradients_on_spread_embeddings(dev):
    """Is the fp8-forward gradient the bf16 step's gradient once the embeddings are spread?  Adapters and heads are trained in bf16
    on a fixed batch of 32 pairs (8, then 40 steps), fp8 scales calibrated on the batch under test, cosine over ALL trainable tensors.

    towers="all" (every tower, round 3's mode) — it is not: measured on MI355X in round 3
        after  8 steps (loss 3.47 -> 3.20, mean mutual cosine of the image embeddings 0.90): cosine(fp8, bf16) 0.92 on the
                       training batch, 0.70 on a fresh batch; max |embedding difference| 0.08;
        after 40 steps (loss 0.03, mutual cosine 0.18): 0.45 / 0.61; max |embedding difference| 0.10 - 0.13.
    The fp8 forward moves the unit-norm IMAGE embedding of a trained tower by ~0.05-0.1 and the temperature (x14.3) turns that into
    O(1) logit noise.  Round 4 asked the oracle why (tools/fp8_policy_study.py, profiles/r04_exp_fp8_policy_study.log): it is the
    3-bit mantissa on a tower whose embedding is ONE token row — per-32-element E8M0 block scales (MXFP8), bf16 class-token rows,
    bf16 last blocks, activation-only and weight-only quantisation all leave the image embedding 3.5-5e-2 away — while the DNA
    tower, whose head averages 133 token rows, moves by 6-8e-3.

    towers="pooled" (round 4, the default): fp8 only on the towers whose head averages its tokens, the ViT in bf16.  The oracle
    predicts cosine 0.9999 on the training batch and 0.990 on a fresh one after 8 steps; the gate VERDICT r3 item 1 set — cosine
    >= 0.98 against the bf16 gradient on trained weights — is what this test now holds, at both stages, on both batches.

    Round 5: the same measurement for the forward selection "pooled_ffn" (fp8 on fc1 / fc2 of the mean-pooled towers only: their loss sat in the
    attention half), for the 8-bit dgrad (numerics dgrad = "fp8") on the mean-pooled towers / on all towers with a bf16 forward, and for the
    combinations.  Seven MI355X runs (fresh batches, worst ... best; training batches >= 0.998 everywhere): pooled 0.9799 ... 0.9899,
    pooled_ffn 0.9844 ... 0.9959, dgrad8(pooled) 0.9998 ... 0.9999, dgrad8(all) 0.9866 ... 0.9950, pooled_ffn + dgrad8(pooled) 0.9843 ... 0.9958,
    pooled_ffn + dgrad8(all) 0.9780 ... 0.9885, pooled + dgrad8(all) 0.9746 ... 0.9835 (which of these is configs[4]'s mode)."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import ClipLoss
    from clibd_amd.train import Trainer

    model = _full_size_pair(dev)
    B = 32
    batch = synthetic_batch(B, dev, seed=3, rank=0, with_text=False)
    fresh = synthetic_batch(B, dev, seed=4, rank=0, with_text=False)
    fresh_more = [synthetic_batch(B, dev, seed=sd, rank=0, with_text=False) for sd in (5, 6)]   # round 6: two more unseen batches for the modes called training-grade
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())

    def run(bt):
        hi, hd, _, scale, _ = model(bt["image"], bt["dna"], None)
        loss = crit(hi, hd, None, bt["labels"], scale)
        g = _named_grads(model, loss)
        model.join_streams()
        torch.cuda.synchronize()
        return hi.detach().float().cpu(), hd.detach().float().cpu(), g

    def compare(tag, out):
        for tw in (model.image_encoder.tower(), model.dna_encoder.tower()):
            tw.grad_sink = None     # plain autograd outputs for the comparison; the trainer's step() below re-installs its sink
        for name, bt in (("train", batch), ("fresh", fresh)):
            model.enable_fp8_forward(enabled=False)
            e16, d16, g16 = run(bt)
            names = sorted(g16)
            spread = float((e16 @ e16.T).fill_diagonal_(0).sum() / (B * (B - 1)))
            for towers in ("pooled", "pooled_ffn", "pooled_mlp", "all"):
                model.enable_fp8_forward(calibration_inputs=(bt["image"], bt["dna"], None), towers=towers)
                e8, d8, g8 = run(bt)
                out[(tag, name, towers)] = (_cosv(torch.cat([g8[n].flatten() for n in names]), torch.cat([g16[n].flatten() for n in names])),
                                            spread, float((e8 - e16).abs().max()), float((d8 - d16).abs().max()))
                if towers in ("pooled", "pooled_ffn"):   # round 5: + the 8-bit dgrad (numerics dgrad = "fp8") on the pooled towers / on BOTH towers
                    for sel, key in (("pooled", f"{towers}+dgrad8(pooled)"), ("all", f"{towers}+dgrad8(all)")):
                        model.enable_fp8_dgrad(towers=sel)
                        e8, d8, g8 = run(bt)
                        model.enable_fp8_dgrad(enabled=False)
                        out[(tag, name, key)] = (_cosv(torch.cat([g8[n].flatten() for n in names]), torch.cat([g16[n].flatten() for n in names])),
                                                 spread, float((e8 - e16).abs().max()), float((d8 - d16).abs().max()))
            model.enable_fp8_forward(enabled=False)
            for sel, key in (("pooled", "dgrad8(pooled)"), ("all", "dgrad8(all)")):     # the 8-bit dgrad alone, forward on bf16 operands everywhere
                model.enable_fp8_dgrad(towers=sel)
                e8, d8, g8 = run(bt)
                model.enable_fp8_dgrad(enabled=False)
                out[(tag, name, key)] = (_cosv(torch.cat([g8[n].flatten() for n in names]), torch.cat([g16[n].flatten() for n in names])),
                                         spread, float((e8 - e16).abs().max()), float((d8 - d16).abs().max()))
        model.enable_fp8_forward(enabled=False)
        # round 6: the modes this build calls TRAINING-GRADE, on two more batches the model has never seen
        for j, bt in enumerate(fresh_more):
            model.enable_fp8_forward(enabled=False)
            _, _, g16 = run(bt)
            names = sorted(g16)
            cat = lambda g_: torch.cat([g_[n].flatten() for n in names])
            for fwd, dg, key in (("pooled_ffn", None, "pooled_ffn"), ("pooled_ffn", "pooled", "pooled_ffn+dgrad8(pooled)"), (None, "pooled", "dgrad8(pooled)"), (None, "all", "dgrad8(all)")):
                if fwd:
                    model.enable_fp8_forward(calibration_inputs=(bt["image"], bt["dna"], None), towers=fwd)
                else:
                    model.enable_fp8_forward(enabled=False)
                if dg:
                    model.enable_fp8_dgrad(towers=dg)
                _, _, g8 = run(bt)
                model.enable_fp8_dgrad(enabled=False)
                extra[(tag, f"fresh{j + 2}", key)] = _cosv(cat(g8), cat(g16))
        model.enable_fp8_forward(enabled=False)
        sink = {id(p): p.grad for p in tr.optimizer.param_groups[0]["params"]}
        for tw in (model.image_encoder.tower(), model.dna_encoder.tower()):
            tw.grad_sink = sink

    out, losses, extra = {}, [], {}
    for stage, nsteps in (("8 steps", 8), ("40 steps", 32)):
        losses += [float(tr.step(batch["image"], batch["dna"], None, batch["labels"])) for _ in range(nsteps)]
        compare(stage, out)
    for k, (c, spread, de, dd) in out.items():
        print(f"[fp8 gradients on trained weights] after {k[0]}, {k[1]} batch, towers={k[2]}: cosine(fp8, bf16) {c:.4f}; mean mutual cosine of "
              f"image embeddings {spread:.3f}; max |embedding difference| image {de:.2e} dna {dd:.2e}")
    print(f"[fp8 gradients on trained weights] loss {losses[0]:.3f} -> {losses[7]:.3f} -> {losses[-1]:.3f}")
    assert losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])
    assert out[("40 steps", "train", "all")][1] < 0.9               # the embeddings did spread
    for k, c in extra.items():
        print(f"[fp8 gradients on trained weights] after {k[0]}, {k[1]} batch, towers={k[2]}: cosine(fp8, bf16) {c:.4f}")
    # Round 6: the 40 training steps are bit-reproducible now (no float atomics on the LoRA path: tests/test_determinism_gpu.py), so every figure
    # below is the SAME number in every run of this test, and the gates are the claims again:
    #   TRAINING-GRADE (cosine >= 0.98 against the bf16 gradient, both stages, training batch and every unseen batch):
    #     pooled_ffn, pooled_ffn + dgrad8(pooled) [configs[4]'s recommended mode], dgrad8(pooled), dgrad8(all);
    #   NOT training-grade, gated at what they measure (round 4's all-site "pooled": 0.977 on the unseen batch after 40 steps; with the ViT's
    #   8-bit dgrad on top of an fp8 forward the two errors add: 0.978 / 0.966) — README / DESIGN call them so;
    #   embedding-grade ("all", "pooled_mlp"): a broken-backward tripwire only.
    # All of it on random-init towers whose adapters and heads were trained 40 steps on 32 synthetic pairs: no pretrained weights exist here.
    TRAINING_GRADE = ("pooled_ffn", "pooled_ffn+dgrad8(pooled)", "dgrad8(pooled)", "dgrad8(all)")
    for k, (c, _, de, dd) in out.items():
        if k[2] == "dgrad8(pooled)":
            assert c >= 0.999 and de == 0.0 and dd == 0.0, (k, c, de, dd)  # the 8-bit dgrad of the mean-pooled towers: free (0.9998 - 1.0000)
        elif k[2] == "dgrad8(all)":
            assert c >= 0.98 and de == 0.0 and dd == 0.0, (k, c, de, dd)   # + the ViT's: 0.9892 - 0.9990
        elif k[2] in ("pooled_ffn", "pooled_ffn+dgrad8(pooled)"):
            assert c >= 0.98 and de == 0.0 and dd < 3e-2, (k, c, de, dd)   # fp8 forward on the pooled towers' MLP pair [+ their 8-bit dgrad]: 0.9896 - 0.9999 over two trees' trajectories (profiles/r06_fp8_fidelity_final_tree.log)
        elif k[2] == "pooled_ffn+dgrad8(all)":
            assert c >= 0.97 and de == 0.0 and dd < 3e-2, (k, c, de, dd)   # NOT training-grade: 0.9779 on the unseen batch after 40 steps
        elif k[2] in ("pooled", "pooled+dgrad8(pooled)"):
            assert c >= 0.97 and de == 0.0 and dd < 6e-2, (k, c, de, dd)   # NOT training-grade (round 4's selection): 0.9770 / 0.9837 on the unseen batch after 40 steps (two trees)
        elif k[2] == "pooled+dgrad8(all)":
            assert c >= 0.96 and de == 0.0 and dd < 6e-2, (k, c, de, dd)   # NOT training-grade: 0.9655
        else:
            # embedding-grade: the floor round 3's measurement set.  "pooled_mlp" (round 5: + the ViT's MLP pair) sits between the two —
            # the oracle study (profiles/r05_exp_fp8_vit_sites.log) has it at 0.985 on the training batch and 0.82 on a fresh one after
            # 8 steps: it does NOT pass the 0.98 gate on both batches, so it is not the default and not called training-grade
            assert c > 0.15 and de < 0.3, (k, c, de)
    for k, c in extra.items():
        assert k[2] in TRAINING_GRADE and c >= 0.98, (k, c)
    for stage in ("8 steps", "40 steps"):
        for name in ("train", "fresh"):
            assert out[(stage, name, "pooled_mlp")][0] >= out[(stage, name, "all")][0] - 0.05, (stage, name, out[(stage, name, "pooled_mlp")][0], out[(stage, name, "all")][0])


def test_fp8_gradient_fidelity_against_the_batch_size(dev):
    """Round 6: what the gradient cosine of each configs[4] mode is AT configs[4]'s batch.  The 8-bit dgrad rounds every gradient row on its own
    (per-row power-of-two scales, round to nearest): its error is independent from row to row and averages out over the batch the parameter
    gradient sums over; the fp8 FORWARD moves a sample's embedding, which the loss sees coherently — that error does not average.  So the 32-pair
    protocol of the test above is the pessimistic end for every mode with an 8-bit dgrad: on an unseen batch of 512 pairs (same trained adapters)
    dgrad8(all) reads 0.9981 (0.9889 at 32 pairs, 0.9925 at 128) and pooled_ffn + dgrad8(all) — 0.978-0.983 at 32 pairs, "not training-grade" there —
    0.9954 (profiles/r06_fp8_fidelity_vs_batch.log).  bench.py's
    `configs4` record measures the same three modes in-run at per-GPU batch 1024 (+7.7 % at 0.9885, +9.2 % at 0.9988, +13.5 % at 0.9874)."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import ClipLoss
    from clibd_amd.train import Trainer

    model = _full_size_pair(dev)
    batch = synthetic_batch(32, dev, seed=3, rank=0, with_text=False)
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
    crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
    losses = [float(tr.step(batch["image"], batch["dna"], None, batch["labels"])) for _ in range(40)]
    assert losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])
    for tw in (model.image_encoder.tower(), model.dna_encoder.tower()):
        tw.grad_sink = None

    def grads(bt):
        torch.manual_seed(77)            # the same dropout masks in every mode
        hi, hd, _, scale, _ = model(bt["image"], bt["dna"], None)
        g = _named_grads(model, crit(hi, hd, None, bt["labels"], scale))
        model.join_streams()
        torch.cuda.synchronize()
        names = sorted(g)
        return torch.cat([g[n].flatten().double() for n in names])

    MODES = (("pooled_ffn", "pooled", "pooled_ffn+dgrad8(pooled)"), (None, "all", "dgrad8(all)"), ("pooled_ffn", "all", "pooled_ffn+dgrad8(all)"))
    res = {}
    for B in (32, 128, 512):
        bt = synthetic_batch(B, dev, seed=11, rank=0, with_text=False)      # never seen by the 40 steps
        model.enable_fp8_forward(enabled=False)
        model.enable_fp8_dgrad(enabled=False)
        g16 = grads(bt)
        for fwd, dg, key in MODES:
            if fwd:
                model.enable_fp8_forward(calibration_inputs=(bt["image"], bt["dna"], None), towers=fwd)
            else:
                model.enable_fp8_forward(enabled=False)
            model.enable_fp8_dgrad(towers=dg)
            res[(B, key)] = _cosv(grads(bt), g16)
            model.enable_fp8_dgrad(enabled=False)
            print(f"[fp8 fidelity vs batch] unseen batch of {B:4d} pairs, {key}: cosine(fp8, bf16) {res[(B, key)]:.4f}")
    model.enable_fp8_forward(enabled=False)
    assert res[(512, "dgrad8(all)")] >= 0.995 and res[(512, "dgrad8(all)")] > res[(32, "dgrad8(all)")], res     # the dgrad's error averages out over the batch
    assert res[(512, "pooled_ffn+dgrad8(all)")] >= 0.98, res                                                     # training-grade at a realistic batch
    assert res[(512, "pooled_ffn+dgrad8(pooled)")] >= 0.98, res
    assert res[(512, "dgrad8(all)")] > res[(512, "pooled_ffn+dgrad8(pooled)")], res                              # at that batch the forward's error is the larger one


def test_fp8_pooled_selection_covers_the_text_tower(dev):
    """Tri-modal model (BASELINE configs[3] + configs[4]): the default fp8 selection puts BarcodeBERT AND BERT-small (H = 512, key
    mask, mean over 20 positions: the other token-averaging head) on fp8 operands and leaves the ViT alone.  Embeddings stay within the
    mode's distance of the bf16 path on the pooled towers, the image rows are bit-identical, a training step runs and learns."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import (CLIBDDNAEncoder, CLIBDImageEncoder, CLIBDLanguageEncoder, SimpleCLIP, create_vit, load_pre_trained_bert,
                                 load_pre_trained_bioscan_bert)
    from clibd_amd.train import Trainer

    torch.manual_seed(17)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768),
                       CLIBDLanguageEncoder(load_pre_trained_bert()[1], r=4, num_classes=768))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    model = model.to(dev).eval()
    B = 32
    batch = synthetic_batch(B, dev, seed=3, rank=0, with_text=True)

    def emb():
        with torch.no_grad():
            out = model(batch["image"], batch["dna"], batch["text"])
        model.join_streams()
        torch.cuda.synchronize()
        return [o.float().cpu() for o in out[:3]]

    i16, d16, t16 = emb()
    model.enable_fp8_forward(calibration_inputs=(batch["image"], batch["dna"], batch["text"]))     # towers="pooled"
    assert model.image_encoder.tower().stack.fp8 is None
    assert model.dna_encoder.tower().stack.fp8 is not None and len(model.language_encoder.tower().stack.fp8) == 4
    i8, d8, t8 = emb()
    cosr = lambda a, b: ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).min().item()
    assert torch.equal(i8, i16)
    assert cosr(d8, d16) > 0.999 and cosr(t8, t16) > 0.99, (cosr(d8, d16), cosr(t8, t16))
    assert (d8 - d16).abs().max() < 2e-2 and (t8 - t16).abs().max() < 5e-2
    tr = Trainer(model.train(), lr=1e-3, world_size=1, rank=0, all_gather=True, fp8_recalibrate_every=2)
    losses = [float(tr.step(batch["image"], batch["dna"], batch["text"], batch["labels"])) for _ in range(4)]
    assert all(l == l and l < 1e4 for l in losses) and losses[-1] < losses[0], losses
    assert model.image_encoder.tower().stack.fp8 is None and model.language_encoder.tower().stack.fp8 is not None   # the selection survives re-calibration
