"""8-bit dgrad (numerics switch dgrad = "fp8", BASELINE.json configs[4]): the backward's activation-gradient GEMMs on e4m3 operands.
Kernel level: the dgrad GEMM forms are exact on integer operands (so the mode's error is the quantisation of the operands, never the
kernel), the LayerNorm backward's e4m3 rows are bit-for-bit a torch quantisation of its own fp32 output under the stated row-scale rule.
Model level (tests further down): tower gradients against the oracle's restatement of the same rule and against the bf16 dgrad."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FP8 = torch.float8_e4m3fn


def ints(shape, lo, hi, g):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def row_scale(v):
    """s_m = 2^(7 - floor(log2 max|row|)), 1 for an all-zero row (clibd_layernorm_bwd_fp8)."""
    amax = v.abs().amax(dim=1, keepdim=True)
    e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax))))
    return torch.where(amax > 0, torch.exp2(7.0 - e), torch.ones_like(amax))


@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (2000, 3072, 768), (1108, 768, 3072), (300, 256, 512), (4, 512, 2048)])
def test_gemm_fp8_dgrad_exact_on_integers(dev, M, N, K):
    from clibd_amd import ops

    g = torch.Generator().manual_seed(M + 2 * N + 3 * K)
    a, w = ints((M, K), -3, 3, g), ints((N, K), -2, 2, g)
    cs = 2.0 ** torch.randint(-3, 2, (N,), generator=g).float()
    rd = 2.0 ** torch.randint(-4, 3, (M,), generator=g).float()
    aux = ints((M, N), -4, 4, g)
    a8, w8 = a.to(FP8).to(dev), w.to(FP8).to(dev)
    acc = a.double() @ w.double().T
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8_dgrad_nt(a8, w8, cs.to(dev), a_row_dequant=rd.to(dev), out_bf16=out)
    assert torch.equal(out.cpu(), (acc * cs.double() * rd.double()[:, None]).float().bfloat16())
    ops.gemm_fp8_dgrad_nt(a8, w8, cs.to(dev), a_row_dequant=rd.to(dev), aux=aux.bfloat16().to(dev), act=ops.ACT_ADD_AUX, out_bf16=out)
    assert torch.equal(out.cpu(), (acc * cs.double() * rd.double()[:, None] + aux.double()).float().bfloat16())
    # the form that writes the next dgrad's operand: e4m3(acc * col_scale * aux * c), the row scale of `a` passing through
    o8 = torch.empty((M, N), dtype=torch.uint8, device=dev).view(FP8)
    ops.gemm_fp8_dgrad_nt(a8, w8, cs.to(dev), aux=aux.bfloat16().to(dev), act=ops.ACT_MUL_AUX, out_fp8=o8, out_fp8_scale=2.0 ** -6)
    want = (acc * cs.double() * aux.double() * 2.0 ** -6).float().clamp(-448, 448).to(FP8)
    assert torch.equal(o8.cpu().view(torch.uint8), want.view(torch.uint8))
    # round 6 (full fine-tune): the same form writing ALSO the de-scaled value as bf16 — the operand of the bf16 weight gradient
    o8b, dual = torch.empty_like(o8), torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8_dgrad_nt(a8, w8, cs.to(dev), aux=aux.bfloat16().to(dev), act=ops.ACT_MUL_AUX, out_fp8=o8b, out_fp8_scale=2.0 ** -6,
                          a_row_dequant=rd.to(dev), out_bf16_dual=dual)
    assert torch.equal(o8b.cpu().view(torch.uint8), want.view(torch.uint8))
    assert torch.equal(dual.cpu(), (acc * cs.double() * aux.double() * rd.double()[:, None]).float().bfloat16())


def test_gemm_fp8_dgrad_rejects_unsupported(dev):
    from clibd_amd import ops

    z8 = lambda r, c: torch.zeros((r, c), dtype=torch.uint8, device=dev).view(FP8)
    cs, rd = torch.ones(256, device=dev), torch.ones(510, device=dev)
    out = torch.empty((510, 256), dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="gemm_fp8_dgrad"):      # M % 4
        ops.gemm_fp8_dgrad_nt(z8(510, 512), z8(256, 512), cs, a_row_dequant=rd, out_bf16=out)
    with pytest.raises(RuntimeError, match="gemm_fp8_dgrad"):      # K = 384
        ops.gemm_fp8_dgrad_nt(z8(512, 384), z8(256, 384), cs, a_row_dequant=torch.ones(512, device=dev), out_bf16=torch.empty((512, 256), dtype=torch.bfloat16, device=dev))
    with pytest.raises(ValueError):                                  # bf16-output forms need the row scales
        ops.gemm_fp8_dgrad_nt(z8(512, 512), z8(256, 512), cs, out_bf16=torch.empty((512, 256), dtype=torch.bfloat16, device=dev))


@pytest.mark.parametrize("M,H,stream", [(1001, 768, "bf16"), (517, 768, "fp32"), (260, 512, "bf16"), (64, 1024, "bf16"), (131200, 768, "bf16")])
def test_layernorm_bwd_fp8_rows_are_the_quantised_fp32_output(dev, M, H, stream):
    """Without dropout the e4m3 copy quantises exactly the values dx_f32 holds: bytes and row scales must equal a torch quantisation of
    that output (M = 131 200 takes the two-rows-per-wave form).  A zero row takes scale 1."""
    from clibd_amd import ops

    g = torch.Generator().manual_seed(M + H)
    x = torch.randn(M, H, generator=g) * 1.5 + 0.3
    dy = (torch.randn(M, H, generator=g) * 10.0 ** torch.randint(-6, 1, (M, 1), generator=g).float()).bfloat16()
    dy[3] = 0
    gamma = 1.0 + 0.2 * torch.randn(H, generator=g)
    stats = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
    dres32 = torch.randn(M, H, generator=g) * dy.float().abs().amax(dim=1, keepdim=True)
    dres32[3] = 0
    kw = dict(dres_bf16=dres32.bfloat16().to(dev)) if stream == "bf16" else dict(dres=dres32.to(dev))
    dx32 = torch.empty((M, H), device=dev)
    d8 = torch.empty((M, H), dtype=torch.uint8, device=dev).view(FP8)
    rd = torch.empty((M,), device=dev)
    res16 = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), stats.to(dev), gamma.to(dev), dx_f32=dx32, dx_res_bf16=res16 if stream == "bf16" else None,
                      dx_fp8=d8, row_dequant=rd, **kw)
    v = dx32.cpu()
    s = row_scale(v)
    assert torch.equal(rd.cpu(), (1.0 / s).flatten())
    assert float(rd[3]) == 1.0 and int(d8[3].view(torch.uint8).max()) == 0
    assert torch.equal(d8.cpu().view(torch.uint8), (v * s).clamp(-448, 448).to(FP8).view(torch.uint8))
    scaled_max = (v * s).abs().amax(dim=1)
    live = v.abs().amax(dim=1) > 0
    assert float(scaled_max[live].min()) >= 128.0 and float(scaled_max.max()) < 256.0
    if stream == "bf16":
        assert torch.equal(res16.cpu(), v.bfloat16())
    # the plain kernel computes the same dx
    ref = torch.empty((M, H), device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), stats.to(dev), gamma.to(dev), dx_f32=ref, dx_res_bf16=torch.empty_like(res16) if stream == "bf16" else None, **kw)
    assert torch.equal(ref, dx32)


def test_layernorm_bwd_fp8_rows_carry_the_dropout_mask(dev):
    """With the dense branch's dropout active the e4m3 copy holds the MASKED gradient (the values of dx_bf16), the residual copy does not."""
    from clibd_amd import ops

    M, H = 777, 768
    g = torch.Generator().manual_seed(5)
    x, dy = torch.randn(M, H, generator=g), torch.randn(M, H, generator=g).bfloat16()
    gamma = 1.0 + 0.1 * torch.randn(H, generator=g)
    stats = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
    drop = ops.Drop(0.1, 1234)
    masked = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
    res = torch.empty_like(masked)
    d8 = torch.empty((M, H), dtype=torch.uint8, device=dev).view(FP8)
    rd = torch.empty((M,), device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), stats.to(dev), gamma.to(dev), dx_res_bf16=res, dx_bf16=masked, drop=drop, dx_fp8=d8, row_dequant=rd)
    deq = d8.float() * rd[:, None]
    zero = masked == 0
    assert 0.08 < float(zero.float().mean()) < 0.12
    assert torch.equal(deq == 0, zero) or float(((deq == 0) != zero).float().mean()) < 1e-4   # (values below the row's e4m3 range also read zero)
    assert float((deq - masked.float()).abs().max() / masked.float().abs().max()) < 2.0 ** -4
    assert float((res != 0).float().mean()) > 0.99
    # without the bf16 copy: same bytes
    d8b, rdb = torch.empty_like(d8), torch.empty_like(rd)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), stats.to(dev), gamma.to(dev), dx_res_bf16=torch.empty_like(res), drop=drop, dx_fp8=d8b, row_dequant=rdb)
    assert torch.equal(d8b.view(torch.uint8), d8.view(torch.uint8)) and torch.equal(rdb, rd)


def test_layernorm_bwd_fp8_two_row_form_equals_the_one_row_form(dev):
    """M >= 131 072 with a bf16 incoming stream and no fp32 output takes the two-rows-per-wave kernel (round 6: the residual gradient waits
    for the reductions as packed bf16 pairs — 118 registers, none spilled); asking for dx_f32 as well selects the one-row kernel.  Same
    arithmetic: e4m3 bytes, row scales and the bf16 copies must be identical, with and without a dropout mask."""
    from clibd_amd import ops

    M, H = 131072 + 320, 768
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(M, H, generator=g) * 1.5 + 0.3).to(dev)
    dy = (torch.randn(M, H, generator=g) * 10.0 ** torch.randint(-5, 1, (M, 1), generator=g).float()).bfloat16().to(dev)
    gamma = (1.0 + 0.2 * torch.randn(H, generator=g)).to(dev)
    stats = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
    dres = (torch.randn(M, H, generator=g).to(dev) * dy.float().abs().amax(dim=1, keepdim=True)).bfloat16()
    for drop in (None, ops.Drop(0.1, 12345)):
        outs = []
        for one_row in (False, True):
            d8 = torch.empty((M, H), dtype=torch.uint8, device=dev).view(FP8)
            rd = torch.empty((M,), device=dev)
            res, msk = torch.empty((M, H), dtype=torch.bfloat16, device=dev), torch.empty((M, H), dtype=torch.bfloat16, device=dev)
            extra = dict(dx_f32=torch.empty((M, H), device=dev)) if one_row else {}
            ops.layernorm_bwd(dy, x, stats, gamma, dres_bf16=dres, dx_res_bf16=res, dx_bf16=msk, drop=drop, dx_fp8=d8, row_dequant=rd, **extra)
            outs.append((d8.view(torch.uint8).clone(), rd.clone(), res.clone(), msk.clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert float(outs[0][1].min()) > 0


def test_layernorm_bwd_fp8_with_parameter_gradients(dev):
    """clibd_layernorm_bwd_fp8_pg (8-bit dgrad under full fine-tune): the e4m3 rows / scales / bf16 copies of the plain fp8 call, bit for
    bit, plus the parameter gradients of the bf16 path's call (float-atomic order only)."""
    from clibd_amd import ops

    M, H = 4256, 768
    g = torch.Generator().manual_seed(78)
    x = (torch.randn(M, H, generator=g) * 1.5 + 0.3).to(dev)
    dy = (torch.randn(M, H, generator=g) * 1e-3).bfloat16().to(dev)
    gamma = (1.0 + 0.2 * torch.randn(H, generator=g)).to(dev)
    stats = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
    drop = ops.Drop(0.1, 999)
    new16 = lambda: torch.empty((M, H), dtype=torch.bfloat16, device=dev)
    d8a, rda, resa, mska = torch.empty((M, H), dtype=torch.uint8, device=dev).view(FP8), torch.empty((M,), device=dev), new16(), new16()
    ops.layernorm_bwd(dy, x, stats, gamma, dx_res_bf16=resa, dx_bf16=mska, drop=drop, dx_fp8=d8a, row_dequant=rda)
    d8b, rdb, resb, mskb = torch.empty_like(d8a), torch.empty_like(rda), new16(), new16()
    dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ops.layernorm_bwd(dy, x, stats, gamma, dx_res_bf16=resb, dx_bf16=mskb, drop=drop, dx_fp8=d8b, row_dequant=rdb, dgamma=dg, dbeta=db)
    assert torch.equal(d8a.view(torch.uint8), d8b.view(torch.uint8)) and torch.equal(rda, rdb) and torch.equal(resa, resb) and torch.equal(mska, mskb)
    dg2, db2 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ops.layernorm_bwd(dy, x, stats, gamma, dx_res_bf16=new16(), dx_bf16=new16(), drop=drop, dgamma=dg2, dbeta=db2)
    xhat = (x - stats[:, :1]) * stats[:, 1:]
    assert torch.allclose(dg, (dy.float() * xhat).sum(0), rtol=1e-4, atol=1e-6) and torch.allclose(db, dy.float().sum(0), rtol=1e-4, atol=1e-6)
    assert torch.allclose(dg, dg2, rtol=1e-4, atol=1e-6) and torch.allclose(db, db2, rtol=1e-4, atol=1e-6)   # (float-atomic order across 1024 blocks)


def test_quantize_rows_fp8_bf16_and_its_l1_bound(dev):
    from clibd_amd import ops

    g = torch.Generator().manual_seed(9)
    w = (torch.randn(3072, 768, generator=g) * 0.02).bfloat16()
    w[5] = 0
    l1 = torch.zeros((1,), device=dev)
    w8, cs = ops.quantize_rows_fp8_bf16(w.to(dev), 1.0, l1)
    s = row_scale(w.float())     # the same power-of-two rule as the gradient rows
    want = (w.float() * s).clamp(-448, 448).to(FP8)
    assert torch.equal(w8.cpu().view(torch.uint8), want.view(torch.uint8))
    assert torch.equal(cs.cpu(), (1.0 / s).flatten())
    deq_l1 = (want.float().abs().sum(dim=1) / s.flatten()).max()
    assert abs(float(l1) - float(deq_l1)) <= 1e-5 * float(deq_l1)


# ------------------------------------------------------------------------------------------------ towers
def _grads(named_params, loss):
    ps = [(n, p) for n, p in named_params if p.requires_grad]
    gs = torch.autograd.grad(loss, [p for _, p in ps], allow_unused=True)
    return {n: (torch.zeros_like(p) if g is None else g).detach().float().cpu() for (n, p), g in zip(ps, gs)}


def _flat(g, names):
    return torch.cat([g[n].flatten().double() for n in names])


def _cos(a, b):
    return float(a @ b / (a.norm() * b.norm()))


def _compare(got8, got16, ora8, what, gate_oracle, gate_bf16):
    names = sorted(ora8)
    assert sorted(got8) == names == sorted(got16)
    f8, f16, o8 = _flat(got8, names), _flat(got16, names), _flat(ora8, names)
    c_or, c_16 = _cos(f8, o8), _cos(f8, f16)
    r_or = float((f8 - o8).norm() / o8.norm())
    per = min(_cos(got8[n].flatten().double(), ora8[n].flatten().double()) for n in names if float(ora8[n].abs().max()) > 1e-9 * float(o8.abs().max()))
    print(f"[dgrad8 {what}] cosine vs oracle(dgrad8) {c_or:.5f} (rel {r_or:.2e}, worst parameter {per:.4f}), vs the bf16 dgrad {c_16:.5f}")
    assert not torch.equal(f8, f16), "the switch did not reach the kernels"
    assert c_or > gate_oracle and per > gate_oracle - 0.01
    assert c_16 > gate_bf16


def test_dgrad8_dna_tower_matches_oracle(dev):
    """BarcodeBERT at its bench width (H = 768, FF = 3072, S = 133; 3 layers, batch 16 -> M = 2128) in TRAIN mode (dropout masks on both dense
    branches: the e4m3 rows carry them): numerics dgrad="fp8" against the oracle's restatement of the same quantisation rule at the oracle's
    own masks, and against the HIP tower's bf16 dgrad.  The two implementations quantise gradients that differ in their last bf16 bits, so
    rounding ties fall differently: the gate is a cosine (>= 0.999 overall), not the 2e-2 element gate of the bf16 path."""
    from oracle import clibd_oracle as O
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder

    torch.manual_seed(31)
    om = O.DNAEncoder(O.BertForMaskedLM(vocab=1027, hidden=768, layers=3, heads=12, ff=3072), 4, 768)
    with torch.no_grad():
        for n, p in om.named_parameters():
            if ".w_b." in n:
                p.normal_(0, 0.02)
    m = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072)), r=4, num_classes=768)
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.to(dev).train()
    B = 16
    ids = synthetic_batch(B, torch.device("cpu"), seed=11, rank=0, with_text=False)["dna"]
    cot = torch.randn(B, 768, generator=torch.Generator().manual_seed(6))
    res = {}
    for mode in ("bf16", "fp8"):
        m.tower().stack.set_numerics(dgrad=mode)
        torch.manual_seed(99)
        base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        torch.manual_seed(99)       # the tower draws the same base seed from the CPU generator
        y = m(ids.to(dev))
        res[mode] = (y.detach().cpu(), _grads(m.named_parameters(), (y * cot.to(dev)).sum()))
    m.tower().stack.set_numerics(dgrad="bf16")
    assert torch.equal(res["bf16"][0], res["fp8"][0])        # the forward is untouched
    with O.precision("bf16"), O.dropout(0.1, 0.1, base), O.dgrad8(True):
        yo = om(ids)
        ora = _grads(om.named_parameters(), (yo * cot).sum())
    keep = lambda g: {n: v for n, v in g.items() if n in ora}
    _compare(keep(res["fp8"][1]), keep(res["bf16"][1]), ora, "BarcodeBERT width, train mode", 0.999, 0.995)


def test_dgrad8_image_tower_matches_oracle(dev):
    """Width-768 ViT of three blocks (two full ones on the 8-bit dgrad, the class-row-only last block on bf16 as in the oracle), batch 16."""
    from oracle import clibd_oracle as O
    from clibd_amd.model import CLIBDImageEncoder, VisionTransformer

    torch.manual_seed(33)
    om = O.ImageEncoder(O.VisionTransformer(img_size=224, patch=16, dim=768, depth=3, heads=12, num_classes=0), 4, 768)
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n:
                p.normal_(0, 0.02)
    m = CLIBDImageEncoder(VisionTransformer(embed_dim=768, depth=3, num_heads=12, num_classes=0), r=4, num_classes=768)
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(34)
    img, cot = torch.rand(16, 3, 224, 224, generator=g), torch.randn(16, 768, generator=g)
    res = {}
    for mode in ("bf16", "fp8"):
        m.tower().stack.set_numerics(dgrad=mode)
        y = m(img.to(dev))
        res[mode] = (y.detach().cpu(), _grads(m.named_parameters(), (y * cot.to(dev)).sum()))
    m.tower().stack.set_numerics(dgrad="bf16")
    assert torch.equal(res["bf16"][0], res["fp8"][0])
    with O.precision("bf16"), O.dgrad8(True):
        yo = om(img)
        ora = _grads(om.named_parameters(), (yo * cot).sum())
    keep = lambda g_: {n: v for n, v in g_.items() if n in ora}
    _compare(keep(res["fp8"][1]), keep(res["bf16"][1]), ora, "ViT width 768", 0.999, 0.995)


@pytest.mark.parametrize("train_mode", [False, True], ids=["eval", "train"])
def test_dgrad8_full_finetune_dna_tower_matches_oracle(dev, train_mode):
    """Round 6: the 8-bit dgrad with TRAINABLE base weights — `disable_lora: true`, the reference's final BIOSCAN-1M / 5M recipe
    (config/model_config/for_bioscan_5m/final_experiments/image_dna_seed_42.yaml:20).  BarcodeBERT width (H = 768, FF = 3072, S = 133), 2 layers,
    batch 16, every parameter trainable: the dgrad GEMMs take e4m3 rows, every weight / bias / LayerNorm gradient stays bf16.  Against the
    oracle's restatement of the same rule (same masks in train mode) and against the HIP tower's bf16 dgrad, over ALL parameters."""
    from oracle import clibd_oracle as O
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder

    torch.manual_seed(33)
    om = O.DNAEncoder(O.BertForMaskedLM(vocab=1027, hidden=768, layers=2, heads=12, ff=3072), 4, 768, lora_layer=[])
    m = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072)), r=4, num_classes=768, lora_layer=[])
    m.load_state_dict(om.state_dict(), strict=True)
    for mod in (om, m):
        for p_ in mod.parameters():
            p_.requires_grad_(True)
    m = m.to(dev).train(train_mode)
    assert m.tower().stack.full_mode()
    B = 16
    ids = synthetic_batch(B, torch.device("cpu"), seed=12, rank=0, with_text=False)["dna"]
    cot = torch.randn(B, 768, generator=torch.Generator().manual_seed(7))
    res, base = {}, 0
    for mode in ("bf16", "fp8"):
        m.tower().stack.set_numerics(dgrad=mode)
        torch.manual_seed(98)
        base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        torch.manual_seed(98)
        y = m(ids.to(dev))
        res[mode] = (y.detach().cpu(), _grads(m.named_parameters(), (y * cot.to(dev)).sum()))
    m.tower().stack.set_numerics(dgrad="bf16")
    assert torch.equal(res["bf16"][0], res["fp8"][0])
    import contextlib
    with O.precision("bf16"), (O.dropout(0.1, 0.1, base) if train_mode else contextlib.nullcontext()), O.dgrad8(True):
        yo = om(ids)
        ora = _grads(om.named_parameters(), (yo * cot).sum())
    live = {n for n, v in ora.items() if float(v.abs().max()) > 0}
    keep = lambda g: {n: v for n, v in g.items() if n in live and n in ora}
    assert any("intermediate.dense.weight" in n for n in live) and any("attention.output.LayerNorm.weight" in n for n in live)
    g8, g16, go = keep(res["fp8"][1]), keep(res["bf16"][1]), keep(ora)
    names = sorted(go)
    f8, f16, o8 = _flat(g8, names), _flat(g16, names), _flat(go, names)
    big = max(float(go[n].double().norm()) for n in names)
    per = {n: _cos(g8[n].flatten().double(), go[n].flatten().double()) for n in names if float(go[n].double().norm()) > 1e-2 * big}
    worst = min(per, key=per.get)
    print(f"[dgrad8 full fine-tune, {'train' if train_mode else 'eval'}] {len(names)} parameters: cosine vs oracle(dgrad8) {_cos(f8, o8):.5f} (rel {float((f8 - o8).norm() / o8.norm()):.2e}), "
          f"worst of {len(per)} large parameters {per[worst]:.4f} ({worst}), vs the bf16 dgrad {_cos(f8, f16):.5f}")
    assert not torch.equal(f8, f16), "the switch did not reach the kernels"
    assert _cos(f8, o8) > 0.999 and per[worst] > 0.985 and _cos(f8, f16) > 0.995


def test_dgrad8_full_finetune_image_tower_matches_oracle(dev):
    """The same with the pre-LN stack: a width-768 ViT of three blocks (two full ones on the 8-bit dgrad, the class-row-only last block on bf16),
    batch 16, `disable_lora` (lora_layer = [], every parameter trainable) — the image half of the reference's final 5M recipe.  Against the
    oracle's dgrad8 rule with trainable weights and against the HIP tower's bf16 dgrad, over all parameters."""
    from oracle import clibd_oracle as O
    from clibd_amd.model import CLIBDImageEncoder, VisionTransformer

    torch.manual_seed(35)
    om = O.ImageEncoder(O.VisionTransformer(img_size=224, patch=16, dim=768, depth=3, heads=12, num_classes=0), 4, 768, lora_layer=[])
    with torch.no_grad():     # (`if lora_layer:` quirk of the reference, image_encoder.py:54-57: an empty list still wraps every block — the adapters train along)
        for n, p in om.named_parameters():
            if "linear_b_" in n:
                p.normal_(0, 0.02)
    m = CLIBDImageEncoder(VisionTransformer(embed_dim=768, depth=3, num_heads=12, num_classes=0), r=4, num_classes=768, lora_layer=[])
    m.load_state_dict(om.state_dict(), strict=True)
    for mod in (om, m):
        for p_ in mod.parameters():
            p_.requires_grad_(True)
    m = m.to(dev).eval()
    assert m.tower().stack.full_mode()
    g = torch.Generator().manual_seed(36)
    img, cot = torch.rand(16, 3, 224, 224, generator=g), torch.randn(16, 768, generator=g)
    res = {}
    for mode in ("bf16", "fp8"):
        m.tower().stack.set_numerics(dgrad=mode)
        y = m(img.to(dev))
        res[mode] = (y.detach().cpu(), _grads(m.named_parameters(), (y * cot.to(dev)).sum()))
    m.tower().stack.set_numerics(dgrad="bf16")
    assert torch.equal(res["bf16"][0], res["fp8"][0])
    with O.precision("bf16"), O.dgrad8(True):
        yo = om(img)
        ora = _grads(om.named_parameters(), (yo * cot).sum())
    live = {n for n, v in ora.items() if float(v.abs().max()) > 0}
    keep = lambda g_: {n: v for n, v in g_.items() if n in live and n in ora}
    g8, g16, go = keep(res["fp8"][1]), keep(res["bf16"][1]), keep(ora)
    names = sorted(go)
    assert any("mlp.fc1.weight" in n for n in names) and any("norm1.weight" in n for n in names), names[:8]
    f8, f16, o8 = _flat(g8, names), _flat(g16, names), _flat(go, names)
    big = max(float(go[n].double().norm()) for n in names)
    per = {n: _cos(g8[n].flatten().double(), go[n].flatten().double()) for n in names if float(go[n].double().norm()) > 1e-2 * big}
    worst = min(per, key=per.get)
    print(f"[dgrad8 full fine-tune, ViT width 768] {len(names)} parameters: cosine vs oracle(dgrad8) {_cos(f8, o8):.5f} (rel {float((f8 - o8).norm() / o8.norm()):.2e}), "
          f"worst of {len(per)} large parameters {per[worst]:.4f} ({worst}), vs the bf16 dgrad {_cos(f8, f16):.5f}")
    assert not torch.equal(f8, f16), "the switch did not reach the kernels"
    assert _cos(f8, o8) > 0.999 and per[worst] > 0.985 and _cos(f8, f16) > 0.995


def test_dgrad8_full_finetune_trainer_step(dev):
    """A Trainer step of a (small-depth, full-width) Image+DNA model with every weight trainable and the 8-bit dgrad on the mean-pooled
    tower: runs, learns, base weights move, and the per-layer d(fc1 out) scale is re-derived on the host every DGRAD8_C2_EVERY steps only."""
    from clibd_amd.data import synthetic_batch
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, VisionTransformer
    from clibd_amd.train import Trainer

    torch.manual_seed(35)
    model = SimpleCLIP(CLIBDImageEncoder(VisionTransformer(embed_dim=768, depth=2, num_heads=12, num_classes=0), r=4, num_classes=768, lora_layer=[]),
                       CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072)), r=4, num_classes=768, lora_layer=[]), None)
    for p_ in model.parameters():
        p_.requires_grad_(True)
    model = model.to(dev).train()
    model.enable_fp8_dgrad(towers="pooled")
    st = model.dna_encoder.tower().stack
    st.DGRAD8_C2_EVERY = 3
    tr = Trainer(model, lr=1e-4, world_size=1, rank=0, all_gather=True)
    batch = synthetic_batch(16, dev, seed=13, rank=0, with_text=False)
    w0 = model.dna_encoder.base_dna_encoder.bert.encoder.layer[0].intermediate.dense.weight.detach().clone()
    losses, ages = [], []
    for _ in range(5):
        losses.append(float(tr.step(batch["image"], batch["dna"], None, batch["labels"])))
        ages.append(st._c2_age)
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    assert ages == [1, 2, 3, 1, 2], ages
    assert not torch.equal(w0, model.dna_encoder.base_dna_encoder.bert.encoder.layer[0].intermediate.dense.weight.detach())
    assert model.numerics()["dna_encoder"]["dgrad"] == "fp8" and model.numerics()["image_encoder"]["dgrad"] == "bf16"


def test_dgrad8_needs_the_bf16_stream_and_supported_widths(dev):
    from clibd_amd.engine import NotSupportedYet
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder

    m = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=512, num_hidden_layers=1, num_attention_heads=8, intermediate_size=1024)), r=4, num_classes=64).to(dev)
    ids = torch.randint(3, 1027, (4, 133))
    m.tower().stack.set_numerics(dgrad="fp8", residual_grad="fp32")
    with pytest.raises(NotSupportedYet, match="dgrad=fp8"):
        m(ids.to(dev)).sum().backward()
    m.tower().stack.set_numerics(residual_grad="bf16")
    m(ids.to(dev)).sum().backward()          # H = 512, FF = 1024: the smallest shapes the fp8 kernel takes
    m3 = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256)), r=4, num_classes=64).to(dev)
    m3.tower().stack.set_numerics(dgrad="fp8")
    with pytest.raises(NotSupportedYet, match="hidden % 256"):
        m3(ids.to(dev)).sum().backward()


def test_dgrad8_falls_back_to_bf16_when_the_token_count_is_not_a_multiple_of_four(dev):
    """The e4m3 dgrad forms load a row group's four dequantisation factors as one vector (M % 4 == 0): a call with another token count
    (here 3 sequences of 133 tokens) takes the bf16 GEMMs — the same gradients as with the switch off."""
    from clibd_amd.model import BertConfigLite, BertForMaskedLM, CLIBDDNAEncoder

    torch.manual_seed(41)
    m = CLIBDDNAEncoder(BertForMaskedLM(BertConfigLite(vocab_size=1027, hidden_size=512, num_hidden_layers=2, num_attention_heads=8, intermediate_size=1024)), r=4, num_classes=64).to(dev).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".w_b." in n:
                p.normal_(0, 0.02)
    cot = torch.randn(4, 64, generator=torch.Generator().manual_seed(1)).to(dev)
    out = {}
    for B in (3, 4):
        ids = torch.randint(3, 1027, (B, 133), generator=torch.Generator().manual_seed(B)).to(dev)
        for mode in ("bf16", "fp8"):
            m.tower().stack.set_numerics(dgrad=mode)
            out[(B, mode)] = _grads(m.named_parameters(), (m(ids) * cot[:B]).sum())
    m.tower().stack.set_numerics(dgrad="bf16")
    # (at M = 399 the adapters' gradients are reduced with float atomics — the partials workspace needs M % 32 == 0 — so "the same" means
    #  to the order of those additions, 1e-6; the 8-bit dgrad moves the gradients by 1e-3 and more)
    dist = lambda B: max(float((out[(B, "bf16")][n] - out[(B, "fp8")][n]).norm() / (out[(B, "bf16")][n].norm() + 1e-30)) for n in out[(B, "bf16")])
    assert dist(3) < 1e-5, dist(3)
    assert dist(4) > 1e-4, dist(4)


def test_dgrad8_text_tower_with_a_padding_mask_matches_oracle(dev):
    """BERT-small (H = 512, FF = 2048 — the smallest shapes the fp8 kernel takes — S = 20 with a key mask, mean over all 20 positions): the third
    tower of the tri-modal configuration on the 8-bit dgrad against the oracle's restatement, and against its own bf16 dgrad."""
    from oracle import clibd_oracle as O
    from tests.test_model_gpu import _bert_small_pair, _text_batch

    om, m = _bert_small_pair(dev)
    x = _text_batch(64, 7)
    cot = torch.randn(64, 768, generator=torch.Generator().manual_seed(8))
    xd = {k: v.to(dev) for k, v in x.items()}
    res = {}
    for mode in ("bf16", "fp8"):
        m.tower().stack.set_numerics(dgrad=mode)
        y = m(xd)
        res[mode] = (y.detach().cpu(), _grads(m.named_parameters(), (y * cot.to(dev)).sum()))
    m.tower().stack.set_numerics(dgrad="bf16")
    assert torch.equal(res["bf16"][0], res["fp8"][0])
    with O.precision("bf16"), O.dgrad8(True):
        yo = om(x)
        ora = _grads(om.named_parameters(), (yo * cot).sum())
    keep = lambda g_: {n: v for n, v in g_.items() if n in ora}
    _compare(keep(res["fp8"][1]), keep(res["bf16"][1]), ora, "BERT-small with a padding mask", 0.999, 0.995)


def test_gemm_fp8_dgrad_takes_the_one_byte_gelu_grad(dev):
    """The fc2 dgrad's MUL_AUX form with gelu' as the one-byte code of numerics gelu_grad = "u8" (code * 1.265625 / 255 - 0.1328125): e4m3 bytes equal to a
    torch evaluation of the same decode, up to the places where the decode's last fp32 bit lands on the other side of an e4m3 rounding boundary."""
    from clibd_amd import ops

    M, N, K = 1108, 3072, 768
    g = torch.Generator().manual_seed(77)
    a, w = ints((M, K), -3, 3, g), ints((N, K), -2, 2, g)
    cs = 2.0 ** torch.randint(-3, 2, (N,), generator=g).float()
    codes = torch.randint(0, 256, (M, N), generator=g, dtype=torch.int64).to(torch.uint8)
    o8 = torch.empty((M, N), dtype=torch.uint8, device=dev).view(FP8)
    ops.gemm_fp8_dgrad_nt(a.to(FP8).to(dev), w.to(FP8).to(dev), cs.to(dev), aux=codes.to(dev), act=ops.ACT_MUL_AUX_U8, out_fp8=o8, out_fp8_scale=2.0 ** -5)
    gp = codes.float() * (1.265625 / 255.0) - 0.1328125
    want = ((a.double() @ w.double().T) * cs.double()).float() * gp * 2.0 ** -5
    wq, got = want.clamp(-448, 448).to(FP8).float(), o8.cpu().float()
    assert float((got != wq).float().mean()) < 5e-3
    assert bool(((got - wq).abs() <= torch.maximum(wq.abs() * 0.125, torch.tensor(2.0 ** -9))).all())
    with pytest.raises(TypeError):       # bf16 aux with the one-byte act
        ops.gemm_fp8_dgrad_nt(a.to(FP8).to(dev), w.to(FP8).to(dev), cs.to(dev), aux=torch.zeros((M, N), dtype=torch.bfloat16, device=dev), act=ops.ACT_MUL_AUX_U8,
                              out_fp8=o8, out_fp8_scale=1.0)


def test_dgrad8_with_the_one_byte_gelu_grad_in_a_tower(dev):
    """numerics dgrad = "fp8" together with gelu_grad = "u8" (ViT width 768, three blocks, batch 16): the gradients of the two-switch mode stay within the
    8-bit dgrad's own noise of the dgrad = "fp8" / bf16-gelu' mode (the one-byte code's error, 2.5e-3 absolute on gelu', is far below e4m3's)."""
    from clibd_amd.model import CLIBDImageEncoder, VisionTransformer

    torch.manual_seed(35)
    m = CLIBDImageEncoder(VisionTransformer(embed_dim=768, depth=3, num_heads=12, num_classes=0), r=4, num_classes=768)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "linear_b_" in n:
                p.normal_(0, 0.02)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(36)
    img, cot = torch.rand(16, 3, 224, 224, generator=g).to(dev), torch.randn(16, 768, generator=g).to(dev)
    res = {}
    for key, kw in (("bf16", dict(dgrad="bf16", gelu_grad="bf16")), ("fp8", dict(dgrad="fp8", gelu_grad="bf16")), ("fp8+u8", dict(dgrad="fp8", gelu_grad="u8")),
                    ("fp8+e4m7", dict(dgrad="fp8", gelu_grad="e4m7"))):
        m.tower().stack.set_numerics(**kw)
        y = m(img)
        res[key] = (y.detach().cpu(), _grads(m.named_parameters(), (y * cot).sum()))
    m.tower().stack.set_numerics(dgrad="bf16", gelu_grad="bf16")
    names = sorted(res["bf16"][1])
    f = {k: _flat(v[1], names) for k, v in res.items()}
    assert torch.equal(res["fp8"][0], res["fp8+u8"][0])                       # the forward's activations do not depend on how gelu' is stored
    assert not torch.equal(f["fp8"], f["fp8+u8"])
    c8, c8u = _cos(f["fp8"], f["bf16"]), _cos(f["fp8+u8"], f["bf16"])
    print(f"[dgrad8 + one-byte gelu'] cosine vs the bf16 dgrad: dgrad8 {c8:.6f}, dgrad8 + u8 {c8u:.6f}; between the two {_cos(f['fp8'], f['fp8+u8']):.6f}")
    assert c8u > 0.999 and c8u > c8 - 2e-4
    # round 6: the twelve-bit form IS the bf16 value (flushed below 2^-14): the 8-bit dgrad's e4m3 operand is the same up to that tail
    assert float((f["fp8+e4m7"] - f["fp8"]).abs().max()) <= 1e-4 * float(f["fp8"].abs().max()) and _cos(f["fp8+e4m7"], f["fp8"]) > 0.999999
