"""Kernel-level parity (GPU): every C-ABI op against a plain PyTorch fp32/fp64 statement of the same op.

All ops are called through clibd_amd.ops -> ctypes -> libclibd_hip.so.  Integer-valued operands make the
MFMA layout checks exact (asymmetric operands so a transposed fragment cannot pass)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def bfr(x):
    """round an fp32 tensor to bf16 precision (kept in fp32)"""
    return x.to(BF16).to(F32)


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def rel_err(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.fixture(scope="module")
def ops(dev):
    from clibd_amd import ops as _ops

    return _ops


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (77, 48, 192), (512, 768, 768), (1, 16, 64)])
def test_gemm_exact_integers(ops, dev, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    # asymmetric structure: distinct row/col ramps
    a[:, 0] += torch.arange(M).float() % 5
    w[:, 1] += torch.arange(N).float() % 7
    ref = a.double() @ w.double().T
    out = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), out_f32=out)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().double(), ref)


def test_gemm_strided_operands_and_bf16_out(ops, dev):
    g = torch.Generator().manual_seed(1)
    M, N, K = 200, 128, 128
    abig = torch.randn(M, 3 * K, generator=g)
    w = torch.randn(N, K, generator=g)
    a = abig[:, K : 2 * K]
    ref = bfr(a) @ bfr(w).T
    outbig = torch.zeros((M, 2 * N), dtype=BF16, device=dev)
    ad = abig.to(dev, BF16)[:, K : 2 * K]
    ops.gemm_nt(ad, w.to(dev, BF16), out_bf16=outbig[:, N:])
    torch.cuda.synchronize()
    got = outbig.cpu().float()
    assert torch.equal(got[:, :N], torch.zeros(M, N))
    assert rel_err(got[:, N:], ref) < 4e-3


def test_gemm_epilogue_bias_gelu_pre(ops, dev):
    g = torch.Generator().manual_seed(2)
    M, N, K = 260, 384, 256
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    pre_ref = bfr(bfr(a) @ bfr(w).T + b)
    act_ref = gelu(pre_ref)
    pre = torch.empty((M, N), dtype=BF16, device=dev)
    act = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), bias=b.to(dev), act=ops.ACT_GELU, out_pre=pre, out_bf16=act)
    torch.cuda.synchronize()
    assert rel_err(pre.cpu().float(), pre_ref) < 3e-3
    assert rel_err(act.cpu().float(), act_ref) < 4e-3
    # gelu is applied to the *rounded* pre-activation the kernel stored
    assert (act.cpu().float() - bfr(gelu(pre.cpu().float()))).abs().max().item() < 2e-2


def test_gemm_epilogue_gelu_grad_and_residual(ops, dev):
    g = torch.Generator().manual_seed(3)
    M, N, K = 130, 256, 128
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    aux = torch.randn(M, N, generator=g) * 2
    res = torch.randn(M, N, generator=g)
    ref = (bfr(a) @ bfr(w).T) * gelu_grad(bfr(aux)) + res
    outf = torch.empty((M, N), dtype=F32, device=dev)
    outb = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), act=ops.ACT_GELU_GRAD, aux=aux.to(dev, BF16), residual=res.to(dev), out_f32=outf,
                out_bf16=outb)
    torch.cuda.synchronize()
    assert rel_err(outf.cpu(), ref) < 1e-4
    assert rel_err(outb.cpu().float(), ref) < 4e-3


def test_gemm_rank8_update(ops, dev):
    g = torch.Generator().manual_seed(4)
    M, N, K = 150, 256, 128
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    u = torch.randn(M, 16, generator=g)  # row stride 16, only the first 8 columns are operands
    v = torch.randn(N, 8, generator=g)
    ref = bfr(a) @ bfr(w).T + bfr(u[:, :8]) @ bfr(v).T
    out = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), rank_u=u.to(dev, BF16), rank_v=v.to(dev, BF16), out_f32=out)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 1e-5


def test_gemm_split_k_accumulates(ops, dev):
    g = torch.Generator().manual_seed(5)
    M, N, K = 96, 64, 1024
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    ref = bfr(a) @ bfr(w).T
    out = torch.zeros((M, N), dtype=F32, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), out_f32=out, split_k=4)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 1e-5


def test_gemm_rejects_bad_shapes(ops, dev):
    from clibd_amd._lib import ClibdHipError

    a = torch.zeros((8, 96), dtype=BF16, device=dev)  # K not a multiple of 64
    w = torch.zeros((16, 96), dtype=BF16, device=dev)
    with pytest.raises(ClibdHipError):
        ops.gemm_nt(a, w, out_f32=torch.empty((8, 16), dtype=F32, device=dev))
    with pytest.raises(ValueError):
        ops.gemm_nt(torch.zeros((8, 64), dtype=BF16), torch.zeros((16, 64), dtype=BF16), out_f32=torch.empty((8, 16)))


def test_transpose_and_casts(ops, dev):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(133, 200, generator=g)
    xt = ops.transpose_bf16(x.to(dev, BF16))
    torch.cuda.synchronize()
    assert xt.shape == (200, 192)
    assert torch.equal(xt.cpu().float()[:, :133], bfr(x).T)
    assert torch.equal(xt.cpu().float()[:, 133:], torch.zeros(200, 59))
    y = torch.randn(70, 130, generator=g)
    assert torch.equal(ops.cast_bf16(y.to(dev)).cpu().float(), bfr(y))
    assert torch.equal(ops.cast_transpose_bf16(y.to(dev)).cpu().float(), bfr(y).T)


# ----------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("H,eps", [(768, 1e-6), (512, 1e-12), (128, 1e-12), (1024, 1e-5)])
def test_layernorm_fwd_bwd(ops, dev, H, eps):
    g = torch.Generator().manual_seed(H)
    M = 37
    x = torch.randn(M, H, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(H, generator=g), torch.randn(H, generator=g)
    acat = torch.randn(8, H, generator=g) * 0.05
    xd = x.double().requires_grad_(True)
    yref = torch.nn.functional.layer_norm(xd, (H,), gamma.double(), beta.double(), eps)
    yb = torch.empty((M, H), dtype=BF16, device=dev)
    yf = torch.empty((M, H), dtype=F32, device=dev)
    stats = torch.empty((M, 2), dtype=F32, device=dev)
    t = torch.empty((M, 8), dtype=BF16, device=dev)
    ops.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev), eps, y_bf16=yb, y_f32=yf, stats=stats, lora_a=acat.to(dev, BF16), t_out=t)
    torch.cuda.synchronize()
    assert (yf.cpu().double() - yref.detach()).abs().max().item() < 2e-5
    assert torch.equal(yb.cpu().float(), bfr(yf.cpu()))
    tref = bfr(yf.cpu()) @ bfr(acat).T
    assert (t.cpu().float() - tref).abs().max().item() < 2e-2 * tref.abs().max().item() + 1e-3
    mean, var = x.double().mean(1), x.double().var(1, unbiased=False)
    assert (stats.cpu()[:, 0].double() - mean).abs().max().item() < 1e-5
    assert rel_err(stats.cpu()[:, 1], 1.0 / torch.sqrt(var + eps)) < 1e-5
    # backward: dx = LN'(dy) + dres
    dy = torch.randn(M, H, generator=g)
    dres = torch.randn(M, H, generator=g)
    (gref,) = torch.autograd.grad(yref, xd, dy.double())
    dxf = torch.empty((M, H), dtype=F32, device=dev)
    dxb = torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), stats, gamma.to(dev), dres=dres.to(dev), dx_f32=dxf, dx_bf16=dxb)
    torch.cuda.synchronize()
    assert rel_err(dxf.cpu(), gref + dres.double()) < 2e-5
    assert torch.equal(dxb.cpu().float(), bfr(dxf.cpu()))
    # bf16 upstream gradient
    (gref2,) = torch.autograd.grad(torch.nn.functional.layer_norm(xd, (H,), gamma.double(), beta.double(), eps), xd, bfr(dy).double())
    ops.layernorm_bwd(dy.to(dev, BF16), x.to(dev), stats, gamma.to(dev), dx_f32=dxf)
    torch.cuda.synchronize()
    assert rel_err(dxf.cpu(), gref2) < 2e-5


# ----------------------------------------------------------------------------------------------- attention
def _attn_ref(qkv, B, S, nh, mask):
    """fp32 statement: scores fp32, softmax fp32, P rounded to bf16 (straight-through), PV fp32."""
    H = nh * 64
    q, k, v = qkv.view(B, S, 3, nh, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    p = torch.softmax(s, dim=-1)
    pb = p + (p.to(BF16).to(p.dtype) - p).detach()
    o = pb @ v
    return o.permute(0, 2, 1, 3).reshape(B * S, H)


@pytest.mark.parametrize("B,S,nh,masked", [(2, 197, 2, False), (3, 133, 3, False), (4, 20, 2, True), (1, 32, 1, False), (2, 64, 1, True),
                                           (1, 256, 1, False), (2, 7, 1, False),
                                           # nine-tile sequences (S in (128, 144]) take the three-wave workgroups; 129 and 144 are its edges, 150 / 160 stay on four
                                           (2, 129, 2, False), (5, 144, 3, False), (2, 150, 2, False), (1, 160, 1, False), (40, 133, 12, False),
                                           (3, 140, 2, True)])   # a key mask on a nine-tile sequence: three-wave forward, four-wave (masked) backward
def test_attention_fwd_bwd(ops, dev, B, S, nh, masked):
    g = torch.Generator().manual_seed(S * 3 + nh)
    H = nh * 64
    qkv = bfr(torch.randn(B * S, 3 * H, generator=g))
    mask = None
    if masked:
        lens = torch.randint(max(1, S // 3), S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int32)
    qd = qkv.double().requires_grad_(True)
    oref = _attn_ref(qd, B, S, nh, mask)
    out = torch.empty((B * S, H), dtype=BF16, device=dev)
    ops.attention_fwd(qkv.to(dev, BF16), B, S, nh, None if mask is None else mask.to(dev), out)
    torch.cuda.synchronize()
    assert rel_err(out.cpu().float(), oref.detach()) < 5e-3
    assert (out.cpu().float() - oref.detach().float()).abs().max().item() < 3e-2
    do = bfr(torch.randn(B * S, H, generator=g))
    (gref,) = torch.autograd.grad(oref, qd, do.double())
    dqkv = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=dev)
    ops.attention_bwd(qkv.to(dev, BF16), do.to(dev, BF16), B, S, nh, None if mask is None else mask.to(dev), dqkv)
    torch.cuda.synchronize()
    got = dqkv.cpu().float()
    assert torch.isfinite(got).all()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert rel_err(got[:, sl], gref[:, sl]) < 1.5e-2, name


@pytest.mark.parametrize("B,S,nh,p_drop", [(2, 197, 2, 0.0), (3, 133, 3, 0.1), (2, 224, 1, 0.0), (5, 20, 2, 0.0), (1, 7, 1, 0.1), (2, 100, 12, 0.0),
                                            (40, 197, 12, 0.0)])
def test_attention_single_pass_backward(ops, dev, B, S, nh, p_drop):
    """clibd_attention_fwd_save + clibd_attention_bwd_sp: the training forward also writes the log-sum-exp and the rounding
    residual of its output (out unchanged, bit for bit), and the single-pass backward built on them must give the gradients of
    the fp64 statement (same gate as the two-phase kernel) and agree with the two-phase kernel — under dropout too, where both
    evaluate the same counter-based masks.  B x heads >= 2 x CUs at S = 197 takes the persistent forward."""
    g = torch.Generator().manual_seed(S * 5 + nh)
    H = nh * 64
    qkv = bfr(torch.randn(B * S, 3 * H, generator=g))
    do = bfr(torch.randn(B * S, H, generator=g))
    qkv_d, do_d = qkv.to(dev, BF16), do.to(dev, BF16)
    drop = ops.Drop(p_drop, 777) if p_drop > 0 else None
    out = torch.empty((B * S, H), dtype=BF16, device=dev)
    ops.attention_fwd(qkv_d, B, S, nh, None, out, drop=drop)
    out2 = torch.full((B * S, H), float("nan"), dtype=BF16, device=dev)
    o_lo = torch.full((B * S, H), float("nan"), dtype=BF16, device=dev)
    lse = torch.full((B * nh * S,), float("nan"), dtype=F32, device=dev)
    ops.attention_fwd(qkv_d, B, S, nh, None, out2, drop=drop, lse=lse, o_lo=o_lo)
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and torch.isfinite(o_lo.float()).all() and torch.isfinite(lse).all()
    nb = min(B, 3)                                       # the reference on the first sequences only (the big case is 40 x 12 heads)
    q, k, _ = qkv[: nb * S].double().view(nb, S, 3, nh, 64).permute(2, 0, 3, 1, 4)
    sc = (q @ k.transpose(-1, -2)) * 0.125
    lse_ref = torch.logsumexp(sc, dim=-1) * 1.4426950408889634        # log2 domain, [nb, nh, S]
    assert (lse.cpu().view(B, nh, S)[:nb].double() - lse_ref).abs().max().item() < 2e-3
    if p_drop == 0.0:
        oref = _attn_ref(qkv[: nb * S].double(), nb, S, nh, None)
        e_hi = (out[: nb * S].cpu().double() - oref).abs().max().item()
        e_sum = (out[: nb * S].cpu().double() + o_lo[: nb * S].cpu().double() - oref).abs().max().item()
        assert e_sum < 1.05 * e_hi + 1e-5, (e_hi, e_sum)       # hi + lo is the kernel's fp32 output: no further from the statement than hi alone
        oh, ol = out.cpu().float(), o_lo.cpu().float()
        assert bool((ol.abs() <= oh.abs() * 2.0 ** -8 + 1e-30).all())           # the residual of a round-to-nearest: at most half an ulp of hi
    dq_sp = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=dev)
    ops.attention_bwd_sp(qkv_d, do_d, out, o_lo, lse, B, S, nh, dq_sp, drop=drop)
    dq_2p = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=dev)
    ops.attention_bwd(qkv_d, do_d, B, S, nh, None, dq_2p, drop=drop)
    torch.cuda.synchronize()
    got, old = dq_sp.cpu().float(), dq_2p.cpu().float()
    assert torch.isfinite(got).all()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert rel_err(got[:, sl], old[:, sl]) < 1.2e-2, name          # two roundings of the same gradient
    if p_drop == 0.0:
        qd = qkv[: nb * S].double().requires_grad_(True)
        (gref,) = torch.autograd.grad(_attn_ref(qd, nb, S, nh, None), qd, do[: nb * S].double())
        for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
            e_new, e_old = rel_err(got[: nb * S, sl], gref[:, sl]), rel_err(old[: nb * S, sl], gref[:, sl])
            assert e_new < 1.5e-2 and e_new < 1.5 * e_old + 1e-3, (name, e_new, e_old)
    again = torch.empty_like(dq_sp)
    ops.attention_bwd_sp(qkv_d, do_d, out, o_lo, lse, B, S, nh, again, drop=drop)
    torch.cuda.synchronize()
    assert torch.equal(again, dq_sp)                                        # no atomics: bit-reproducible
    if S > 224 - 32:
        with pytest.raises(RuntimeError, match="224"):
            big = torch.zeros((256, 3 * 64), dtype=BF16, device=dev)
            ops.attention_bwd_sp(big, big[:, :64].contiguous(), big[:, :64].contiguous(), big[:, :64].contiguous(),
                                 torch.zeros(256, device=dev), 1, 256, 1, torch.empty_like(big))


@pytest.mark.parametrize("S,masked,p_drop", [(197, False, 0.0), (133, False, 0.1), (224, True, 0.0), (170, True, 0.1), (256, False, 0.0)])
def test_attention_persistent_kernels_match_per_head_kernels(ops, dev, S, masked, p_drop):
    """A forward launch with many heads (B * heads >= 2 x CUs) and S > 160 takes the persistent kernel (one workgroup of 16 waves per
    CU walking heads, K / V double-buffered); few heads take the per-head kernel.  The arithmetic per element is the same, so the
    first sequences of a big launch must equal a small launch on those sequences bit for bit — masks and dropout included (the
    dropout index is a function of the global head, so the small launch covers the same heads).  The backward (one kernel family)
    is held to the same batch-split invariance."""
    B, nh, Bs = 48, 12, 2
    H = nh * 64
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.7).to(dev, BF16)
    do = torch.randn(B * S, H, generator=g).to(dev, BF16)
    mask = None
    if masked:
        lens = torch.randint(S // 3, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int32).to(dev)
    drop = ops.Drop(p_drop, 1234) if p_drop > 0 else None
    out = torch.empty((B * S, H), dtype=BF16, device=dev)
    dqkv = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=dev)
    ops.attention_fwd(qkv, B, S, nh, mask, out, drop=drop)
    ops.attention_bwd(qkv, do, B, S, nh, mask, dqkv, drop=drop)
    out_s = torch.empty((Bs * S, H), dtype=BF16, device=dev)
    dqkv_s = torch.full((Bs * S, 3 * H), float("nan"), dtype=BF16, device=dev)
    ms = None if mask is None else mask[:Bs].contiguous()
    ops.attention_fwd(qkv[: Bs * S].contiguous(), Bs, S, nh, ms, out_s, drop=drop)
    ops.attention_bwd(qkv[: Bs * S].contiguous(), do[: Bs * S].contiguous(), Bs, S, nh, ms, dqkv_s, drop=drop)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
    assert torch.equal(out[: Bs * S], out_s)
    assert torch.equal(dqkv[: Bs * S], dqkv_s)
    # and against torch on the unmasked, undropped case
    if not masked and p_drop == 0.0:
        qd = qkv[: Bs * S].cpu().double().requires_grad_(True)
        oref = _attn_ref(qd, Bs, S, nh, None)
        assert rel_err(out[: Bs * S].cpu().float(), oref.detach()) < 5e-3


@pytest.mark.parametrize("S,nq", [(197, 1), (224, 40), (200, 17)])
def test_attention_persistent_forward_query_prefix(ops, dev, S, nq):
    """The [CLS]-only last ViT block at the metric's batch is a persistent-kernel launch with nq = 1: only the first nq query rows
    are evaluated (`out` is [B * nq, H]).  Bit-equal to the per-head kernel on the same sequences, and the rows beyond B * nq of a
    larger buffer stay untouched."""
    B, nh, Bs = 48, 12, 2
    H = nh * 64
    g = torch.Generator().manual_seed(S + nq)
    qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.7).to(dev, BF16)
    buf = torch.full((B * nq + 8, H), 7.0, dtype=BF16, device=dev)
    out = buf[: B * nq]
    ops.attention_fwd(qkv, B, S, nh, None, out, nq=nq)
    out_s = torch.empty((Bs * nq, H), dtype=BF16, device=dev)
    ops.attention_fwd(qkv[: Bs * S].contiguous(), Bs, S, nh, None, out_s, nq=nq)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out[: Bs * nq], out_s)
    assert bool((buf[B * nq :].float() == 7.0).all())
    qd = qkv[: Bs * S].cpu().double()
    oref = _attn_ref(qd, Bs, S, nh, None).view(Bs, S, H)[:, :nq].reshape(Bs * nq, H)
    assert rel_err(out_s.cpu().float(), oref) < 5e-3


# ----------------------------------------------------------------------------------------------- LoRA
def test_lora_pack_and_wgrad(ops, dev):
    g = torch.Generator().manual_seed(11)
    H, M = 128, 700
    a_q, a_v = torch.randn(4, H, generator=g), torch.randn(4, H, generator=g)
    b_q, b_v = torch.randn(H, 4, generator=g), torch.randn(H, 4, generator=g)
    v_fwd = torch.empty((3 * H, 8), dtype=BF16, device=dev)
    v_bwd = torch.empty((H, 8), dtype=BF16, device=dev)
    a_cat = torch.empty((8, H), dtype=BF16, device=dev)
    w_dt = torch.empty((16, 3 * H), dtype=BF16, device=dev)
    ops.lora_pack(a_q.to(dev), a_v.to(dev), b_q.to(dev), b_v.to(dev), v_fwd, v_bwd, a_cat, w_dt)
    torch.cuda.synchronize()
    vf = torch.zeros(3 * H, 8)
    vf[:H, :4] = bfr(b_q)
    vf[2 * H :, 4:] = bfr(b_v)
    assert torch.equal(v_fwd.cpu().float(), vf)
    assert torch.equal(v_bwd.cpu().float(), torch.cat([bfr(a_q).T, bfr(a_v).T], dim=1))
    assert torch.equal(a_cat.cpu().float(), torch.cat([bfr(a_q), bfr(a_v)], dim=0))
    wd = torch.zeros(16, 3 * H)
    wd[:4, :H] = bfr(b_q).T
    wd[4:8, 2 * H :] = bfr(b_v).T
    assert torch.equal(w_dt.cpu().float(), wd)

    dqkv = bfr(torch.randn(M, 3 * H, generator=g))
    x = bfr(torch.randn(M, H, generator=g))
    t = bfr(torch.randn(M, 8, generator=g))
    # dt through the GEMM against w_dt (what the backward pass does)
    dt = torch.empty((M, 16), dtype=BF16, device=dev)
    ops.gemm_nt(dqkv.to(dev, BF16), w_dt, out_bf16=dt)
    torch.cuda.synchronize()
    dt_ref = torch.cat([dqkv[:, :H] @ bfr(b_q), dqkv[:, 2 * H :] @ bfr(b_v)], dim=1)
    assert rel_err(dt.cpu().float()[:, :8], dt_ref) < 4e-3
    assert torch.equal(dt.cpu().float()[:, 8:], torch.zeros(M, 8))
    dt2 = torch.full((M, 16), float("nan"), dtype=BF16, device=dev)   # as the engine calls it: the k segment is skipped
    ops.gemm_nt(dqkv.to(dev, BF16), w_dt, out_bf16=dt2, k_hole=(H, H))
    torch.cuda.synchronize()
    assert torch.equal(dt2, dt)
    dA_q, dA_v = torch.zeros((4, H), device=dev), torch.zeros((4, H), device=dev)
    dB_q, dB_v = torch.zeros((H, 4), device=dev), torch.zeros((H, 4), device=dev)
    ops.lora_wgrad(dqkv.to(dev, BF16), x.to(dev, BF16), t.to(dev, BF16), dt, dA_q, dA_v, dB_q, dB_v)
    torch.cuda.synchronize()
    dtf = dt.cpu().float()
    assert rel_err(dB_q.cpu(), dqkv[:, :H].T @ t[:, :4]) < 1e-5
    assert rel_err(dB_v.cpu(), dqkv[:, 2 * H :].T @ t[:, 4:]) < 1e-5
    assert rel_err(dA_q.cpu(), dtf[:, :4].T @ x) < 1e-5
    assert rel_err(dA_v.cpu(), dtf[:, 4:8].T @ x) < 1e-5


@pytest.mark.parametrize("M,H", [(8192, 768), (8192 + 32 * 37, 512), (16384, 1024), (8192, 384)])
def test_lora_wgrad_mfma_form(ops, dev, M, H):
    """Large M (whole 32-token slabs, H a multiple of 128) takes the MFMA form of the adapter weight gradients; same contract as
    the VALU kernel (accumulates into the caller's buffers), checked against fp64 products of the same bf16 operands."""
    g = torch.Generator().manual_seed(M + H)
    dqkv = bfr(torch.randn(M, 3 * H, generator=g))
    x = bfr(torch.randn(M, H, generator=g))
    t = bfr(torch.randn(M, 8, generator=g))
    dt = bfr(torch.randn(M, 16, generator=g))
    dt[:, 8:] = 0
    init = [torch.randn(4, H, generator=g), torch.randn(4, H, generator=g), torch.randn(H, 4, generator=g), torch.randn(H, 4, generator=g)]
    dA_q, dA_v, dB_q, dB_v = [v.clone().to(dev) for v in init]
    ops.lora_wgrad(dqkv.to(dev, BF16), x.to(dev, BF16), t.to(dev, BF16), dt.to(dev, BF16), dA_q, dA_v, dB_q, dB_v)
    torch.cuda.synchronize()
    d64, x64, t64, g64 = dqkv.double(), x.double(), t.double(), dt.double()
    refs = [g64[:, :4].T @ x64, g64[:, 4:8].T @ x64, d64[:, :H].T @ t64[:, :4], d64[:, 2 * H :].T @ t64[:, 4:]]
    for name, got, ref, i0 in zip(("dA_q", "dA_v", "dB_q", "dB_v"), (dA_q, dA_v, dB_q, dB_v), refs, init):
        assert rel_err(got.cpu().double() - i0.double(), ref) < 2e-5, name


@pytest.mark.parametrize("M,H", [(131072, 768), (131072 + 32 * 41, 512), (131072, 1024), (8192, 768), (8192, 384), (700, 128)])
def test_lora_backward_one_pass_over_dq_dv(ops, dev, M, H):
    """clibd_lora_backward: dt = dqkv . w_dt^T and the adapters' parameter gradients in one call.  Large M (>= 131 072 whole slabs' worth of rows) with H % 256 == 0 takes the
    fused kernels (dq, dv read once for dt and dB; dA from x and the finished dt), the rest the skinny GEMM + weight-gradient kernels;
    both against fp64 products of the same bf16 operands, dt bit-equal to the GEMM's (fp32 accumulation in another order: a bf16 ulp)."""
    g = torch.Generator().manual_seed(M + H)
    a_q, a_v = torch.randn(4, H, generator=g), torch.randn(4, H, generator=g)
    b_q, b_v = torch.randn(H, 4, generator=g) * 0.2, torch.randn(H, 4, generator=g) * 0.2
    v_fwd = torch.empty((3 * H, 8), dtype=BF16, device=dev); v_bwd = torch.empty((H, 8), dtype=BF16, device=dev)
    a_cat = torch.empty((8, H), dtype=BF16, device=dev); w_dt = torch.empty((16, 3 * H), dtype=BF16, device=dev)
    ops.lora_pack(a_q.to(dev), a_v.to(dev), b_q.to(dev), b_v.to(dev), v_fwd, v_bwd, a_cat, w_dt)
    dqkv = bfr(torch.randn(M, 3 * H, generator=g))
    x = bfr(torch.randn(M, H, generator=g))
    t = bfr(torch.randn(M, 8, generator=g))
    init = [torch.randn(4, H, generator=g), torch.randn(4, H, generator=g), torch.randn(H, 4, generator=g), torch.randn(H, 4, generator=g)]
    dA_q, dA_v, dB_q, dB_v = [v.clone().to(dev) for v in init]
    dt = torch.full((M, 16), float("nan"), dtype=BF16, device=dev)
    ops.lora_backward(dqkv.to(dev, BF16), x.to(dev, BF16), t.to(dev, BF16), w_dt, dt, dA_q, dA_v, dB_q, dB_v)
    dt_gemm = torch.empty((M, 16), dtype=BF16, device=dev)
    ops.gemm_nt(dqkv.to(dev, BF16), w_dt, out_bf16=dt_gemm, k_hole=(H, H))
    torch.cuda.synchronize()
    dt_ref = torch.cat([dqkv[:, :H].double() @ bfr(b_q).double(), dqkv[:, 2 * H :].double() @ bfr(b_v).double()], dim=1)
    dtf = dt.cpu().float()
    assert torch.equal(dtf[:, 8:], torch.zeros(M, 8))
    assert rel_err(dtf[:, :8].double(), dt_ref) < 4e-3
    assert rel_err(dtf, dt_gemm.cpu().float()) < 2e-3     # both are bf16 roundings of the same fp32-accumulated products
    d64, x64, t64, g64 = dqkv.double(), x.double(), t.double(), dtf.double()
    refs = [g64[:, :4].T @ x64, g64[:, 4:8].T @ x64, d64[:, :H].T @ t64[:, :4], d64[:, 2 * H :].T @ t64[:, 4:]]
    for name, got, ref, i0 in zip(("dA_q", "dA_v", "dB_q", "dB_v"), (dA_q, dA_v, dB_q, dB_v), refs, init):
        assert rel_err(got.cpu().double() - i0.double(), ref) < 2e-5, name


@pytest.mark.parametrize("M", [50432, 262144])
def test_lora_backward_with_a_partials_workspace_is_deterministic(ops, dev, M):
    """Round 5 (ABI 3): with a workspace (the default of ops.lora_backward / ops.lora_wgrad) the MFMA forms of the adapters' gradient
    kernels write per-workgroup partials and a last kernel adds them in a fixed order — no contended float atomics.  The gradients are
    then bit-identical run to run (the float-atomic form, workspace=False, is not in general) and equal to the atomic form's to
    summation order; dt is unaffected.  M = 50 432 takes the two-call path (skinny GEMM + weight-gradient kernel), M = 262 144 the fused one."""
    H = 768
    g = torch.Generator().manual_seed(M)
    w_dt = (torch.randn(16, 3 * H, generator=g) * 0.1).to(dev, BF16)
    w_dt[8:] = 0
    w_dt[:, H:2 * H] = 0
    dqkv = torch.randn(M, 3 * H, generator=g).to(dev, BF16)
    x = torch.randn(M, H, generator=g).to(dev, BF16)
    t = torch.randn(M, 8, generator=g).to(dev, BF16)

    def run(workspace):
        grads = [torch.zeros((4, H), device=dev), torch.zeros((4, H), device=dev), torch.zeros((H, 4), device=dev), torch.zeros((H, 4), device=dev)]
        dt = torch.empty((M, 16), dtype=BF16, device=dev)
        ops.lora_backward(dqkv, x, t, w_dt, dt, *grads, workspace=workspace)
        torch.cuda.synchronize()
        return [v.clone() for v in grads] + [dt]

    a, b_, c = run(None), run(None), run(False)
    for u, v in zip(a, b_):
        assert torch.equal(u, v), "the workspace form must be bit-reproducible"
    assert torch.equal(a[4], c[4])
    for name, u, v in zip(("dA_q", "dA_v", "dB_q", "dB_v"), a, c):
        assert rel_err(u.cpu().double(), v.cpu().double()) < 2e-6, name
    own = torch.empty((int(ops._lib.load().clibd_lora_workspace_bytes(M, H)),), dtype=torch.uint8, device=dev)   # the caller's buffer
    d = run(own)
    for u, v in zip(a, d):
        assert torch.equal(u, v)
    with pytest.raises(ValueError):
        ops.lora_backward(dqkv, x, t, w_dt, torch.empty((M, 16), dtype=BF16, device=dev), *[torch.zeros_like(v) for v in a[:4]], workspace=own[:1024])


# ----------------------------------------------------------------------------------------------- embeddings / heads
def test_patchify_matches_conv_unfold(ops, dev):
    g = torch.Generator().manual_seed(12)
    B = 2
    img = torch.rand(B, 3, 224, 224, generator=g)
    ref = torch.nn.functional.unfold(img, kernel_size=16, stride=16).transpose(1, 2).reshape(B * 196, 768)
    got = ops.patchify(img.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu().float(), bfr(ref))


def test_patchify_from_image_bytes_equals_the_host_side_totensor(ops, dev):
    """clibd_patchify_u8: the dataset's uint8 images, divided by 255 on the device, give the patch matrix of the reference's host-side
    ToTensor (u8.float() / 255, util/dataset.py:185-195) bit for bit — every one of the 256 byte values occurs."""
    g = torch.Generator().manual_seed(14)
    B = 3
    img8 = torch.randint(0, 256, (B, 3, 224, 224), generator=g, dtype=torch.uint8)
    img8.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
    f32 = img8.float() / 255.0
    got8 = ops.patchify(img8.to(dev))
    got32 = ops.patchify(f32.to(dev))
    ref = torch.nn.functional.unfold(f32, kernel_size=16, stride=16).transpose(1, 2).reshape(B * 196, 768)
    torch.cuda.synchronize()
    assert torch.equal(got8.cpu().view(torch.int16), got32.cpu().view(torch.int16))
    assert torch.equal(got8.cpu().float(), bfr(ref))
    with pytest.raises(TypeError):
        ops.patchify(img8.to(dev).to(torch.int32))


def test_vit_assemble_gelu_bwd_and_bert_embed(ops, dev):
    g = torch.Generator().manual_seed(13)
    B, S, H = 3, 5, 64
    cls, pos = torch.randn(H, generator=g), torch.randn(S, H, generator=g)
    proj = torch.randn(B * (S - 1), H, generator=g)
    tok = ops.vit_assemble_tokens(proj.to(dev), cls.to(dev), pos.to(dev), B)
    torch.cuda.synchronize()
    ref = torch.cat([cls.expand(B, 1, H), proj.view(B, S - 1, H)], dim=1) + pos[None]
    assert torch.equal(tok.cpu(), ref)
    dyv, pre = bfr(torch.randn(40, 64, generator=g)), bfr(torch.randn(40, 64, generator=g) * 2)
    dxv = ops.gelu_bwd(dyv.to(dev, BF16), pre.to(dev, BF16))
    torch.cuda.synchronize()
    assert rel_err(dxv.cpu().float(), dyv * gelu_grad(pre)) < 4e-3
    V = 50
    ids = torch.randint(0, V, (B, S), generator=g)
    tt = torch.randint(0, 2, (B, S), generator=g)
    word, posw, typ = torch.randn(V, H, generator=g), torch.randn(16, H, generator=g), torch.randn(2, H, generator=g)
    out = torch.empty((B * S, H), device=dev)
    ops.bert_embed(ids.to(dev), tt.to(dev), word.to(dev), posw.to(dev), typ.to(dev), out)
    torch.cuda.synchronize()
    ref = word[ids] + posw[:S][None] + typ[tt]
    assert torch.equal(out.cpu(), ref.reshape(B * S, H))
    ops.bert_embed(ids.to(dev), None, word.to(dev), posw.to(dev), typ.to(dev), out)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), (word[ids] + posw[:S][None] + typ[0]).reshape(B * S, H))


@pytest.mark.parametrize("B,S,C", [(3, 133, 768), (2, 5, 128), (1, 20, 1024)])
def test_softmax_mean_fwd_bwd(ops, dev, B, S, C):
    g = torch.Generator().manual_seed(C + S)
    logits = bfr(torch.randn(B * S, C, generator=g) * 3)
    ld = logits.double().requires_grad_(True)
    ref = torch.softmax(ld.view(B, S, C), dim=-1).mean(dim=1)
    out = ops.softmax_mean_fwd(logits.to(dev, BF16), B, S)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref.detach()) < 1e-5
    assert (out.cpu().sum(1) - 1).abs().max().item() < 1e-5
    dout = torch.randn(B, C, generator=g)
    (gref,) = torch.autograd.grad(ref, ld, dout.double())
    dl = ops.softmax_mean_bwd(logits.to(dev, BF16), dout.to(dev), B, S)
    torch.cuda.synchronize()
    assert rel_err(dl.cpu().float(), gref) < 4e-3


def test_token_mean_colsum_gather_scatter(ops, dev):
    g = torch.Generator().manual_seed(14)
    B, S, H = 4, 20, 512
    x = torch.randn(B, S, H, generator=g)
    m = ops.token_mean_fwd(x.to(dev))
    torch.cuda.synchronize()
    assert (m.cpu().float() - bfr(x.mean(1))).abs().max().item() < 1e-2
    d = torch.randn(B, H, generator=g)
    dx = ops.token_mean_bwd(d.to(dev), S)
    torch.cuda.synchronize()
    assert torch.allclose(dx.cpu(), (d / S)[:, None, :].expand(B, S, H), rtol=1e-6, atol=1e-7)
    y = bfr(torch.randn(1000, 200, generator=g))
    cs = torch.ones(200, device=dev)
    ops.colsum_bf16(y.to(dev, BF16), cs)
    torch.cuda.synchronize()
    assert rel_err(cs.cpu(), y.sum(0) + 1) < 1e-5
    assert torch.equal(ops.gather_rows(x.to(dev)).cpu(), x[:, 0])
    sb, sf = ops.scatter_rows(d.to(dev), S, bf16=True, f32=True)
    torch.cuda.synchronize()
    ref = torch.zeros(B, S, H)
    ref[:, 0] = d
    assert torch.equal(sf.cpu().view(B, S, H), ref)
    assert torch.equal(sb.cpu().float().view(B, S, H), bfr(ref))


def test_l2norm_fwd_bwd(ops, dev):
    g = torch.Generator().manual_seed(15)
    x = torch.randn(50, 768, generator=g) * 3
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.normalize(xd, p=2, dim=-1)
    y, inv = ops.l2norm_fwd(x.to(dev))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), ref.detach()) < 1e-6
    dy = torch.randn(50, 768, generator=g)
    (gref,) = torch.autograd.grad(ref, xd, dy.double())
    dx = ops.l2norm_bwd(dy.to(dev), y, inv)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu(), gref) < 1e-5


# ----------------------------------------------------------------------------------------------- K9 loss
def _softce_ref(x, y, labels, row0, scale):
    S = scale * (x @ y.T)
    T = (labels[row0 : row0 + x.shape[0], None] == labels[None, :]).to(S.dtype)
    return -(T * torch.log_softmax(S, dim=1)).sum()


@pytest.mark.parametrize("Nx,N,row0,dup", [(32, 32, 0, False), (64, 64, 0, True), (24, 96, 48, True), (256, 2048, 512, False), (10, 10, 0, True), (7, 21, 14, False),
                                             (1024, 8192, 3072, True)])   # BASELINE configs[4]: rank 3's rows of a global batch of 8192
def test_softce_rows_fwd_bwd(ops, dev, Nx, N, row0, dup):
    g = torch.Generator().manual_seed(N + Nx)
    D = 768
    x = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1)[row0 : row0 + Nx].contiguous()
    y = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1)
    labels = torch.arange(N) // 2 if dup else torch.arange(N)
    scale = 1.0 / 0.07
    xd, yd = x.double().requires_grad_(True), y.double().requires_grad_(True)
    sd = torch.tensor(scale, dtype=torch.float64, requires_grad=True)
    ref = _softce_ref(xd, yd, labels, row0, sd)
    w = 0.37 / N
    gx, gy, gs = torch.autograd.grad(ref * w, (xd, yd, sd))
    ws = ops.softce_workspace(Nx, N, D, dev)
    loss = torch.zeros(1, device=dev)
    sc = torch.tensor([scale], device=dev)
    ops.softce_rows_fwd(x.to(dev), y.to(dev), labels.to(dev), row0, sc, loss, ws)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 2e-4 * abs(ref.item()) + 1e-3
    dx = torch.zeros((Nx, D), device=dev)
    dy0 = torch.randn(N, D, generator=g) * float(gy.abs().mean())  # accumulate semantics (same magnitude: no cancellation in the check)
    dy = dy0.to(dev)
    ds = torch.zeros(1, device=dev)
    ops.softce_rows_bwd(labels.to(dev), Nx, N, D, row0, sc, w, dx, dy, ds, ws)
    torch.cuda.synchronize()
    # backward operands are split hi + lo like the forward's (csrc/loss.hip): fp32-class feature gradients, as the reference's
    # fp32 loss gives (was 6e-3 with a bf16 coefficient matrix against the hi halves only)
    assert rel_err(dx.cpu(), gx) < 1e-4
    assert rel_err(dy.cpu() - dy0, gy) < 1e-4
    assert abs(ds.item() - gs.item()) < 1e-3 * abs(gs.item()) + 1e-6


def test_adamw_matches_torch(ops, dev):
    g = torch.Generator().manual_seed(16)
    n = 10007
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) for _ in range(3)]
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step, gr in enumerate(grads, 1):
        pt.grad = gr.clone()
        opt.step()
        ops.adamw_step(p, (gr * 4).to(dev), m, v, 3e-3, 0.9, 0.999, 1e-8, 0.01, step, grad_scale=0.25)
    torch.cuda.synchronize()
    assert (p.cpu() - pt.detach()).abs().max().item() < 2e-6


def test_fused_adamw_under_one_cycle_lr_on_the_gpu(ops, dev):
    """`clibd_amd.optim.FusedAdamW` (flat bucket + clibd_adamw_step) driven by the reference's OneCycleLR configuration
    (scripts/train_cl.py:222-236: max_lr scaled by the batch, pct_start 0.3, cos, cycle_momentum=False) for ten steps equals
    torch.optim.AdamW under the same scheduler on the CPU."""
    from torch.optim import lr_scheduler

    from clibd_amd.optim import FusedAdamW

    g = torch.Generator().manual_seed(41)
    shapes = [(4, 768), (768, 4), (768, 768), (768,), ()]
    init = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    mine = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    lr = 1e-3 * 2048 / 500
    fo, ro = FusedAdamW(mine, lr=lr), torch.optim.AdamW(ref, lr=lr)
    mk = lambda o: lr_scheduler.OneCycleLR(o, max_lr=4 * lr, total_steps=10, pct_start=0.3, anneal_strategy="cos", cycle_momentum=False)
    fs, rs = mk(fo), mk(ro)
    for _ in range(10):
        fo.zero_grad()
        ro.zero_grad()
        for a, b in zip(mine, ref):
            gr = torch.randn(a.shape, generator=g)
            a.grad.copy_(gr.to(dev))
            b.grad = gr.clone()
        fo.step(); ro.step(); fs.step(); rs.step()
        assert fo.param_groups[0]["lr"] == ro.param_groups[0]["lr"]
    torch.cuda.synchronize()
    for a, b in zip(mine, ref):
        assert (a.detach().cpu() - b.detach()).abs().max().item() < 5e-6


# ----------------------------------------------------------------------------------------------- GEMM, 256x256 8-phase kernel
@pytest.mark.parametrize("M,N,K", [(2048, 4096, 128), (2000, 4096, 256), (4096, 2048, 768), (1500, 6144, 3072), (50432, 768, 768), (2048, 4096, 1536), (3000, 2304, 64 * 7)])
def test_gemm256_exact_integers(ops, dev, M, N, K):
    """Shapes the dispatcher routes to gemm256_bf16_nt_kernel (>=128 tiles of 256x256, K % 128 == 0): exact integer products,
    asymmetric operands, ragged M; repeated launches must be bit-identical (LDS-DMA pipeline races show up as flaky tiles)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    a[:, 0] += (torch.arange(M) % 5).float()
    w[:, 1] += (torch.arange(N) % 7).float()
    ad, wd = a.to(dev, BF16), w.to(dev, BF16)
    ref = (ad.float() @ wd.float().T)  # exact in fp32: |values| small, K <= 3072
    out = torch.empty((M, N), dtype=F32, device=dev)
    for _ in range(3):
        out.fill_(float("nan"))
        ops.gemm_nt(ad, wd, out_f32=out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_gemm256_epilogues(ops, dev):
    g = torch.Generator().manual_seed(77)
    M, N, K = 4096, 2304, 768
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    u, v = torch.randn(M, 8, generator=g), torch.randn(N, 8, generator=g) * 0.1
    res = torch.randn(M, N, generator=g)
    ad, wd = a.to(dev, BF16), w.to(dev, BF16)
    base = ad.float() @ wd.float().T
    # bias + rank-8 -> bf16
    out = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(ad, wd, bias=b.to(dev), rank_u=u.to(dev, BF16), rank_v=v.to(dev, BF16), out_bf16=out)
    ref = base + (bfr(u) @ bfr(v).T).to(dev) + b.to(dev)
    assert rel_err(out.float().cpu(), ref.cpu()) < 4e-3
    # bias + GELU with saved pre-activation
    pre = torch.empty((M, N), dtype=BF16, device=dev)
    act = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(ad, wd, bias=b.to(dev), act=ops.ACT_GELU, out_pre=pre, out_bf16=act)
    pre_ref = bfr((base + b.to(dev)).cpu())
    assert rel_err(pre.float().cpu(), pre_ref) < 3e-3
    assert rel_err(act.float().cpu(), gelu(pre_ref)) < 5e-3
    # residual fp32 + GELU' aux
    aux = torch.randn(M, N, generator=g)
    outf = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm_nt(ad, wd, act=ops.ACT_GELU_GRAD, aux=aux.to(dev, BF16), residual=res.to(dev), out_f32=outf)
    ref = base.cpu() * gelu_grad(bfr(aux)) + res
    assert rel_err(outf.cpu(), ref) < 1e-4


GELUQ_LO, GELUQ_STEP = -0.1328125, 1.265625 / 255.0   # include/clibd_hip.h: CLIBD_ACT_GELU_SAVE_GRAD_U8 / CLIBD_ACT_MUL_AUX_U8


def test_gemm_gelu_save_grad_and_mul_aux(ops, dev):
    g = torch.Generator().manual_seed(91)
    M, N, K = 300, 256, 128
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    pre_ref = bfr(bfr(a) @ bfr(w).T + b)
    dg = torch.empty((M, N), dtype=BF16, device=dev)
    act = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), bias=b.to(dev), act=ops.ACT_GELU_SAVE_GRAD, out_pre=dg, out_bf16=act)
    torch.cuda.synchronize()
    assert rel_err(act.cpu().float(), gelu(pre_ref)) < 5e-3
    assert rel_err(dg.cpu().float(), gelu_grad(pre_ref)) < 5e-3
    aux = bfr(torch.randn(M, N, generator=g))
    out = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), act=ops.ACT_MUL_AUX, aux=aux.to(dev, BF16), out_f32=out)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), (bfr(a) @ bfr(w).T) * aux) < 1e-5


@pytest.mark.parametrize("K", [256, 768])
@pytest.mark.parametrize("kind,bias,lora", [("bf16", False, False), ("bf16", True, True), ("bf16", False, True), ("gelu_save", True, False),
                                            ("gelu_save", False, False), ("mul_aux", False, False), ("res_f32", True, False),
                                            ("res_f32", False, True), ("res_f32_drop", True, False), ("generic_two_outputs", True, False),
                                            ("add_aux", False, False), ("add_aux", False, True), ("add_aux", True, False),
                                            ("gelu_save_u8", True, False), ("gelu_save_u8", False, True), ("mul_aux_u8", False, False),
                                            ("mul_aux_u8", True, True)])
def test_gemm256_epilogue_kinds(ops, dev, kind, bias, lora, K):
    """Every specialised epilogue instantiation of the 256x256 kernel (kind x bias x rank-8 update) on a ragged M (last
    m-tile has 5 live rows), against a torch fp32 reference with the kernel's rounding points; rows past M stay untouched."""
    from oracle import clibd_oracle as O

    M, N = 11 * 256 + 5, 3072
    g = torch.Generator().manual_seed(1000 + K)
    a = (torch.randn(M, K, generator=g)).to(dev, BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev, BF16)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    u = torch.randn(M, 8, generator=g).to(dev, BF16) if lora else None
    v = (torch.randn(N, 8, generator=g) * 0.1).to(dev, BF16) if lora else None
    ref = a.float() @ w.float().T
    if lora:
        ref = ref + u.float() @ v.float().T
    if bias:
        ref = ref + b
    GUARD = 3  # sentinel rows behind the live ones
    def buf(dt):
        t = torch.full((M + GUARD, N), 7.0, dtype=dt, device=dev)
        return t, t[:M]
    kw = dict(bias=b, rank_u=u, rank_v=v)
    checks = []
    if kind == "bf16":
        full, out = buf(BF16)
        ops.gemm_nt(a, w, out_bf16=out, **kw)
        checks.append((full, ref, 4e-3))
    elif kind == "gelu_save":
        fa, act = buf(BF16)
        fg, dg = buf(BF16)
        ops.gemm_nt(a, w, act=ops.ACT_GELU_SAVE_GRAD, out_pre=dg, out_bf16=act, **kw)
        pre = ref.to(BF16).float()
        checks += [(fa, torch.nn.functional.gelu(pre), 5e-3), (fg, gelu_grad(pre.cpu()).to(dev), 5e-3)]
    elif kind == "mul_aux":
        aux = torch.randn(M, N, generator=g).to(dev, BF16)
        full, out = buf(BF16)
        ops.gemm_nt(a, w, act=ops.ACT_MUL_AUX, aux=aux, out_bf16=out, **kw)
        checks.append((full, ref * aux.float(), 4e-3))
    elif kind == "gelu_save_u8":   # gelu' as one byte per element: code = rint((g' - LO) / STEP), exact against the stated code
        fa, act = buf(BF16)
        fg = torch.full((M + GUARD, N), 7, dtype=torch.uint8, device=dev)
        ops.gemm_nt(a, w, act=ops.ACT_GELU_SAVE_GRAD_U8, out_pre=fg[:M], out_bf16=act, **kw)
        pre = ref.to(BF16).float()
        checks.append((fa, torch.nn.functional.gelu(pre), 5e-3))
        torch.cuda.synchronize()
        dec = fg[:M].float().cpu() * GELUQ_STEP + GELUQ_LO
        gref = gelu_grad(pre.cpu())
        # half a step, plus what one bf16 ulp of the pre-activation (accumulation order) moves gelu' by: |gelu''| <= 0.8, ulp(4) = 0.03
        assert float((dec - gref).abs().max()) <= 0.5 * GELUQ_STEP + 8e-3
        assert float(((dec - gref).abs() > 0.5 * GELUQ_STEP + 1e-3).float().mean()) < 2e-2
        assert float((dec - gref).abs().mean()) < 0.3 * GELUQ_STEP
        assert bool((fg[M:] == 7).all()), "rows past M were written"
        small_a = torch.empty((300, 256), dtype=BF16, device=dev)     # the same epilogue through the 128x128 kernel (M < 1024)
        small_g = torch.empty((300, 256), dtype=torch.uint8, device=dev)
        ops.gemm_nt(a[:300], w[:256], act=ops.ACT_GELU_SAVE_GRAD_U8, out_pre=small_g, out_bf16=small_a, bias=b[:256] if bias else None,
                    rank_u=u[:300] if lora else None, rank_v=v[:256] if lora else None)
        torch.cuda.synchronize()
        assert float((small_g.float().cpu() * GELUQ_STEP + GELUQ_LO - gref[:300, :256]).abs().max()) <= 0.5 * GELUQ_STEP + 8e-3
        assert rel_err(small_a.float().cpu(), torch.nn.functional.gelu(pre[:300, :256]).cpu()) < 5e-3
    elif kind == "mul_aux_u8":
        codes = torch.randint(0, 256, (M, N), generator=g, dtype=torch.int64).to(torch.uint8).to(dev)
        auxf = codes.float() * GELUQ_STEP + GELUQ_LO
        full, out = buf(BF16)
        ops.gemm_nt(a, w, act=ops.ACT_MUL_AUX_U8, aux=codes, out_bf16=out, **kw)
        checks.append((full, ref * auxf, 4e-3))
        small = torch.empty((300, 256), dtype=BF16, device=dev)
        ops.gemm_nt(a[:300], w[:256], act=ops.ACT_MUL_AUX_U8, aux=codes[:300, :256], out_bf16=small, bias=b[:256] if bias else None,
                    rank_u=u[:300] if lora else None, rank_v=v[:256] if lora else None)
        assert rel_err(small.float().cpu(), (ref[:300, :256] * auxf[:300, :256]).cpu()) < 4e-3
    elif kind == "add_aux":   # dgrad joining a bf16 residual-gradient stream (CLIBD_ACT_ADD_AUX)
        aux = torch.randn(M, N, generator=g).to(dev, BF16)
        full, out = buf(BF16)
        ops.gemm_nt(a, w, act=ops.ACT_ADD_AUX, aux=aux, out_bf16=out, **kw)
        checks.append((full, ref + aux.float(), 4e-3))
        small = torch.empty((300, 256), dtype=BF16, device=dev)    # the same epilogue through the 128x128 kernel (M < 1024)
        ops.gemm_nt(a[:300], w[:256], act=ops.ACT_ADD_AUX, aux=aux[:300, :256], out_bf16=small, bias=b[:256] if bias else None,
                    rank_u=u[:300] if lora else None, rank_v=v[:256] if lora else None)
        assert rel_err(small.float().cpu(), (ref[:300, :256] + aux[:300, :256].float()).cpu()) < 4e-3
    elif kind in ("res_f32", "res_f32_drop"):
        res = torch.randn(M, N, generator=g).to(dev)
        full, out = buf(F32)
        if kind == "res_f32_drop":
            seed, pdrop = 0x1234567, 0.1
            ops.gemm_nt(a, w, residual=res, out_f32=out, drop=ops.Drop(pdrop, seed), **kw)
            idx = (torch.arange(M, dtype=torch.int64)[:, None] * N + torch.arange(N, dtype=torch.int64)[None, :])
            ref = ref * O.drop_factor(seed, idx, pdrop).to(dev)
        else:
            ops.gemm_nt(a, w, residual=res, out_f32=out, **kw)
        checks.append((full, ref + res, 2e-5))
    else:
        f1, o32 = buf(F32)
        f2, o16 = buf(BF16)
        ops.gemm_nt(a, w, out_f32=o32, out_bf16=o16, **kw)
        checks += [(f1, ref, 2e-5), (f2, ref, 4e-3)]
    torch.cuda.synchronize()
    for full, r, tol in checks:
        assert rel_err(full[:M].float().cpu(), r.cpu()) < tol
        assert bool((full[M:].float() == 7.0).all()), "rows past M were written"


@pytest.mark.parametrize("B,S,nh,nq", [(3, 197, 2, 1), (2, 133, 1, 20), (2, 64, 1, 17)])
def test_attention_query_prefix(ops, dev, B, S, nh, nq):
    """nq < S: only the first nq queries are evaluated; backward with dO given for those rows only."""
    g = torch.Generator().manual_seed(S + nq)
    H = nh * 64
    qkv = bfr(torch.randn(B * S, 3 * H, generator=g))
    qd = qkv.double().requires_grad_(True)
    oref = _attn_ref(qd, B, S, nh, None).view(B, S, H)[:, :nq].reshape(B * nq, H)
    out = torch.empty((B * nq, H), dtype=BF16, device=dev)
    ops.attention_fwd(qkv.to(dev, BF16), B, S, nh, None, out, nq=nq)
    torch.cuda.synchronize()
    assert rel_err(out.cpu().float(), oref.detach()) < 5e-3
    do = bfr(torch.randn(B * nq, H, generator=g))
    (gref,) = torch.autograd.grad(oref, qd, do.double())
    dqkv = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=dev)
    ops.attention_bwd(qkv.to(dev, BF16), do.to(dev, BF16), B, S, nh, None, dqkv, nq=nq)
    torch.cuda.synchronize()
    got = dqkv.cpu().float()
    assert torch.isfinite(got).all()
    assert rel_err(got, gref) < 1.5e-2
    dq = got[:, :H].view(B, S, H)
    assert torch.equal(dq[:, nq:], torch.zeros(B, S - nq, H))


# ----------------------------------------------------------------------------------------------- eval-side "next" rows
@pytest.mark.parametrize("Q,Nk,D,k", [(37, 500, 768, 5), (130, 2048, 128, 5), (5, 9, 64, 3), (64, 21000, 768, 5), (2500, 21000, 768, 5),
                                      (200, 409600, 768, 5), (70, 100, 36, 8), (1, 64, 768, 1)])
def test_topk_inner_product_indices_bit_exact(ops, dev, Q, Nk, D, k):
    """Integer top-k indices must equal the fp32 reference's (north-star: bit-exact); ties -> lower index.  Sizes cover one
    key split and many (few queries: the keys are split over workgroups and merged), a ragged last key tile, the reference's
    BIOSCAN-1M key bank (~21 k) and a BIOSCAN-5M-sized one (409 600 keys), whose [Q, Nk] score matrix is never written."""
    from oracle import clibd_oracle as O

    g = torch.Generator().manual_seed(Q + Nk)
    keys = torch.nn.functional.normalize(torch.randn(Nk, D, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g), dim=1)
    q[0] = keys[Nk // 2]              # an exact match
    sim, idx = ops.topk_ip(q.to(dev), keys.to(dev), k)
    torch.cuda.synchronize()
    q64 = q.double()
    if Nk <= 32768:
        s64 = q64 @ keys.double().T
        ref_idx = torch.argsort(-s64, dim=1, stable=True)[:, :k]
        ref_sim = torch.gather(s64, 1, ref_idx)
    else:   # key chunks: top-k of every chunk (fp64 scores of random data have no ties), merged
        vs, ids = [], []
        for c0 in range(0, Nk, 65536):
            v, i = torch.topk(q64 @ keys[c0 : c0 + 65536].double().T, k, dim=1)
            vs.append(v)
            ids.append(i + c0)
        v, sel = torch.topk(torch.cat(vs, 1), k, dim=1)
        ref_idx, ref_sim = torch.gather(torch.cat(ids, 1), 1, sel), v
    assert torch.equal(idx.cpu(), ref_idx)
    assert torch.allclose(sim.cpu().double(), ref_sim, atol=2e-6)
    if Nk <= 2048:
        osim, oidx = O.topk_inner_product(q, keys, k)
        assert torch.equal(idx.cpu(), oidx)


def test_topk_ties_prefer_lower_index(ops, dev):
    keys = torch.zeros(40, 64)
    keys[:, 0] = 1.0                   # all keys identical -> all scores tie
    q = torch.zeros(3, 64)
    q[:, 0] = 1.0
    _, idx = ops.topk_ip(q.to(dev), keys.to(dev), 4)
    torch.cuda.synchronize()
    assert idx.cpu().tolist() == [[0, 1, 2, 3]] * 3
    # the same across key tiles, key splits and the four per-query lists of a workgroup: 3000 identical keys, two better ones
    keys = torch.zeros(3000, 64)
    keys[:, 0] = 0.5
    keys[1777, 0] = keys[65, 0] = 1.0
    q = torch.zeros(70, 64)
    q[:, 0] = 1.0
    sim, idx = ops.topk_ip(q.to(dev), keys.to(dev), 8)
    torch.cuda.synchronize()
    assert idx.cpu().tolist() == [[65, 1777, 0, 1, 2, 3, 4, 5]] * 70
    assert sim.cpu().tolist() == [[1.0, 1.0] + [0.5] * 6] * 70


@pytest.mark.parametrize("Q,Nk,D,k", [(37, 5000, 768, 5), (130, 21000, 768, 5), (1024, 409600, 768, 5), (300, 70000, 128, 8), (3, 4096, 64, 1)])
def test_topk_prefiltered_search_equals_the_exact_kernel(ops, dev, Q, Nk, D, k):
    """clibd_topk_ip_fast (bf16 approximate scores, exact fp32 re-score of every key inside the rigorous error band of the k-th best)
    returns the exact kernel's indices AND similarities bit for bit — the similarities because the re-score is the same k-ordered
    fmaf chain the fp32 MFMA evaluates.  Random unit vectors: no query overflows its candidate lists."""
    g = torch.Generator().manual_seed(Q + Nk)
    keys = torch.nn.functional.normalize(torch.randn(Nk, D, generator=g), dim=1).to(dev)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g), dim=1)
    q[0] = keys[Nk // 2].cpu()              # an exact match
    q = q.to(dev)
    bank = ops.KeyBank(keys)
    assert abs(float(bank.max_norm) - 1.0) < 1e-4
    assert torch.equal(bank.keys_bf16, keys.to(torch.bfloat16))
    sim, idx, ovf = ops.topk_ip_fast(q, bank, k)
    esim, eidx = ops.topk_ip(q, keys, k)
    torch.cuda.synchronize()
    assert int(ovf.sum()) == 0
    assert torch.equal(idx, eidx)
    assert torch.equal(sim, esim)


def test_topk_prefiltered_search_flags_what_it_cannot_guarantee(ops, dev):
    """Exactness never depends on the data: a bank with 3000 copies of one vector puts thousands of keys inside the error band, the
    candidate lists fill up, the queries are flagged — and `eval.topk_search` re-runs exactly those through the exact kernel
    (ties -> lower index, as faiss).  Unscaled (non-unit) vectors: the band scales with ||q|| max ||key||."""
    from clibd_amd.eval import topk_search

    g = torch.Generator().manual_seed(9)
    Nk, D = 8192, 128
    keys = torch.randn(Nk, D, generator=g)
    keys[2000:5000] = keys[1999]                   # 3001 identical keys
    q = torch.randn(40, D, generator=g)
    q[:8] = keys[1999] + 0.01 * torch.randn(8, D, generator=g)    # these queries' best keys are the duplicates
    keys, q = keys.to(dev), q.to(dev)
    sim, idx, ovf = ops.topk_ip_fast(q, ops.KeyBank(keys), 5)
    esim, eidx = ops.topk_ip(q, keys, 5)
    torch.cuda.synchronize()
    assert ovf[:8].all()                            # lists full inside the band
    ok = ovf == 0
    assert torch.equal(idx[ok], eidx[ok]) and torch.equal(sim[ok], esim[ok])     # everything not flagged is exact
    assert eidx[:8, :5].cpu().tolist() == [[1999, 2000, 2001, 2002, 2003]] * 8
    # the eval wrapper (L2-normalises, pre-filters, repairs the flagged rows): equal to the exact path everywhere
    s1, i1 = topk_search(q, keys, 5)
    s2, i2 = topk_search(q, keys, 5, exact=True)
    torch.cuda.synchronize()
    assert torch.equal(i1, i2) and torch.equal(s1, s2)


def test_kmer_tokenizer_matches_oracle(dev):
    from clibd_amd.eval import tokenize_barcodes
    from oracle import clibd_oracle as O
    import random

    rnd = random.Random(3)
    seqs = ["".join(rnd.choice("ACGT") for _ in range(660)), "ACGTAC", "A" * 1000, "ACGTN" * 100, "", "acgta" * 10 + "ACGTT" * 50]
    got = tokenize_barcodes(seqs, dev)
    torch.cuda.synchronize()
    ref = torch.tensor([O.kmer_tokenize(s) for s in seqs])
    assert got.shape == (len(seqs), 133)
    assert torch.equal(got.cpu(), ref)


def test_make_prediction_contract(dev):
    from clibd_amd.eval import make_prediction, LEVELS

    g = torch.Generator().manual_seed(5)
    keys = torch.randn(30, 64, generator=g)
    labels = [{lv: f"{lv}_{i}" for lv in LEVELS} for i in range(30)]
    pred, sim, idx = make_prediction(keys[[4, 7]] * 3.0, keys, labels, with_similarity=True, with_indices=True, max_k=5, device=dev)
    assert idx[:, 0].tolist() == [4, 7] and abs(sim[0, 0] - 1.0) < 1e-5
    assert pred[0]["species"][0] == "species_4" and len(pred[1]["order"]) == 5


# ----------------------------------------------------------------------------------------------- full fine-tune reductions
@pytest.mark.parametrize("M,H,f32dy,drop", [(300, 768, False, False), (133, 512, True, False), (257, 128, True, True), (64, 1024, False, True)])
def test_layernorm_param_grads(ops, dev, M, H, f32dy, drop):
    from oracle import clibd_oracle as O

    g = torch.Generator().manual_seed(M + H)
    x = torch.randn(M, H, generator=g) * 2 + 0.5
    dy = torch.randn(M, H, generator=g)
    dyd = dy.to(dev) if f32dy else dy.to(dev, BF16)
    mean = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-12)
    stats = torch.cat([mean, rstd], dim=1).to(dev)
    dg = torch.full((H,), 0.25, device=dev)
    db = torch.full((H,), -0.5, device=dev)
    d = ops.Drop(0.1, 99) if drop else None
    ops.layernorm_param_grads(dyd, x.to(dev), stats, dg, db, drop=d)
    dyr = dyd.float().cpu()
    if drop:
        idx = torch.arange(M, dtype=torch.int64)[:, None] * H + torch.arange(H, dtype=torch.int64)[None, :]
        dyr = dyr * O.drop_factor(99, idx, 0.1)
    xh = (x - mean) * rstd
    assert rel_err(dg.cpu(), 0.25 + (dyr * xh).sum(0)) < 2e-5   # accumulates on top of what was there
    assert rel_err(db.cpu(), -0.5 + dyr.sum(0)) < 2e-5


def test_batch_sum_embed_bwd_slice_and_dropout_apply(ops, dev):
    from oracle import clibd_oracle as O

    g = torch.Generator().manual_seed(5)
    B, S, H, V = 70, 13, 128, 50
    x = torch.randn(B, S, H, generator=g)
    out = torch.ones(S, H, device=dev)
    ops.batch_sum(x.to(dev), out)
    assert rel_err(out.cpu(), 1.0 + x.sum(0)) < 1e-5
    ids = torch.randint(0, V, (B * S,), generator=g)
    tt = torch.randint(0, 2, (B * S,), generator=g)
    de = torch.randn(B * S, H, generator=g)
    dword, dtype_t = torch.zeros(V, H, device=dev), torch.zeros(2, H, device=dev)
    ops.bert_embed_bwd(ids.to(dev), tt.to(dev), de.to(dev), dword, dtype_t)
    ref_w = torch.zeros(V, H).index_add_(0, ids, de)
    ref_t = torch.zeros(2, H).index_add_(0, tt, de)
    assert rel_err(dword.cpu(), ref_w) < 1e-5 and rel_err(dtype_t.cpu(), ref_t) < 1e-5
    dtype0 = torch.zeros(2, H, device=dev)
    ops.bert_embed_bwd(ids.to(dev), None, de.to(dev), None, dtype0)   # no token types: everything lands in row 0
    assert rel_err(dtype0[0].cpu(), de.sum(0)) < 1e-5 and float(dtype0[1].abs().max()) == 0.0
    sl = ops.slice_rows_cast_bf16(x.to(dev), 1, S)
    assert torch.equal(sl.cpu(), x[:, 1:].reshape(B * (S - 1), H).to(BF16))
    d = ops.Drop(0.1, 4242)
    y = ops.dropout_apply(de.to(dev), d)
    idx = torch.arange(B * S * H, dtype=torch.int64).view(B * S, H)
    assert torch.allclose(y.cpu(), de * O.drop_factor(4242, idx, 0.1), rtol=1e-6, atol=0)


@pytest.mark.parametrize("M,N,K", [(768, 768, 50432), (3072, 768, 34048), (300, 256, 4096), (768, 3072, 8192 + 128)])
def test_gemm_splitk_workspace_exact(ops, dev, M, N, K):
    """Split-K through the partials workspace (weight-gradient shapes): exact small-integer products, accumulate semantics,
    bit-identical across repeats (no atomics)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-1, 2, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float()
    ad, wd = a.to(dev, BF16), w.to(dev, BF16)
    ref = ad.float() @ wd.float().T   # |sums| <= K < 2^24: exact in fp32 in any order
    out = torch.full((M, N), 3.0, device=dev)
    assert ops.gemm_nt_splitk(ad, wd, out, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(out, ref + 3.0)
    out2 = torch.empty((M, N), device=dev)
    assert ops.gemm_nt_splitk(ad, wd, out2, accumulate=False)
    assert torch.equal(out2, ref)
    assert not ops.gemm_nt_splitk(ad[:, :448], wd[:, :448], out2)   # K < 512: caller falls back to gemm_nt(split_k=)


@pytest.mark.parametrize("M,Na,Nb,ld_extra", [(50432, 768, 768, 0), (34048, 3072, 768, 0), (256, 256, 256, 0), (4096 + 128, 768, 2304, 8),
                                              (640, 512, 256, 0), (50432, 2304, 768, 0),
                                              # M values whose first K-slice plan leaves a tail < 4 K-tiles: 6272 = ViT patch-embed
                                              # rows at per-GPU batch 32, 3200 = text tower at batch 160, and the next few of that family
                                              (6272, 768, 768, 0), (3200, 768, 768, 0), (1664, 768, 768, 0), (12928, 768, 768, 0), (4736, 512, 512, 8)])
def test_gemm_tn_splitk_exact(ops, dev, M, Na, Nb, ld_extra):
    """Rows-contracting GEMM (weight gradient without transposes): out[Na,Nb] (+)= a[M,Na]^T b[M,Nb] — exact small-integer
    products (every k-slot of every fragment matters: position-dependent values), accumulate semantics, operands that are
    column slices of wider buffers, bit-identical across repeats."""
    g = torch.Generator().manual_seed(M + Na + Nb)
    a = torch.randint(-2, 3, (M, Na + ld_extra), generator=g).float()
    b = torch.randint(-2, 3, (M, Nb + ld_extra), generator=g).float()
    ad, bd = a.to(dev, BF16)[:, :Na], b.to(dev, BF16)[:, ld_extra:]
    ref = (a[:, :Na].double().T @ b[:, ld_extra:].double()).float()   # |sums| <= 4 M < 2^24: exact in fp32 in any order
    out = torch.full((Na, Nb), 3.0, device=dev)
    assert ops.gemm_tn_splitk(ad, bd, out, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref + 3.0)
    out2 = torch.empty((Na, Nb), device=dev)
    assert ops.gemm_tn_splitk(ad, bd, out2, accumulate=False)
    assert torch.equal(out2.cpu(), ref)
    out3 = torch.empty((Na, Nb), device=dev)
    cs = torch.full((Na,), 2.0, device=dev)
    assert ops.gemm_tn_splitk(ad, bd, out3, accumulate=False, colsum=cs) and torch.equal(out3, out2)
    assert torch.equal(cs.cpu(), a[:, :Na].sum(0) + 2.0)               # column sums of a (bias gradient), exact on integers
    assert not ops.gemm_tn_splitk(ad[:192], bd[:192], out2)             # M % 128 != 0: the caller takes the transposing path
    assert not ops.gemm_tn_splitk(ad[:, :128], bd, out2[:128])          # Na % 256 != 0


# ----------------------------------------------------------------------------------------------- round 2: full fine-tune fusions
@pytest.mark.parametrize("R,C,ld_extra", [(300, 256, 0), (50432, 768, 0), (133 * 7 + 3, 3072, 8), (64, 64, 0), (1000, 72, 16)])
def test_transpose_with_column_sums(ops, dev, R, C, ld_extra):
    """clibd_transpose_colsum_bf16: bit-exact transpose (zero padded rows) + column sums accumulated in fp32."""
    g = torch.Generator().manual_seed(R + C)
    big = torch.randn(R, C + ld_extra, generator=g).to(BF16)
    x = big.to(dev)[:, :C]
    cs = torch.full((C,), 0.5, device=dev)
    out = ops.transpose_bf16(x, pad_to=128, colsum=cs)
    torch.cuda.synchronize()
    Rp = (R + 127) // 128 * 128
    assert tuple(out.shape) == (C, Rp)
    ref = big[:, :C].T.contiguous()
    assert torch.equal(out[:, :R].cpu(), ref) and float(out[:, R:].float().abs().sum()) == 0.0
    assert rel_err(cs.cpu() - 0.5, big[:, :C].double().sum(0)) < 2e-5
    assert torch.equal(ops.transpose_bf16(x, pad_to=128).cpu(), out.cpu())           # same kernel without the sums


@pytest.mark.parametrize("M,H,f32dy,drop,res", [(300, 768, False, False, True), (5000, 768, True, True, False), (133, 512, True, False, False),
                                                (64, 1024, False, False, True)])
def test_layernorm_bwd_with_fused_param_grads(ops, dev, M, H, f32dy, drop, res):
    """clibd_layernorm_bwd_pg = clibd_layernorm_bwd (bit-identical dx) + d(gamma), d(beta) accumulated in the same pass."""
    g = torch.Generator().manual_seed(M + H)
    x = torch.randn(M, H, generator=g) * 2 + 0.3
    gam, bet = torch.randn(H, generator=g), torch.randn(H, generator=g)
    dy = torch.randn(M, H, generator=g)
    dyd = dy.to(dev) if f32dy else dy.to(dev, BF16)
    dyr = dy if f32dy else bfr(dy)
    st = torch.empty((M, 2), device=dev)
    y = torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), 1e-6, y_bf16=y, stats=st)
    d = ops.Drop(0.1, 777) if drop else None
    dres = torch.randn(M, H, generator=g).to(dev) if res else None
    outs = []
    for fused in (False, True):
        dxf, dxb = torch.empty((M, H), device=dev), torch.empty((M, H), dtype=BF16, device=dev)
        dgam, dbet = torch.full((H,), 0.25, device=dev), torch.full((H,), -0.5, device=dev)
        kw = dict(dgamma=dgam, dbeta=dbet) if fused else {}
        ops.layernorm_bwd(dyd, x.to(dev), st, gam.to(dev), dres=dres, dx_f32=dxf, dx_bf16=dxb, drop=d, **kw)
        torch.cuda.synchronize()
        outs.append((dxf.cpu(), dxb.cpu(), dgam.cpu(), dbet.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    xd = x.double()
    xhat = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-6)
    assert rel_err(outs[1][2] - 0.25, (dyr.double() * xhat).sum(0)) < 2e-4
    assert rel_err(outs[1][3] + 0.5, dyr.double().sum(0)) < 2e-4


# ----------------------------------------------------------------------------------------------- round 3: bf16 residual-gradient stream
@pytest.mark.parametrize("M,H,f32dy,drop", [(1000, 768, False, False), (333, 768, True, True), (70, 512, False, True), (64, 1024, True, False)])
def test_layernorm_bwd_bf16_residual_stream(ops, dev, M, H, f32dy, drop):
    """clibd_layernorm_bwd_res16 = clibd_layernorm_bwd with the residual gradient read and written as bf16: fed the bf16
    image of the fp32 call's residual it must reproduce that call's dx bit for bit (same arithmetic, fp32 inside), as an
    un-dropped residual copy and — under dropout — a masked copy for the dense branch."""
    g = torch.Generator().manual_seed(M + H)
    x = torch.randn(M, H, generator=g) * 2 + 0.3
    gam, bet = torch.randn(H, generator=g), torch.randn(H, generator=g)
    dy = torch.randn(M, H, generator=g)
    dyd = dy.to(dev) if f32dy else dy.to(dev, BF16)
    st = torch.empty((M, 2), device=dev)
    y = torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), 1e-6, y_bf16=y, stats=st)
    d = ops.Drop(0.1, 4242) if drop else None
    dres16 = torch.randn(M, H, generator=g).to(dev, BF16)
    dxf, dxb = torch.empty((M, H), device=dev), torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_bwd(dyd, x.to(dev), st, gam.to(dev), dres=dres16.float(), dx_f32=dxf, dx_bf16=dxb, drop=d)
    res, masked = torch.empty((M, H), dtype=BF16, device=dev), torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_bwd(dyd, x.to(dev), st, gam.to(dev), dres_bf16=dres16, dx_res_bf16=res, dx_bf16=masked, drop=d)
    torch.cuda.synchronize()
    assert torch.equal(res.cpu().float(), bfr(dxf.cpu()))            # the un-dropped gradient of the residual sum
    assert torch.equal(masked.cpu(), dxb.cpu())                      # the dense branch's copy carries the dropout mask
    if drop:
        assert not torch.equal(res.cpu(), masked.cpu())
    only = torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_bwd(dyd, x.to(dev), st, gam.to(dev), dres_bf16=dres16, dx_res_bf16=only)      # one output, no incoming mask
    nores = torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_bwd(dyd, x.to(dev), st, gam.to(dev), dx_bf16=nores, dx_res_bf16=only if False else None, dres_bf16=dres16)
    torch.cuda.synchronize()
    assert torch.equal(only.cpu(), res.cpu())
    if not drop:
        assert torch.equal(nores.cpu(), res.cpu())
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dyd, x.to(dev), st, gam.to(dev), dres=dres16.float(), dres_bf16=dres16, dx_bf16=nores)


@pytest.mark.parametrize("M,H,f32dy,drop", [(777, 768, False, True), (5000, 768, True, False), (64, 512, False, False)])
def test_layernorm_bwd_any_bf16_stream_with_param_grads(ops, dev, M, H, f32dy, drop):
    """clibd_layernorm_bwd_any (round 4: full fine-tune on the bf16 residual-gradient stream): the bf16-stream backward with
    d(gamma), d(beta) accumulated in the same pass and, for the bottom layer, an fp32 dx beside the bf16 copies — every output equal
    to what the separate entry points give (res16 for dx, pg for the parameter gradients), bit for bit."""
    g = torch.Generator().manual_seed(M * 3 + H)
    x = (torch.randn(M, H, generator=g) * 2 + 0.3).to(dev)
    gam, bet = torch.randn(H, generator=g).to(dev), torch.randn(H, generator=g).to(dev)
    dy = torch.randn(M, H, generator=g)
    dyd = dy.to(dev) if f32dy else dy.to(dev, BF16)
    st = torch.empty((M, 2), device=dev)
    y = torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_fwd(x, gam, bet, 1e-6, y_bf16=y, stats=st)
    d = ops.Drop(0.1, 99) if drop else None
    dres16 = torch.randn(M, H, generator=g).to(dev, BF16)
    # references: the two existing entry points
    res_r, msk_r = torch.empty((M, H), dtype=BF16, device=dev), torch.empty((M, H), dtype=BF16, device=dev)
    ops.layernorm_bwd(dyd, x, st, gam, dres_bf16=dres16, dx_res_bf16=res_r, dx_bf16=msk_r, drop=d)
    dxf_r, dxb_r = torch.empty((M, H), device=dev), torch.empty((M, H), dtype=BF16, device=dev)
    dg_r, db_r = torch.zeros((H,), device=dev), torch.zeros((H,), device=dev)
    ops.layernorm_bwd(dyd, x, st, gam, dres=dres16.float(), dx_f32=dxf_r, dx_bf16=dxb_r, drop=d, dgamma=dg_r, dbeta=db_r)
    # the general entry: bf16 residual in, all three outputs, parameter gradients
    res, msk, dxf = torch.empty((M, H), dtype=BF16, device=dev), torch.empty((M, H), dtype=BF16, device=dev), torch.empty((M, H), device=dev)
    dg, db = torch.zeros((H,), device=dev), torch.zeros((H,), device=dev)
    ops.layernorm_bwd(dyd, x, st, gam, dres_bf16=dres16, dx_res_bf16=res, dx_bf16=msk, dx_f32=dxf, drop=d, dgamma=dg, dbeta=db)
    torch.cuda.synchronize()
    assert torch.equal(res, res_r) and torch.equal(msk, msk_r) and torch.equal(dxf, dxf_r)
    # float atomics over blocks: equal up to summation order
    assert rel_err(dg.cpu(), dg_r.cpu().double()) < 1e-5 and rel_err(db.cpu(), db_r.cpu().double()) < 1e-5
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dyd, x, st, gam, dres_bf16=dres16, dx_res_bf16=res, dgamma=dg)       # d(gamma) and d(beta) come together


def test_ln_fold_epilogues(ops, dev):
    """Round 5, the algebraic LayerNorm -> Linear fold (clibd_gemm_epilogue.row_sums / row_stats): the PRODUCER epilogue (projection:
    bias + fp32 residual -> fp32 out, its bf16 copy and per-128-column-slice row sums), clibd_rowsum_finalize (mean, rstd), clibd_ln_fold_weights
    and the CONSUMER epilogue (rstd (acc - mean s) + b' -> GELU + GELU') against fp64 statements of the same operands, and against the
    unfolded pair LayerNorm -> Linear it replaces."""
    M, H, FF = 50 * 256 + 5, 768, 3072
    g = torch.Generator().manual_seed(321)
    a = torch.randn(M, H, generator=g).to(dev, BF16)
    wo = (torch.randn(H, H, generator=g) * 0.03).to(dev, BF16)
    bo = (torch.randn(H, generator=g) * 0.1).to(dev)
    res = (torch.randn(M, H, generator=g) * 1.5 + 0.2).to(dev)
    x1, x1b = torch.empty((M, H), device=dev), torch.empty((M, H), dtype=BF16, device=dev)
    sums = torch.full((H // 128, M, 2), float("nan"), device=dev)
    ops.gemm_nt(a, wo, bias=bo, residual=res, out_f32=x1, out_bf16=x1b, row_sums=sums)
    ref = torch.empty((M, H), device=dev)
    ops.gemm_nt(a, wo, bias=bo, residual=res, out_f32=ref)
    torch.cuda.synchronize()
    assert torch.equal(x1, ref)                                           # the plain kind's value, bit for bit
    assert torch.equal(x1b.view(torch.int16), x1.to(BF16).view(torch.int16))   # its bf16 rounding
    s_ref = torch.stack([x1.double().view(M, H // 128, 128).sum(-1).T, (x1.double() ** 2).view(M, H // 128, 128).sum(-1).T], dim=-1)
    assert rel_err(sums.cpu().double(), s_ref.cpu()) < 2e-6
    stats = torch.empty((M, 2), device=dev)
    ops.rowsum_finalize(sums, 1e-6, stats)
    mean64 = x1.double().mean(-1)
    rstd64 = (x1.double().var(-1, unbiased=False) + 1e-6).rsqrt()
    torch.cuda.synchronize()
    assert float((stats[:, 0].double() - mean64).abs().max()) < 1e-5 and rel_err(stats[:, 1].cpu().double(), rstd64.cpu()) < 1e-5
    # consumer
    gamma, beta = (1.0 + 0.2 * torch.randn(H, generator=g)).to(dev), (0.1 * torch.randn(H, generator=g)).to(dev)
    w1 = (torch.randn(FF, H, generator=g) * 0.03).to(dev)
    b1 = (torch.randn(FF, generator=g) * 0.1).to(dev)
    wg, s_n, bp = ops.ln_fold_weights(w1, gamma, beta, b1)
    torch.cuda.synchronize()
    assert torch.equal(wg.view(torch.int16), (w1 * gamma).to(BF16).view(torch.int16))
    assert rel_err(s_n.cpu().double(), wg.double().sum(-1).cpu()) < 1e-6 and rel_err(bp.cpu().double(), (b1.double() + w1.double() @ beta.double()).cpu()) < 1e-6
    act, dg = torch.empty((M, FF), dtype=BF16, device=dev), torch.empty((M, FF), dtype=BF16, device=dev)
    ops.gemm_nt(x1b, wg, bias=bp, act=ops.ACT_GELU_SAVE_GRAD, out_pre=dg, out_bf16=act, row_stats=stats, col_sum_w=s_n)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 512), torch.arange(M - 300, M)])
    h64 = stats[rows, 1:2].double() * (x1b[rows].double() @ wg.double().T - stats[rows, 0:1].double() * s_n.double()) + bp.double()
    pre = h64.float().to(BF16).float()
    assert rel_err(act[rows].float().cpu(), torch.nn.functional.gelu(pre).cpu()) < 5e-3
    assert rel_err(dg[rows].float().cpu(), gelu_grad(pre.cpu())) < 5e-3
    # against the pair it replaces: LayerNorm (fp32 statistics, bf16 output) -> Linear -> GELU
    xn = torch.nn.functional.layer_norm(x1[rows], (H,), gamma, beta, 1e-6).to(BF16)
    h_std = (xn.double() @ w1.to(BF16).double().T + b1.double()).float()
    assert rel_err(act[rows].float().cpu(), torch.nn.functional.gelu(h_std.to(BF16).float()).cpu()) < 1.2e-2
    with pytest.raises(Exception):   # the fold epilogues exist in the 256x256 kernel only
        ops.gemm_nt(a[:300], wo, bias=bo, residual=res[:300], out_f32=x1[:300], out_bf16=x1b[:300], row_sums=sums[:, :300].contiguous())


# ----------------------------------------------------------------------------------------------- gelu' in twelve bits (round 6)
def decode_e4m7(buf, N):
    """host statement of the e4m7 layout (include/clibd_hip.h CLIBD_ACT_*_E12): uint8 [M, 3N/2] -> fp32 [M, N]"""
    M = buf.shape[0]
    b = buf.cpu().view(M, N // 8, 12).to(torch.int64)
    w = [b[..., 4 * i] | (b[..., 4 * i + 1] << 8) | (b[..., 4 * i + 2] << 16) | (b[..., 4 * i + 3] << 24) for i in range(3)]
    c = [w[0] & 0xfff, (w[0] >> 12) & 0xfff, (w[0] >> 24) | ((w[1] & 0xf) << 8), (w[1] >> 4) & 0xfff, (w[1] >> 16) & 0xfff,
         (w[1] >> 28) | ((w[2] & 0xff) << 4), (w[2] >> 8) & 0xfff, w[2] >> 20]
    c = torch.stack(c, dim=-1).view(M, N)
    u = c & 0x7ff
    t = torch.where(u > 0, u + (112 << 7), torch.zeros_like(u))
    bits = (((c & 0x800) << 4) | t) << 16
    return bits.to(torch.int32).view(torch.float32)


@pytest.mark.parametrize("M,N,K", [(4096, 3072, 768), (300, 256, 128), (2048, 3072, 768)])   # 256x256 kernel (192 tiles) | 128x128 kernel | the class-row block's shape
def test_gelu_grad_e4m7_is_the_bf16_value(ops, dev, M, N, K):
    """CLIBD_ACT_GELU_SAVE_GRAD_E12 keeps bf16(gelu') in twelve bits — sign, 4-bit exponent, 7 mantissa bits: decoded, it must EQUAL the bf16 form's
    gelu' wherever |gelu'| >= 2^-14 and be zero below; the GELU output itself is the bf16 form's bit for bit.  CLIBD_ACT_MUL_AUX_E12 then multiplies by
    exactly that decoded operand: bit-identical to CLIBD_ACT_MUL_AUX fed the decoded values as bf16.  Pre-activations are spread over +-9 so that
    the flushed tail (x < -4.55), the zero crossing (x = -0.75) and the saturated side (gelu' -> 1) are all in the sample."""
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.35).to(BF16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.35).to(BF16).to(dev)
    bias = (torch.randn(N, generator=g) * 2.0).to(dev)
    h16, a16 = torch.empty((M, N), dtype=BF16, device=dev), torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a, w, bias=bias, act=ops.ACT_GELU_SAVE_GRAD, out_pre=h16, out_bf16=a16)
    h12, a12 = torch.empty((M, 3 * N // 2), dtype=torch.uint8, device=dev), torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a, w, bias=bias, act=ops.ACT_GELU_SAVE_GRAD_E12, out_pre=h12, out_bf16=a12)
    torch.cuda.synchronize()
    assert torch.equal(a12, a16)
    want, got = h16.float().cpu(), decode_e4m7(h12, N)
    big = want.abs() >= 2.0 ** -14
    assert big.float().mean() > 0.5 and (~big).sum() > 0                 # both regimes are sampled
    assert torch.equal(got[big], want[big])                             # bit for bit
    assert float(got[~big].abs().max()) == 0.0                          # flushed: |error| < 2^-14 = 6.1e-5
    assert float(want.max()) > 1.1 and float(want.min()) < -0.12         # the whole range of gelu' occurs
    # the fc2 dgrad's operand: the decoded values, bit for bit
    dy = (torch.randn(M, 256, generator=g) * 0.1).to(BF16).to(dev)
    wt = (torch.randn(N, 256, generator=g) * 0.1).to(BF16).to(dev)
    o12, o16 = torch.empty((M, N), dtype=BF16, device=dev), torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(dy, wt, act=ops.ACT_MUL_AUX_E12, aux=h12, out_bf16=o12)
    ops.gemm_nt(dy, wt, act=ops.ACT_MUL_AUX, aux=got.to(BF16).to(dev), out_bf16=o16)
    torch.cuda.synchronize()
    assert torch.equal(o12, o16)


def test_gelu_grad_e4m7_covers_every_bf16_value_of_gelu_grad(ops, dev):
    """Exhaustive over the domain: every bf16 pre-activation x (65 536 bit patterns, non-finite ones excluded) through the kernel's own gelu' —
    a GEMM with a one-hot operand reproduces x exactly — encoded and decoded: equal to the bf16 form's gelu'(x) whenever that is >= 2^-14 in
    magnitude, zero otherwise; and NO value of gelu' reaches the code's upper end (|gelu'| < 2)."""
    bits = torch.arange(65536, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    x = x[torch.isfinite(x) & (x.abs() < 1e30)]
    n = x.numel() // 256 * 256
    x = x[:n].view(-1, 256)                                                # [M, 256] pre-activations
    M, N = x.shape[0], 256
    a = x.clone()                                                          # [M, 256] as the A operand (K = 256)
    w = torch.eye(N)                                                       # out[m, n] = x[m, n] exactly: one non-zero product per output
    h16, o16 = torch.empty((M, N), dtype=BF16, device=dev), torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a.to(BF16).to(dev), w.to(BF16).to(dev), act=ops.ACT_GELU_SAVE_GRAD, out_pre=h16, out_bf16=o16)
    h12, o12 = torch.empty((M, 3 * N // 2), dtype=torch.uint8, device=dev), torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a.to(BF16).to(dev), w.to(BF16).to(dev), act=ops.ACT_GELU_SAVE_GRAD_E12, out_pre=h12, out_bf16=o12)
    torch.cuda.synchronize()
    want, got = h16.float().cpu(), decode_e4m7(h12, N)
    ok = torch.isfinite(want)
    big = ok & (want.abs() >= 2.0 ** -14)
    assert float(want[ok].abs().max()) < 2.0
    assert torch.equal(got[big], want[big]) and float(got[ok & ~big].abs().max()) == 0.0
    assert int(big.sum()) > 20000


def test_key_bank_cache_follows_the_callers_tensor(dev):
    """eval.topk_search keeps ONE prepared bank per device key tensor — and only as long as the caller's tensor lives, never for a
    temporary device copy of host keys, and not at all for a tensor whose version counter cannot be read."""
    import gc

    import numpy as np

    from clibd_amd import eval as ev

    g = torch.Generator().manual_seed(3)
    keys = torch.nn.functional.normalize(torch.randn(4096, 128, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(16, 128, generator=g), dim=1).to(dev)
    kd = keys.to(dev)
    ev.clear_key_bank_cache()
    s1, i1 = ev.topk_search(q, kd, 5)
    bank = ev._bank_cache["entry"][2]
    s2, i2 = ev.topk_search(q, kd, 5)
    assert ev._bank_cache["entry"][2] is bank and torch.equal(i1, i2) and torch.equal(s1, s2)      # hit
    kd.mul_(1.0)                                                                                  # in-place update: new version, new bank
    ev.topk_search(q, kd, 5)
    assert ev._bank_cache["entry"][2] is not bank
    del kd
    gc.collect()
    assert "entry" not in ev._bank_cache                                                           # the bank went with the caller's tensor
    labels = [{lv: str(j) for lv in ev.LEVELS} for j in range(4096)]
    ev.make_prediction(q, keys.numpy(), labels)                                                    # host keys: nothing is pinned
    assert "entry" not in ev._bank_cache
    with torch.inference_mode():
        ki = keys.to(dev)
    s3, i3 = ev.topk_search(q, ki, 5)                                                              # no version counter: works, uncached
    assert torch.equal(i3, i1) and "entry" not in ev._bank_cache
    exact = ev.topk_search(q, keys.to(dev), 5, exact=True)
    assert torch.equal(exact[1], i1)


# ----------------------------------------------------------------------------------------------- stream-K tail of the 256x256 GEMM (round 6)
@pytest.mark.parametrize("M,N,K,parts,tail", [(50432, 768, 3072, 3, 79), (50432, 768, 2304, 3, 79), (43776, 768, 3072, 4, 1), (50400, 768, 3072, 3, 79),
                                              (22528, 768, 1536, 4, 8)])
def test_gemm_stream_k_tail_is_exact_and_deterministic(ops, dev, M, N, K, parts, tail):
    """clibd_gemm_bf16_nt_ws: the last, partial tile round cut into K-slices over the idle CUs (per-rank shapes of the 8-GPU configuration: 591 tiles =
    2 full rounds + 79 tiles -> 3 slices each).  Integer operands make every product and partial sum exact in fp32, so every epilogue kind that has the
    form must equal the plain launch BIT FOR BIT (and the fp64 statement where there is one): plain bf16 output, + adapters' rank update, + bf16 aux,
    bias + fp32 residual, bias -> dropout -> + fp32 residual; ragged M; the flags are back at zero; two runs give the same bits."""
    from clibd_amd import _lib

    lib = _lib.load()
    if torch.cuda.get_device_properties(dev).multi_processor_count != 256:
        pytest.skip("the plan under test is the 256-CU one")
    need = int(lib.clibd_gemm_tail_workspace_bytes(M, N, K))
    assert need == 1024 + tail * (parts - 1) * 256 * 256 * 4, need
    g = torch.Generator().manual_seed(M + K)
    a = torch.randint(-2, 3, (M, K), generator=g).to(BF16).to(dev)
    w = torch.randint(-2, 3, (N, K), generator=g).to(BF16).to(dev)
    bias = torch.randint(-4, 5, (N,), generator=g).float().to(dev)
    res = torch.randint(-8, 9, (M, N), generator=g).float().to(dev)
    aux = torch.randint(-4, 5, (M, N), generator=g).to(BF16).to(dev)
    u = torch.zeros((M, 8)); u[:, :4] = torch.randint(-1, 2, (M, 4), generator=g).float(); u[:, 4:] = torch.randint(-1, 2, (M, 4), generator=g).float()
    v = torch.randint(-1, 2, (N, 8), generator=g).float()
    u, v = u.to(BF16).to(dev), v.to(BF16).to(dev)
    drop = ops.Drop(0.1, 4242)

    def run_all():
        o = {}
        o["bf16"] = torch.empty((M, N), dtype=BF16, device=dev); ops.gemm_nt(a, w, out_bf16=o["bf16"])
        o["lora"] = torch.empty((M, N), dtype=BF16, device=dev); ops.gemm_nt(a, w, rank_u=u, rank_v=v, out_bf16=o["lora"])
        o["add"] = torch.empty((M, N), dtype=BF16, device=dev); ops.gemm_nt(a, w, act=ops.ACT_ADD_AUX, aux=aux, out_bf16=o["add"])
        o["res"] = torch.empty((M, N), dtype=F32, device=dev); ops.gemm_nt(a, w, bias=bias, residual=res, out_f32=o["res"])
        o["drop"] = torch.empty((M, N), dtype=F32, device=dev); ops.gemm_nt(a, w, bias=bias, residual=res, out_f32=o["drop"], drop=drop)
        torch.cuda.synchronize()
        return o

    was = ops._STREAMK        # (opt-in: measured slower in-step, profiles/r06_exp_gemm_stream_k_tail.log)
    ops._STREAMK = True
    try:
        sk1 = run_all()
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        assert key in ops._tail_ws and int(ops._tail_ws[key][:1024].to(torch.int32).sum()) == 0      # every flag is back at zero
        sk2 = run_all()
        ops._STREAMK = False
        plain = run_all()
    finally:
        ops._STREAMK = was
    for k in sk1:
        assert torch.equal(sk1[k], sk2[k]), k
        assert torch.equal(sk1[k], plain[k]), k
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 700, M)])       # the tail tiles are the LAST tiles of the XCD-ordered walk: check both ends in fp64
    acc = a[rows.to(dev)].double() @ w.double().T
    assert torch.equal(sk1["bf16"][rows.to(dev)].double(), acc.float().to(BF16).double())
    assert torch.equal(sk1["res"][rows.to(dev)].double(), acc + bias.double() + res[rows.to(dev)].double())
    assert torch.equal(sk1["lora"][rows.to(dev)].double(), (acc + u[rows.to(dev)].double() @ v.double().T).float().to(BF16).double())


def test_gemm256_f32_and_nograd_gelu_kinds(ops, dev):
    """Round 6: two epilogue forms that ran on the generic kind of the 256x256 kernel have their own instantiations — [bias ->] fp32 output (patch
    embedding, the MLM head's dgrad) and bias -> GELU -> bf16 (fc1 of the no-grad forward).  Exact on integers / equal to the 128x128 kernel's GELU."""
    M, N, K = 4096, 3072, 768            # 192 tiles: the 256x256 kernel
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    ref = a.double() @ w.double().T
    out = torch.empty((M, N), dtype=F32, device=dev)
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), bias=bias.to(dev), out_f32=out)
    assert torch.equal(out.cpu().double(), ref + bias.double())
    ops.gemm_nt(a.to(dev, BF16), w.to(dev, BF16), out_f32=out)
    assert torch.equal(out.cpu().double(), ref)
    # GELU form against torch on the exact pre-activation, and against the 128x128 kernel (same arithmetic on a shape the 256x256 kernel declines)
    a2, w2 = (a * 0.0625).to(dev, BF16), (w * 0.0625).to(dev, BF16)
    b2 = (bias * 0.25).to(dev)
    o16 = torch.empty((M, N), dtype=BF16, device=dev)
    ops.gemm_nt(a2, w2, bias=b2, act=ops.ACT_GELU, out_bf16=o16)
    pre = (ref * 0.0625 * 0.0625 + bias.double() * 0.25).float()
    assert rel_err(o16.float().cpu(), gelu(pre)) < 3e-3
    small = torch.empty((512, N), dtype=BF16, device=dev)
    ops.gemm_nt(a2[:512], w2, bias=b2, act=ops.ACT_GELU, out_bf16=small)       # M < 1024: the 128x128 kernel
    assert torch.equal(small, o16[:512])
