"""CPU study (oracle only, no GPU): which fp8-forward POLICY keeps the contrastive gradient faithful to the bf16 path's?

VERDICT r3 item 1 asks for cosine(fp8 gradient, bf16 gradient) >= 0.98 on trained weights (0.45-0.92 with the round-3 mode) and
names the hardware's per-32-element E8M0 block scales (MXFP8) as the means.  Before a kernel is written, the oracle answers what
each candidate buys on the full-size towers (ViT-B/16 + BERT-base, batch 16, adapters + heads trained for a few bf16 steps on the
batch so that the embeddings are spread):

  tensor    : round 3 — one power-of-two scale per layer and site for activations, one scale per weight row
  mx32      : MXFP8 — e4m3 payload with an E8M0 (power-of-two) scale per 32 consecutive K elements, both operands (scale per the OCP
              MX rule); mx32_noclip: the smallest power-of-two scale that does not saturate the block maximum
  cls_bf16  : `tensor`, but the class-token row of every ViT block (the only row the image embedding reads) is recomputed on
              bf16 operands in all four GEMMs (1/197 of the rows); BERT unchanged (its head averages 133 tokens)
  mx32+cls  : both
  lastK_bf16: `tensor` with the last K ViT blocks entirely bf16
  dna_only  : the image tower entirely bf16, only the DNA tower (whose head averages its 133 tokens) on fp8 operands
  tensor_cal: per-tensor activation scales as a calibration pass sets them (2^floor(log2(448 / (2 amax))) of the tensor itself)
  STUDY_OUTLIER_GAIN=g (environment): four channels of every LayerNorm weight multiplied by g before anything else (outlier channels)
  act_only / w_only : diagnostics (no such MFMA exists): only the activations / only the weights quantised to e4m3

Reports max |embedding - bf16 embedding| per tower, |loss difference| and the cosine of the full trainable gradient against the
bf16 mode's, on the training batch and on a fresh batch.

    python tools/fp8_policy_study.py [train_steps=8] [batch=16] > profiles/r04_exp_fp8_policy_study.log
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import clibd_oracle as O

torch.set_num_threads(8)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16

POLICY = dict(gran="tensor", cls_bf16=False, in_image=False, bf16_from_block=None, block_index=None, image_bf16=False, image_sites=None,
              fp8_until_block=None, dgrad=None, dgrad_margin=1.0)


def fp8_grad(x, fmt, margin=1.0):
    """A gradient tensor as an 8-bit dgrad GEMM would take it: one power-of-two scale for the tensor, 2^floor(log2(fmax / (2 amax)))
    of the tensor itself (`margin` > 1 stands for a DELAYED scale that is off by that factor: the tensor is `margin` x larger than the
    scale was made for), payload e5m2 (fmax 57344) or e4m3 (448).  Returns the de-quantised values."""
    fmax, dt = (57344.0, torch.float8_e5m2) if fmt == "e5m2" else (448.0, torch.float8_e4m3fn)
    if POLICY.get("dgrad_rows"):   # one scale per ROW (token) of the gradient: it factors out of the dgrad's contraction like a tensor scale does
        amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        s = torch.exp2(torch.floor(torch.log2(fmax / (2.0 * amax * margin))))
        return (x * s).clamp(-fmax, fmax).to(dt).to(x.dtype) / s
    amax = x.abs().amax().clamp_min(1e-30)
    s = float(2.0 ** torch.floor(torch.log2(fmax / (2.0 * amax * margin))))
    return (x * s).clamp(-fmax, fmax).to(dt).to(x.dtype) / s


def vit_site(weight):
    """Which of a ViT block's four GEMMs a weight belongs to, from its shape ([out, in]; ViT-B/16: H = 768)."""
    o, i = weight.shape
    if o == 3 * i:
        return "qkv"
    if o == 4 * i:
        return "fc1"
    if i == 4 * o:
        return "fc2"
    return "proj"


def e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype)


MX_NOCLIP = False


def mx_quant(x):
    """MXFP8 along the last axis: per 32 elements a power-of-two (E8M0) scale, payload e4m3(x / scale).  Scale as the OCP MX
    specification derives it, 2^(floor(log2 amax) - 8) (e4m3's largest binade is 2^8; a block maximum with a mantissa above 1.75
    then saturates at 448), or — MX_NOCLIP — the smallest power of two that keeps the block maximum representable,
    2^ceil(log2(amax / 448)).  Returns the de-quantised values (what the block-scaled MFMA multiplies)."""
    shp = x.shape
    xb = x.reshape(-1, shp[-1] // 32, 32)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    safe = torch.where(amax > 0, amax, torch.ones_like(amax))
    e = torch.ceil(torch.log2(safe / 448.0)) if MX_NOCLIP else torch.floor(torch.log2(safe)) - 8.0
    s = torch.exp2(e)
    return (e4m3(xb / s) * s).reshape(shp)


class PolicyLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, sa):
        ctx.save_for_backward(weight)
        ctx.in_image = POLICY["in_image"]
        ctx.blk = POLICY["block_index"]
        xf, wf = x.detach().float(), weight.detach().float()
        rb = lambda t: t.to(torch.bfloat16).to(t.dtype)
        blk, frm = POLICY["block_index"], POLICY["bf16_from_block"]
        if POLICY["in_image"] and POLICY["image_bf16"]:
            return F.linear(rb(xf), rb(wf))
        if (not POLICY["in_image"]) and POLICY.get("dna_sites") is not None and vit_site(weight) not in POLICY["dna_sites"]:
            return F.linear(rb(xf), rb(wf))   # round 5: fp8 forward on SOME of the DNA tower's GEMM sites (q / k / v / attention output are [H,H]: "proj")
        if POLICY.get("dna_fwd_bf16"):    # diagnostic: the 8-bit dgrad alone, forward on bf16 operands
            return F.linear(rb(xf), rb(wf))
        if POLICY["in_image"] and frm is not None and blk is not None and blk >= frm:
            return F.linear(rb(xf), rb(wf))
        # round 5: fp8 only on SOME of the ViT's GEMM sites / only in its first blocks, the rest of the tower bf16
        if POLICY["in_image"] and POLICY["image_sites"] is not None and vit_site(weight) not in POLICY["image_sites"]:
            return F.linear(rb(xf), rb(wf))
        if POLICY["in_image"] and POLICY["fp8_until_block"] is not None and blk is not None and blk >= POLICY["fp8_until_block"]:
            return F.linear(rb(xf), rb(wf))
        global MX_NOCLIP
        MX_NOCLIP = POLICY["gran"] == "mx32_noclip"
        if POLICY["gran"] == "act_only":       # diagnostic: e4m3 activations against bf16 weights (no such MFMA exists)
            y = F.linear(e4m3(xf * sa), rb(wf)) * (1.0 / sa)
        elif POLICY["gran"] == "w_only":       # diagnostic: bf16 activations against e4m3 weights
            w8, sn = O.quantize_rows_e4m3(wf)
            y = F.linear(rb(xf), w8) * (1.0 / sn).view(-1)
        elif POLICY["gran"] in ("mx32", "mx32_noclip"):
            y = F.linear(mx_quant(xf), mx_quant(wf))
        else:
            w8, sn = O.quantize_rows_e4m3(wf)
            if POLICY["gran"] == "tensor_cal":      # the scale a calibration pass would set for this very tensor: 2^floor(log2(448 / (2 amax)))
                sa = float(2.0 ** torch.floor(torch.log2(448.0 / (2.0 * xf.abs().amax().clamp_min(1e-30)))))
            y = F.linear(e4m3(xf * sa), w8) * (1.0 / (sn * sa)).view(-1)
        if POLICY["cls_bf16"] and POLICY["in_image"] and xf.dim() == 3:
            y = y.clone()
            y[:, 0, :] = F.linear(rb(xf[:, 0, :]), rb(wf))
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        rb = lambda t: t.to(torch.bfloat16).to(t.dtype)
        # round 5: 8-bit dgrad on the towers whose forward is fp8 (never the ViT under image_bf16): the gradient as e5m2 / e4m3 with one
        # scale per tensor, the weight as e4m3 with one scale per INPUT channel (the dgrad's output column, so the scale factors out)
        img_ok = POLICY.get("dgrad_image") and (POLICY.get("dgrad_image_until") is None or (ctx.blk is not None and ctx.blk < POLICY["dgrad_image_until"]))
        if POLICY["dgrad"] is not None and (img_ok or not ctx.in_image):
            wt = weight.detach().float().t().contiguous()          # [in, out]: rows = the dgrad GEMM's output channels
            w8, sn = O.quantize_rows_e4m3(wt)
            fmt, margin = POLICY["dgrad"], POLICY["dgrad_margin"]
            if fmt == "mixed":   # rows that a LayerNorm backward produced (exact row maximum known): e4m3; d(fc1 out), whose scale is a bound: e5m2
                is_dh = weight.shape[0] == 4 * weight.shape[1]
                fmt, margin = ("e5m2", margin) if is_dh else ("e4m3", 1.0)
            return fp8_grad(dy.float(), fmt, margin) @ (w8 / sn).t(), None, None
        return rb(dy) @ rb(weight.detach()), None, None


O._Fp8Linear = PolicyLinear


OUTLIERS = float(os.environ.get("STUDY_OUTLIER_GAIN", "0"))   # > 0: four channels of every LayerNorm weight are multiplied by this


def build():
    torch.manual_seed(11)
    om = O.build_image_dna_model()
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
        if OUTLIERS > 0:    # outlier channels as trained checkpoints have them: the activations feeding every GEMM get a 30-50x dynamic range
            g = torch.Generator().manual_seed(77)
            for n, p in om.named_parameters():
                if p.dim() == 1 and ("norm" in n.lower()) and n.endswith("weight"):
                    idx = torch.randperm(p.numel(), generator=g)[:4]
                    p[idx] *= OUTLIERS
    # block index for the lastK policy
    for i, blk in enumerate(om.image_encoder.base_image_encoder.blocks):
        orig = blk.forward

        def fwd(x, _orig=orig, _i=i):
            POLICY["block_index"] = _i
            try:
                return _orig(x)
            finally:
                POLICY["block_index"] = None
        blk.forward = fwd
    return om


def batch(seed):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(B, 3, 224, 224, generator=g)
    dna = torch.cat([torch.zeros(B, 1, dtype=torch.long), torch.randint(3, 1027, (B, 132), generator=g)], dim=1)
    return image, dna, torch.arange(B)


def evaluate(om, image, dna, labels):
    POLICY["in_image"] = True
    img = F.normalize(om.image_encoder(image).float(), p=2, dim=-1)
    POLICY["in_image"] = False
    d = F.normalize(om.dna_encoder(dna).float(), p=2, dim=-1)
    loss = O.contrastive_loss([img, d, None], labels, om.logit_scale.exp())
    ps = [p for p in om.parameters() if p.requires_grad]
    gs = torch.autograd.grad(loss, ps, allow_unused=True)
    g = torch.cat([(torch.zeros_like(p) if g_ is None else g_).flatten() for p, g_ in zip(ps, gs)])
    return img.detach(), d.detach(), float(loss.detach()), g


def main():
    om = build()
    tr, fr = batch(3), batch(4)
    opt = torch.optim.AdamW([p for p in om.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-2)
    t0 = time.time()
    losses = []
    with O.precision("bf16"):
        for _ in range(STEPS):
            losses.append(float(O.train_step(om, opt, *tr[:2], tr[2])))
    print(f"trained {STEPS} bf16 steps on a fixed batch of {B}: loss {losses[0]:.3f} -> {losses[-1]:.3f}  ({time.time() - t0:.0f} s)", flush=True)
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    policies = [("tensor", dict(gran="tensor", cls_bf16=False, bf16_from_block=None)),
                ("dna_only", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True)),
                ("dna_only_mx32", dict(gran="mx32", cls_bf16=False, bf16_from_block=None, image_bf16=True)),
                ("mx32", dict(gran="mx32", cls_bf16=False, bf16_from_block=None)),
                ("mx32_noclip", dict(gran="mx32_noclip", cls_bf16=False, bf16_from_block=None)),
                ("tensor_cal", dict(gran="tensor_cal", cls_bf16=False, bf16_from_block=None)),
                ("dna_only_cal", dict(gran="tensor_cal", cls_bf16=False, bf16_from_block=None, image_bf16=True)),
                ("dna_only_mx32_noclip", dict(gran="mx32_noclip", cls_bf16=False, bf16_from_block=None, image_bf16=True)),
                ("act_only", dict(gran="act_only", cls_bf16=False, bf16_from_block=None)),
                ("w_only", dict(gran="w_only", cls_bf16=False, bf16_from_block=None)),
                ("cls_bf16", dict(gran="tensor", cls_bf16=True, bf16_from_block=None)),
                ("mx32+cls_bf16", dict(gran="mx32", cls_bf16=True, bf16_from_block=None)),
                ("last2_bf16", dict(gran="tensor", cls_bf16=False, bf16_from_block=10)),
                ("last6_bf16", dict(gran="tensor", cls_bf16=False, bf16_from_block=6)),
                # round 5: site selections inside the ViT (the DNA tower on fp8 throughout)
                ("vit_mlp", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_sites=("fc1", "fc2"))),
                ("vit_fc1", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_sites=("fc1",))),
                ("vit_fc2", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_sites=("fc2",))),
                ("vit_qkv_proj", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_sites=("qkv", "proj"))),
                ("vit_mlp_cal", dict(gran="tensor_cal", cls_bf16=False, bf16_from_block=None, image_sites=("fc1", "fc2"))),
                ("vit_mlp_first6", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_sites=("fc1", "fc2"), fp8_until_block=6)),
                ("vit_mlp_first3", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_sites=("fc1", "fc2"), fp8_until_block=3)),
                ("vit_fc2_first6", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_sites=("fc2",), fp8_until_block=6)),
                # round 5: 8-bit dgrad on the DNA tower (the "pooled" selection), forward as dna_only
                ("dna_only+dgrad_e5m2", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e5m2")),
                ("dna_only+dgrad_e4m3", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3")),
                ("dna_only+dgrad_e5m2_x16", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e5m2", dgrad_margin=16.0)),
                ("dna_only+dgrad_e4m3_x16", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3", dgrad_margin=16.0)),
                ("dna_only+dgrad_rows_e5m2_x32", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e5m2", dgrad_margin=32.0, dgrad_rows=True)),
                ("dna_only+dgrad_both_e5m2", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e5m2", dgrad_image=True)),
                ("dna_only+dgrad_both_rows_e5m2_x32", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e5m2", dgrad_margin=32.0, dgrad_rows=True, dgrad_image=True)),
                ("dna_only+dgrad_both_e4m3", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3", dgrad_image=True)),
                ("dna_only+dgrad_both_rows_e4m3", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3", dgrad_rows=True, dgrad_image=True)),
                ("dna_only+dgrad_both_rows_e4m3_x32", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3", dgrad_margin=32.0, dgrad_rows=True, dgrad_image=True)),
                ("dna_only+dgrad_both_rows_mixed_x32", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="mixed", dgrad_margin=32.0, dgrad_rows=True, dgrad_image=True)),
                ("dna_only+dgrad_rows_mixed_x32", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="mixed", dgrad_margin=32.0, dgrad_rows=True)),
                ("dna_only+dgrad_rows_e4m3+vit_first9", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3", dgrad_rows=True, dgrad_image=True, dgrad_image_until=9)),
                ("dna_only+dgrad_rows_e4m3+vit_first6", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3", dgrad_rows=True, dgrad_image=True, dgrad_image_until=6)),
                ("dna_only+dgrad_rows_e4m3", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e4m3", dgrad_rows=True)),
                ("dna_mlp", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dna_sites=("fc1", "fc2"))),
                ("dna_mlp+dgrad_both_rows_e4m3", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dna_sites=("fc1", "fc2"), dgrad="e4m3", dgrad_rows=True, dgrad_image=True)),
                ("dna_fc2", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dna_sites=("fc2",))),
                ("dna_attn", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dna_sites=("proj",))),
                ("dna_bf16fwd+dgrad_e5m2", dict(gran="tensor", cls_bf16=False, bf16_from_block=None, image_bf16=True, dgrad="e5m2", dna_fwd_bf16=True))]
    for name, (image, dna, labels) in (("train batch", tr), ("fresh batch", fr)):
        with O.precision("bf16"):
            i16, d16, l16, g16 = evaluate(om, image, dna, labels)
        spread = float((i16 @ i16.T).fill_diagonal_(0).sum() / (B * (B - 1)))
        print(f"== {name}: bf16 loss {l16:.4f}, mean mutual cosine of image embeddings {spread:.3f}", flush=True)
        for pname, pol in policies:
            if only is not None and pname not in only:
                continue
            POLICY.update(image_bf16=False, image_sites=None, fp8_until_block=None, dgrad=None, dgrad_margin=1.0, dna_fwd_bf16=False, dgrad_rows=False, dgrad_image=False, dgrad_image_until=None, dna_sites=None)
            POLICY.update(pol)
            with O.precision("fp8"):
                i8, d8, l8, g8 = evaluate(om, image, dna, labels)
            cosv = float(g8.double() @ g16.double() / (g8.double().norm() * g16.double().norm()))
            print(f"   {pname:16s} image emb err {float((i8 - i16).abs().max()):.2e}  dna emb err {float((d8 - d16).abs().max()):.2e}  "
                  f"|dloss| {abs(l8 - l16):.2e}  cosine(grad fp8, grad bf16) {cosv:.4f}", flush=True)


if __name__ == "__main__":
    main()
