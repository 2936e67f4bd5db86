"""CPU study (oracle only): what the algebraic LN2 -> fc1 fold would do to the fc1 pre-activation.

    standard (the kernels, the reference under autocast):  h = bf16(LN(x)) . bf16(W)^T + b        LN in fp32 on the fp32 residual stream
    fold:   h = rstd . ( bf16(x) . bf16(gamma o W)^T  -  mu . s ) + b',   s = bf16(gamma o W) 1,  b' = b + W beta
            (x rounded to bf16 BEFORE the mean is taken out: the rounding error of a row is 2^-9 |x|, not 2^-9 |x - mu|)

Both against the fp64 value, per ViT block, on the residual stream the full-size ViT-B/16 oracle produces for a batch of 8 images:
relative L2 error of h over all rows, the worst row, and the |mu| / sigma of the rows that bounds the cancellation.  A second pass
repeats it with outlier channels in the residual stream (four channels of the stream offset by `gain` sigma, what trained checkpoints
with massive activations look like; the random-init stream has |mu| << sigma and hides the problem).

    python tools/ln_fold_study.py > profiles/r05_exp_ln_fold_study.log
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import clibd_oracle as O

torch.set_num_threads(8)
rb = lambda t: t.to(torch.bfloat16).to(torch.float64)


def study(x, gamma, beta, W, b, eps, tag):
    x64, g64, be64, W64, b64 = x.double(), gamma.double(), beta.double(), W.double(), b.double()
    mu = x64.mean(-1, keepdim=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdim=True)
    rstd = (var + eps).rsqrt()
    xn = (x64 - mu) * rstd * g64 + be64
    truth = xn @ W64.T + b64
    std_form = rb(xn.float()) @ rb(W).T + b64
    gW = rb((g64 * W64).float())                       # bf16 image of gamma o W, built once per weight version
    s = gW.sum(-1)
    # row sums taken in fp32 from the fp32 accumulators of the projection epilogue (before rounding x): mu, rstd are the standard ones
    fold = rstd * (rb(x) @ gW.T - mu * s) + (b64 + W64 @ be64)
    rel = lambda a: float((a - truth).norm() / truth.norm())
    worst = lambda a: float(((a - truth).norm(dim=-1) / truth.norm(dim=-1)).max())
    ratio = (mu.abs() / var.sqrt()).flatten()
    amp = (x64.abs().amax(-1, keepdim=True) / var.sqrt()).flatten()
    print(f"  {tag}: |mu|/sigma median {float(ratio.median()):.3f} max {float(ratio.max()):.2f}; max|x|/sigma median {float(amp.median()):.1f} max {float(amp.max()):.1f};  "
          f"rel L2 error of h: standard {rel(std_form):.2e} (worst row {worst(std_form):.2e})   fold {rel(fold):.2e} (worst row {worst(fold):.2e})   ratio {rel(fold) / rel(std_form):.2f}", flush=True)


def main():
    torch.manual_seed(5)
    om = O.build_image_dna_model()
    vit = om.image_encoder.base_image_encoder
    g = torch.Generator().manual_seed(9)
    image = torch.rand(8, 3, 224, 224, generator=g)
    for gain in (0.0, 20.0, 100.0):
        print(f"== residual stream of the random-init ViT-B/16 oracle, batch 8" + (f", four channels offset by {gain:g} sigma (massive activations)" if gain else ""), flush=True)
        with torch.no_grad():
            x = vit.patch_embed(image)
            x = torch.cat([vit.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + vit.pos_embed
            if gain:
                idx = torch.randperm(x.shape[-1], generator=torch.Generator().manual_seed(77))[:4]
                x[..., idx] += gain * x.std()
            for i, blk in enumerate(vit.blocks):
                x1 = x + blk.attn(blk.norm1(x))
                if i in (0, 5, 11):
                    study(x1.reshape(-1, x1.shape[-1]), blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.norm2.eps, f"block {i:2d}")
                x = x1 + blk.mlp(blk.norm2(x1))


if __name__ == "__main__":
    main()
