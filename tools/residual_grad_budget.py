"""Gradient budget of the bf16 residual-gradient stream (clibd_amd.engine.NUMERICS_CHOICES['residual_grad']).

Full-size towers (ViT-B/16, BERT-base), batch 8, LoRA B matrices non-zero, one random cotangent on the tower output: the
trainable gradients (adapters + head) of the HIP tower with the residual gradient carried in fp32 and in bf16, against the
CPU oracle in fp32 (the reference's default arithmetic) and in its bf16 mode (the kernels' rounding points, fp32 residual
gradients like the reference's autocast).  Prints relative L2 error and cosine over all trainable tensors and the worst
per-tensor relative error, per tower and mode.

    python tools/residual_grad_budget.py                                  # CLIBD_RESIDUAL_GRAD: fp32 | bf16
    python tools/residual_grad_budget.py CLIBD_GELU_GRAD bf16 u8          # any two-valued knob: reference mode first
"""
import os
import sys

import torch

sys.path.insert(0, ".")
from oracle import clibd_oracle as O  # noqa: E402
from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, create_vit, load_pre_trained_bioscan_bert  # noqa: E402

dev = torch.device("cuda:0")
B = 8


def grads(module, out, cot):
    ps = {n: p for n, p in module.named_parameters() if p.requires_grad}
    gs = torch.autograd.grad((out * cot).sum(), list(ps.values()), allow_unused=True)
    return {n: (torch.zeros_like(p) if g is None else g).detach().double().cpu() for (n, p), g in zip(ps.items(), gs)}


def compare(tag, got, ref):
    names = sorted(ref)
    a = torch.cat([got[n].flatten() for n in names])
    e = torch.cat([ref[n].flatten() for n in names])
    rel = float((a - e).norm() / e.norm())
    cosv = float(a @ e / (a.norm() * e.norm()))
    worst = max(float((got[n] - ref[n]).norm() / (ref[n].norm() + 1e-30)) for n in names if float(ref[n].norm()) > 0)
    print(f"    {tag:44s} rel {rel:.4e}  cos {cosv:.6f}  worst tensor rel {worst:.4e}")
    return rel


def main():
    knob, mode_a, mode_b = (sys.argv[1:4] if len(sys.argv) >= 4 else ("CLIBD_RESIDUAL_GRAD", "fp32", "bf16"))
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(4)
    om = O.build_image_dna_model()
    with torch.no_grad():
        for n, p in om.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    image = torch.rand(B, 3, 224, 224, generator=g)
    dna = torch.cat([torch.zeros(B, 1, dtype=torch.long), torch.randint(3, 1027, (B, 132), generator=g)], dim=1)
    cot = torch.randn(B, 768, generator=g)
    for name, oenc, build, inp in (("image tower (ViT-B/16, pre-LN)", om.image_encoder, lambda: CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768), image),
                                   ("DNA tower (BERT-base, post-LN)", om.dna_encoder, lambda: CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), dna)):
        print(name)
        ref32 = grads(oenc, oenc(inp), cot)
        with O.precision("bf16"):
            ref16 = grads(oenc, oenc(inp), cot)
        henc = build()
        henc.load_state_dict(oenc.state_dict(), strict=True)
        henc = henc.to(dev).eval()
        res = {}
        for mode in (mode_a, mode_b):
            henc.tower().stack.set_numerics(**{knob.replace("CLIBD_", "").lower(): mode})   # CLIBD_RESIDUAL_GRAD -> residual_grad
            out = henc(inp.to(dev))
            res[mode] = grads(henc, out, cot.to(dev))
            torch.cuda.synchronize()
        compare("oracle bf16 mode  vs oracle fp32", ref16, ref32)
        for mode in (mode_a, mode_b):
            compare(f"HIP, {knob}={mode}  vs oracle fp32", res[mode], ref32)
            compare(f"HIP, {knob}={mode}  vs oracle bf16", res[mode], ref16)
        compare(f"HIP {mode_b}  vs HIP {mode_a}", res[mode_b], res[mode_a])


if __name__ == "__main__":
    main()
