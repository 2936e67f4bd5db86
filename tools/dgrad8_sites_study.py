"""MI355X: which of the ViT's 8-bit dgrad GEMMs cost the gradient its fidelity?  Full-size pair, adapters + heads trained in bf16 on 32 fixed
pairs (8, then 32 more steps, as tests/test_fp8_gpu.py::test_fp8_gradients_on_spread_embeddings), then the cosine of the full trainable gradient
against the bf16 dgrad's for: the DNA tower's 8-bit dgrad alone; + the ViT's MLP pair only; + the ViT's projection only; + both (dgrad8 on all).
    python tools/dgrad8_sites_study.py > gpurun_out/<tag>/dgrad8_sites.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clibd_amd.data import synthetic_batch
from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, ClipLoss, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert
from clibd_amd.train import Trainer

dev = torch.device("cuda:0")
torch.manual_seed(11)
model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                   CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
with torch.no_grad():
    for n, p in model.named_parameters():
        if "linear_b_" in n or ".w_b." in n:
            p.normal_(0, 0.02)
model = model.to(dev).eval()
B = 32
batch, fresh = synthetic_batch(B, dev, seed=3, rank=0, with_text=False), synthetic_batch(B, dev, seed=4, rank=0, with_text=False)
tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True)
crit = ClipLoss(local_loss=False, gather_with_grad=True, rank=0, world_size=1, criterion=torch.nn.CrossEntropyLoss())
vit = model.image_encoder.tower().stack


def grads(bt):
    hi, hd, _, scale, _ = model(bt["image"], bt["dna"], None)
    loss = crit(hi, hd, None, bt["labels"], scale)
    ps = {n: p for n, p in model.named_parameters() if p.requires_grad}
    gs = torch.autograd.grad(loss, list(ps.values()), allow_unused=True)
    model.join_streams()
    torch.cuda.synchronize()
    return torch.cat([(torch.zeros_like(p) if g is None else g).detach().double().flatten().cpu() for p, g in zip(ps.values(), gs)])


def cosv(a, b):
    return float(a @ b / (a.norm() * b.norm()))


CASES = [("dna only", "pooled", ("mlp", "proj")), ("+ ViT MLP pair", "all", ("mlp",)), ("+ ViT projection", "all", ("proj",)), ("+ ViT both", "all", ("mlp", "proj"))]
for stage, nsteps in (("8 steps", 8), ("40 steps", 32)):
    for _ in range(nsteps):
        tr.step(batch["image"], batch["dna"], None, batch["labels"])
    sinks = [(tw, tw.grad_sink) for tw in (model.image_encoder.tower(), model.dna_encoder.tower())]
    for tw, _ in sinks:
        tw.grad_sink = None
    for name, bt in (("train", batch), ("fresh", fresh)):
        model.enable_fp8_dgrad(enabled=False)
        g16 = grads(bt)
        row = []
        for label, sel, sites in CASES:
            vit.dgrad8_sites = sites
            model.enable_fp8_dgrad(towers=sel)
            row.append(f"{label} {cosv(grads(bt), g16):.4f}")
        vit.dgrad8_sites = ("mlp", "proj")
        model.enable_fp8_dgrad(enabled=False)
        print(f"after {stage}, {name} batch: " + "   ".join(row), flush=True)
    for tw, sk in sinks:
        tw.grad_sink = sk
