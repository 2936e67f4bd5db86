"""Loss trajectories of the same training run in bf16, fp8-forward "pooled" and fp8-forward "all" (full-size ViT-B/16 + BERT-base, LoRA r=4,
the towers in train mode with identical dropout seeds, 256 fixed synthetic pairs cycled in batches of 64, AdamW through Trainer.step,
fp8 scales re-calibrated every 10 steps): how far the fp8 modes drift from the bf16 run they approximate.
Round 5: a mode is <forward>[+dgrad8|+dgrad8p] with forward in bf16 | pooled | pooled_ffn | pooled_mlp | all; "+dgrad8" switches the 8-bit dgrad on for every tower (numerics dgrad = "fp8"), "+dgrad8p" for the mean-pooled towers only.
    python tools/fp8_trajectory.py [steps=80] [modes=pooled,all] > gpurun_out/<tag>/fp8_trajectory.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clibd_amd.data import synthetic_batch
from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert
from clibd_amd.train import Trainer

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["pooled", "all"]
data = synthetic_batch(256, dev, seed=7, rank=0, with_text=False)


def run(mode):
    torch.manual_seed(123)
    model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                       CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "linear_b_" in n or ".w_b." in n:
                p.normal_(0, 0.02)
    model = model.to(dev).train()
    fwd = mode.split("+")[0]
    if fwd != "bf16":
        model.enable_fp8_forward(towers=fwd)
    if mode.endswith("+dgrad8"):
        model.enable_fp8_dgrad(towers="all")
    elif mode.endswith("+dgrad8p"):     # the 8-bit dgrad on the mean-pooled towers only
        model.enable_fp8_dgrad(towers="pooled")
    tr = Trainer(model, lr=1e-3, world_size=1, rank=0, all_gather=True, fp8_recalibrate_every=10 if fwd != "bf16" else 0)
    out = []
    torch.manual_seed(999)                      # the same dropout seeds in every mode
    for s in range(steps):
        sl = slice(64 * (s % 4), 64 * (s % 4) + 64)
        out.append(float(tr.step(data["image"][sl], data["dna"][sl], None, data["labels"][sl])))
    return out


res = {m: run(m) for m in ["bf16"] + modes}
print(f"# {steps} steps, batch 64 of 256 fixed pairs, lr 1e-3; loss per step (every 5th) and the fp8 runs' distance from the bf16 run")
for s in range(0, steps, 5):
    b = res["bf16"][s]
    print(f"step {s:3d}  bf16 {b:8.4f}   " + "   ".join(f"{m} {res[m][s]:8.4f} ({res[m][s] - b:+.4f})" for m in modes))
for m in modes:
    d = [abs(a - b) for a, b in zip(res[m], res["bf16"])]
    print(f"{m:14s}: mean |loss - bf16 loss| {sum(d) / len(d):.4f}, max {max(d):.4f}, final loss {res[m][-1]:.4f} (bf16 {res['bf16'][-1]:.4f}), mean of the last 10 {sum(res[m][-10:]) / 10:.4f} (bf16 {sum(res['bf16'][-10:]) / 10:.4f})")
