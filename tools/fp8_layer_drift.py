"""Per-layer drift of the fp8-forward mode from the bf16 path (ViT-B/16 tower, random init): relative L2 distance of qkv, the
post-attention stream x1 and the block input, layer by layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clibd_amd import ops
from clibd_amd.data import synthetic_batch
from clibd_amd.model import CLIBDImageEncoder, create_vit

dev = torch.device("cuda:0")
torch.manual_seed(11)
m = CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768).to(dev).eval()
tw = m.tower()
B = 8
img = synthetic_batch(B, dev, seed=5, rank=0, with_text=False)["image"]

def run():
    out, state = tw._forward((img,), True)
    torch.cuda.synchronize()
    return out.float().cpu(), state

def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm()).item()

o16, s16 = run()
tw.stack.enable_fp8()
o8, s8 = run()
print("out rel", rel(o8, o16))
for i, (r8, r16) in enumerate(zip(s8["saved"], s16["saved"])):
    print(f"layer {i:2d}: x_in {rel(r8['x_in'], r16['x_in']):.4f}  xn {rel(r8['xn'], r16['xn']):.4f} qkv {rel(r8['qkv'], r16['qkv']):.4f}  x1 {rel(r8['x1'], r16['x1']):.4f}"
          f"  |x_in| rms {r16['x_in'].float().pow(2).mean().sqrt().item():.3f} mean-over-tokens share {(r16['x_in'].float().view(B, 197, -1).mean(1).norm() ** 2 * 197 / r16['x_in'].float().norm() ** 2).item():.3f}")
