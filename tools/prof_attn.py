"""Attention kernels alone for a rocprofv3 pass (ViT shape by default): python tools/prof_attn.py [S] [B] [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 197
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
nh = 12; H = nh * 64
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(BF16)
out = torch.empty(B * S, H, device=dev, dtype=BF16)
do = torch.randn(B * S, H, device=dev).to(BF16)
dqkv = torch.empty_like(qkv)
for _ in range(reps):
    ops.attention_fwd(qkv, B, S, nh, None, out)
    ops.attention_bwd(qkv, do, B, S, nh, None, dqkv)
torch.cuda.synchronize()
print("done")
