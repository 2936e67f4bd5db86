"""Attention forward / backward timing at the bench shapes (ViT 197 tokens, DNA 133 tokens; 256 x 12 heads)."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
for S in (197, 133):
    B, nh = 256, 12
    H = nh * 64
    qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(BF16)
    out = torch.empty(B * S, H, device=dev, dtype=BF16)
    do = torch.randn(B * S, H, device=dev).to(BF16)
    dqkv = torch.empty_like(qkv)
    f = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, None, out))
    b = timeit(lambda: ops.attention_bwd(qkv, do, B, S, nh, None, dqkv))
    print(f"S={S}: fwd {f*1e3:7.1f} us   bwd {b*1e3:7.1f} us", flush=True)
