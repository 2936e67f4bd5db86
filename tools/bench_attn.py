"""Attention forward / backward timing at the bench shapes (ViT 197 tokens, DNA 133 tokens with dropout): the plain forward, the
training forward that also saves lse / o_lo, the two-phase backward and the single-pass backward.
    python tools/bench_attn.py [batch, default 256 and 2048]"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
batches = [int(a) for a in sys.argv[1:]] or [256, 2048]
for B in batches:
    for S, p in ((197, 0.0), (133, 0.1)):
        nh = 12
        H = nh * 64
        qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(BF16)
        out, o_lo = torch.empty(B * S, H, device=dev, dtype=BF16), torch.empty(B * S, H, device=dev, dtype=BF16)
        lse = torch.empty(B * nh * S, device=dev)
        do = torch.randn(B * S, H, device=dev).to(BF16)
        dqkv = torch.empty_like(qkv)
        drop = ops.Drop(p, 5) if p > 0 else None
        f = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, None, out, drop=drop))
        fs = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, None, out, drop=drop, lse=lse, o_lo=o_lo))
        b2 = timeit(lambda: ops.attention_bwd(qkv, do, B, S, nh, None, dqkv, drop=drop))
        b1 = timeit(lambda: ops.attention_bwd_sp(qkv, do, out, o_lo, lse, B, S, nh, dqkv, drop=drop))
        print(f"B={B} S={S} p={p}: fwd {f*1e3:7.1f} us  fwd+save {fs*1e3:7.1f} us   bwd two-phase {b2*1e3:7.1f} us  bwd single-pass {b1*1e3:7.1f} us", flush=True)
