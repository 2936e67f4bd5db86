"""Attention forward / backward time against the number of query rows evaluated (nq): the intercept is staging + launch, the
slope the per-query-tile cost.  python tools/exp_attn_parts.py [B] [dropout p]"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
drop = ops.Drop(float(sys.argv[2]), 77) if len(sys.argv) > 2 and float(sys.argv[2]) > 0 else None
nh = 12; H = nh * 64
for S in (197, 133):
    qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(BF16)
    dqkv = torch.empty_like(qkv)
    row = []
    for nq in (1, 32, 64, 128, S):
        out = torch.empty(B * nq, H, device=dev, dtype=BF16)
        do = torch.randn(B * nq, H, device=dev).to(BF16)
        f = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, None, out, nq=nq, drop=drop))
        b = timeit(lambda: ops.attention_bwd(qkv, do, B, S, nh, None, dqkv, nq=nq, drop=drop))
        row.append(f"nq={nq}: fwd {f*1e3:6.1f} bwd {b*1e3:6.1f}")
    print(f"B={B} S={S} | " + " | ".join(row), flush=True)
