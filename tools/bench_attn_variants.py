"""Round 6: the attention instantiations that spilled, at the shapes that select them, forward and backward.
    python tools/bench_attn_variants.py          (CLIBD_HIP_LIB selects a variant library)"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
B, nh = 256, 12
H = nh * 64
for S, p, masked, what in ((150, 0.0, False, "bwd<10,.,nomask,nodrop,4,160>"), (150, 0.1, False, "bwd<10,.,nomask,drop,4,160>"),
                           (250, 0.1, False, "bwd<16,.,nomask,drop>, fwd persistent<16,nomask,drop>"), (220, 0.1, True, "fwd persistent<14,mask,drop>"),
                           (250, 0.1, True, "fwd persistent<16,mask,drop>")):
    qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(BF16)
    out = torch.empty(B * S, H, device=dev, dtype=BF16)
    do = torch.randn(B * S, H, device=dev).to(BF16)
    dqkv = torch.empty_like(qkv)
    mask = None
    if masked:
        mask = torch.ones((B, S), dtype=torch.int32, device=dev)
        mask[:, S - 7:] = 0
    drop = ops.Drop(p, 5) if p > 0 else None
    f = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, mask, out, drop=drop))
    b = timeit(lambda: ops.attention_bwd(qkv, do, B, S, nh, mask, dqkv, drop=drop))
    print(f"B={B} S={S} p={p} mask={masked}: fwd {f*1e3:7.1f} us   bwd {b*1e3:7.1f} us   [{what}]", flush=True)
