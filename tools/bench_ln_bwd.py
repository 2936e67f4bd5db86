"""LayerNorm backward timing at the bench shapes: the pre-LN form with the bf16 residual-gradient stream (dy bf16, x fp32, dres bf16 in;
dx bf16 and the bf16 residual gradient out: 12 B per element) and the post-LN fp32 form."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16; F32 = torch.float32
for M in (403456, 272384, 50432):
    H = 768
    x = torch.randn(M, H, device=dev); dy = torch.randn(M, H, device=dev).to(BF16)
    g = torch.randn(H, device=dev); st = torch.stack([x.mean(1), 1.0 / x.std(1)], dim=1).contiguous()
    dres16 = torch.randn(M, H, device=dev).to(BF16); dx16 = torch.empty(M, H, device=dev, dtype=BF16); dxr16 = torch.empty(M, H, device=dev, dtype=BF16)
    dres32 = torch.randn(M, H, device=dev); dx32 = torch.empty(M, H, device=dev)
    t0 = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dres_bf16=dres16, dx_res_bf16=dxr16, dx_bf16=dx16))
    t1 = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dres=dres32, dx_f32=dx32, dx_bf16=dx16))
    print(f"M={M}: bf16 residual stream {t0*1e3:6.1f} us ({(M*H*12)/t0/1e9:5.2f} TB/s)   fp32 stream {t1*1e3:6.1f} us ({(M*H*16)/t1/1e9:5.2f} TB/s)", flush=True)
