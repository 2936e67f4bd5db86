"""LayerNorm backward with the e4m3 row output (8-bit dgrad) against the plain kernel, at the bench shapes.
pre-LN form (ViT): dy bf16, x fp32, dres bf16 in; dx bf16 [+ e4m3 rows + row factors] out.  post-LN form (BERT, dropout on the dense copy):
dy bf16, x fp32 in; residual copy bf16 + masked bf16 copy (plain) or residual copy bf16 + masked e4m3 rows (8-bit dgrad) out."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16; F32 = torch.float32
H = 768
for M in (403456, 272384, 50432):
    x = torch.randn(M, H, device=dev); dy = torch.randn(M, H, device=dev).to(BF16)
    g = torch.randn(H, device=dev); st = torch.stack([x.mean(1), 1.0 / x.std(1)], dim=1).contiguous()
    dres16 = torch.randn(M, H, device=dev).to(BF16); dx16 = torch.empty(M, H, device=dev, dtype=BF16); dxr16 = torch.empty(M, H, device=dev, dtype=BF16)
    d8 = torch.empty(M, H, device=dev, dtype=torch.uint8).view(ops.FP8); rd = torch.empty(M, device=dev)
    drop = ops.Drop(0.1, 77)
    t0 = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dres_bf16=dres16, dx_bf16=dx16))
    t1 = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dres_bf16=dres16, dx_bf16=dx16, dx_fp8=d8, row_dequant=rd))
    t2 = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dx_res_bf16=dxr16, dx_bf16=dx16, drop=drop))
    t3 = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dx_res_bf16=dxr16, drop=drop, dx_fp8=d8, row_dequant=rd))
    print(f"M={M}: pre-LN plain {t0*1e3:6.1f} us ({M*H*10/t0/1e9:5.2f} TB/s)  + e4m3 rows {t1*1e3:6.1f} us ({M*H*11/t1/1e9:5.2f} TB/s)   "
          f"post-LN plain {t2*1e3:6.1f} us ({M*H*10/t2/1e9:5.2f} TB/s)  e4m3 rows instead of the masked copy {t3*1e3:6.1f} us ({M*H*9/t3/1e9:5.2f} TB/s)", flush=True)
