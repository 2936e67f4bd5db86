"""LayerNorm forward timing at the bench shapes, with and without the fused LoRA down-projection."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16; F32 = torch.float32
for M in (403456, 272384, 50432):
    H = 768
    x = torch.randn(M, H, device=dev)
    g, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
    y = torch.empty(M, H, device=dev, dtype=BF16); yf = torch.empty(M, H, device=dev)
    st = torch.empty(M, 2, device=dev)
    a = torch.randn(8, H, device=dev).to(BF16); t = torch.empty(M, 8, device=dev, dtype=BF16)
    t0 = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=y, stats=st))
    t1 = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=y, stats=st, lora_a=a, t_out=t))
    t2 = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=y, y_f32=yf, stats=st, lora_a=a, t_out=t))
    print(f"M={M}: plain {t0*1e3:6.1f} us ({(M*H*6)/t0/1e9:5.2f} TB/s)   +lora {t1*1e3:6.1f} us   +lora +f32 out {t2*1e3:6.1f} us", flush=True)
