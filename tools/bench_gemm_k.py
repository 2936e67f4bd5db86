"""GEMM time vs K at fixed M x N (per-tile fixed overhead = intercept) and epilogue cost, for kernel tuning."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
M, N = 50432, 2304
for K in (128, 256, 512, 768, 1536, 3072):
    a = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    out = torch.empty(M, N, device=dev, dtype=BF16)
    ms = timeit(lambda: ops.gemm_nt(a, w, out_bf16=out))
    print(f"plain  M={M} N={N} K={K:5d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
K, N = 768, 3072
a = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
out = torch.empty(M, N, device=dev, dtype=BF16); pre = torch.empty_like(out); bias = torch.randn(N, device=dev)
aux = torch.randn(M, N, device=dev).to(BF16); res = torch.randn(M, N, device=dev); outf = torch.empty(M, N, device=dev)
for name, kw in (("plain", dict(out_bf16=out)), ("bias", dict(bias=bias, out_bf16=out)), ("bias+pre (2 stores)", dict(bias=bias, out_pre=pre, out_bf16=out)),
                 ("bias+gelu+pre", dict(bias=bias, act=ops.ACT_GELU, out_pre=pre, out_bf16=out)), ("gelu only", dict(act=ops.ACT_GELU, out_bf16=out)),
                 ("gelu_grad(aux)", dict(act=ops.ACT_GELU_GRAD, aux=aux, out_bf16=out)),
                 ("bias+gelu_save_grad", dict(bias=bias, act=ops.ACT_GELU_SAVE_GRAD, out_pre=pre, out_bf16=out)),
                 ("mul_aux", dict(act=ops.ACT_MUL_AUX, aux=aux, out_bf16=out)), ("res f32 -> f32", dict(residual=res, out_f32=outf))):
    ms = timeit(lambda: ops.gemm_nt(a, w, **kw))
    print(f"fc1-shape {name:22s}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
