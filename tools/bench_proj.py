"""Projection-shape GEMMs (N = K = 768, fp32 residual in/out): v1 128x128 kernel (CLIBD_GEMM_KERNEL=1) vs the 256x256 kernel."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
for M in (50432, 34048):
    for (N, K) in ((768, 768), (768, 3072), (768, 2304)):
        a = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); outf = torch.empty(M, N, device=dev); outb = torch.empty(M, N, device=dev, dtype=BF16)
        for name, kw in (("res f32->f32", dict(bias=bias, residual=res, out_f32=outf)), ("bf16 out", dict(out_bf16=outb)), ("f32 out", dict(out_f32=outf))):
            ms = timeit(lambda: ops.gemm_nt(a, w, **kw))
            print(f"M={M} N={N} K={K:5d} {name:14s}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
