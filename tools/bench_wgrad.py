"""Weight-gradient GEMM dW = dy^T x: transposes + NT split-K against the in-place TN kernel."""
import sys; sys.path.insert(0, ".")
import torch
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
for M, N, K in ((256 * 197, 768, 768), (256 * 197, 2304, 768), (256 * 197, 3072, 768), (256 * 197, 768, 3072), (256 * 133, 3072, 768), (2048 * 197, 3072, 768)):
    dy = (torch.randn(M, N, device=dev) * 0.1).to(BF16); x = torch.randn(M, K, device=dev).to(BF16)
    gw = torch.zeros(N, K, device=dev)
    def old():
        dyT = ops.transpose_bf16(dy, pad_to=128); xT = ops.transpose_bf16(x, pad_to=128)
        ops.gemm_nt_splitk(dyT, xT, gw, accumulate=True)
    def new():
        ops.gemm_tn_splitk(dy, x, gw, accumulate=True)
    t_old = timeit(old); t_new = timeit(new)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: transposes+NT {t_old*1e3:8.1f} us   TN {t_new*1e3:8.1f} us ({fl/t_new/1e9:.0f} TF)   x{t_old/t_new:.2f}", flush=True)
