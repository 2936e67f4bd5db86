import sys; sys.path.insert(0, ".")
import torch
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0")
for M, C in ((256 * 197, 768), (256 * 197, 3072), (2048 * 197, 768), (256 * 133, 2304)):
    x = torch.randn(M, C, device=dev).bfloat16()
    cs = torch.zeros(C, device=dev)
    t1 = timeit(lambda: ops.transpose_bf16(x))
    t2 = timeit(lambda: ops.transpose_bf16(x, colsum=cs))
    gb = 2 * M * C * 2 / 1e9
    print(f"M={M} C={C}: transpose {t1*1e3:.1f} us ({gb/t1:.0f} GB/s)  +colsum {t2*1e3:.1f} us ({gb/t2:.0f} GB/s)")
