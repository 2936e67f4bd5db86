"""fc1-forward GEMM (GELU_SAVE_GRAD epilogue) at the step's ViT shape: time per launch (for epilogue experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clibd_amd import ops
dev = torch.device("cuda:0")
M = 2048 * 197
g = torch.Generator(device=dev).manual_seed(1)
rb = lambda shape: (torch.randn(shape, device=dev, generator=g) * 0.5).bfloat16()
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
a, w = rb((M, 768)), rb((3072, 768)) * 0.1
bias = torch.zeros(3072, device=dev)
pre = torch.empty((M, 3072), dtype=torch.bfloat16, device=dev); out = torch.empty_like(pre)
t = timeit(lambda: ops.gemm_nt(a, w, bias=bias, act=ops.ACT_GELU_SAVE_GRAD, out_pre=pre, out_bf16=out))
print(f"fc1 gelu_save: {t:.1f} us  {2.0 * M * 3072 * 768 / t / 1e6:.1f} TF")
t = timeit(lambda: ops.gemm_nt(a, w, bias=bias, out_bf16=out))
print(f"fc1 plain bf16 out: {t:.1f} us")
aux = pre
a2, w2 = rb((M, 768)), rb((3072, 768)) * 0.1
t = timeit(lambda: ops.gemm_nt(a2, w2, act=ops.ACT_MUL_AUX, aux=aux, out_bf16=out))
print(f"fc2-dgrad mul_aux: {t:.1f} us")
