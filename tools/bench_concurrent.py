"""Two GEMM chains (ViT-sized and DNA-sized fc1 -> fc2 pairs) on one stream vs two streams (full persistent grids that
queue behind each other: 1.112 -> 0.985 ms per pair; halving each grid so the kernels split the CUs was slower, 1.112 ms)."""
import sys, time
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16

def chain(M):
    H, F = 768, 3072
    x = torch.randn(M, H, device=dev).to(BF16)
    w1 = (torch.randn(F, H, device=dev) * 0.03).to(BF16); b1 = torch.randn(F, device=dev)
    w2 = (torch.randn(H, F, device=dev) * 0.03).to(BF16); b2 = torch.randn(H, device=dev)
    a = torch.empty(M, F, device=dev, dtype=BF16); g = torch.empty_like(a)
    res = torch.randn(M, H, device=dev); out = torch.empty(M, H, device=dev)
    def run():
        ops.gemm_nt(x, w1, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, out_pre=g, out_bf16=a)
        ops.gemm_nt(a, w2, bias=b2, residual=res, out_f32=out)
    return run

ra, rb = chain(50432), chain(34048)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one_stream(n):
    for _ in range(n):
        ra(); rb()
def two_streams(n):
    with torch.cuda.stream(s1):
        for _ in range(n): ra()
    with torch.cuda.stream(s2):
        for _ in range(n): rb()
for name, fn in (("one stream", one_stream), ("two streams", two_streams)):
    fn(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(10); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt / 10 * 1e3:.3f} ms per (ViT + DNA) fc1+fc2 pair", flush=True)
