"""Run-to-run determinism of the kernels changed this round, on one stream and beside a second stream's work.
python tools/stress_determinism.py [iters]"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()


def noise():
    """second-stream neighbour: a GEMM and an attention forward"""
    with torch.cuda.stream(side):
        ops.gemm_nt(na, nw, bias=nb, out_bf16=no)
        ops.attention_fwd(nqkv, 16, 133, 12, None, nout)


M2 = 2128
na = torch.randn(M2, 768, generator=g).to(dev, BF16); nw = (torch.randn(3072, 768, generator=g) * 0.05).to(dev, BF16)
nb = torch.randn(3072, generator=g).to(dev); no = torch.empty(M2, 3072, device=dev, dtype=BF16)
nqkv = torch.randn(16 * 133, 2304, generator=g).to(dev, BF16); nout = torch.empty(16 * 133, 768, device=dev, dtype=BF16)


def check(name, fn, outs, with_noise):
    fn(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    bad = 0
    for _ in range(iters):
        for o in outs:
            o.fill_(0)
        if with_noise:
            noise()
        fn()
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(ref, outs)):
            bad += 1
    print(f"{name:34s} neighbour={int(with_noise)}  mismatching runs {bad}/{iters}", flush=True)


for M in (3152, 403456 // 8):
    a = torch.randn(M, 768, generator=g).to(dev, BF16); w = (torch.randn(3072, 768, generator=g) * 0.05).to(dev, BF16)
    bias = torch.randn(3072, generator=g).to(dev)
    pre = torch.empty(M, 3072, device=dev, dtype=BF16); act = torch.empty(M, 3072, device=dev, dtype=BF16)
    aux = torch.randn(M, 3072, generator=g).to(dev, BF16); ob = torch.empty(M, 3072, device=dev, dtype=BF16)
    a2 = torch.randn(M, 3072, generator=g).to(dev, BF16); w2 = (torch.randn(768, 3072, generator=g) * 0.05).to(dev, BF16)
    b2 = torch.randn(768, generator=g).to(dev); res = torch.randn(M, 768, generator=g).to(dev); of = torch.empty(M, 768, device=dev)
    for wn in (False, True):
        check(f"gemm fc1 gelu2 M={M}", lambda: ops.gemm_nt(a, w, bias=bias, act=ops.ACT_GELU_SAVE_GRAD, out_pre=pre, out_bf16=act), [pre, act], wn)
        check(f"gemm fc2dgrad aux M={M}", lambda: ops.gemm_nt(a, w, act=ops.ACT_MUL_AUX, aux=aux, out_bf16=ob), [ob], wn)
        check(f"gemm fc2 res M={M}", lambda: ops.gemm_nt(a2, w2, bias=b2, residual=res, out_f32=of), [of], wn)
        check(f"gemm qkv bias M={M}", lambda: ops.gemm_nt(a, w, bias=bias, out_bf16=ob), [ob], wn)
for B in (16, 64):
    S, nh = 197, 12
    qkv = (torch.randn(B * S, 2304, generator=g) * 0.5).to(dev, BF16); out = torch.empty(B * S, 768, device=dev, dtype=BF16)
    do = torch.randn(B * S, 768, generator=g).to(dev, BF16); dqkv = torch.empty_like(qkv)
    for wn in (False, True):
        check(f"attention fwd B={B}", lambda: ops.attention_fwd(qkv, B, S, nh, None, out), [out], wn)
        check(f"attention bwd B={B}", lambda: ops.attention_bwd(qkv, do, B, S, nh, None, dqkv), [dqkv], wn)
