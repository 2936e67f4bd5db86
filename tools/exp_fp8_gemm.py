"""fp8 GEMM (clibd_gemm_fp8_nt): exact-integer check of every epilogue form + timing against the bf16 kernel on the step's shapes.
    python tools/exp_fp8_gemm.py [--batch 2048]"""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clibd_amd import ops

dev = torch.device("cuda:0")
FP8 = torch.float8_e4m3fn


def ints(shape, lo, hi, g):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def check():
    g = torch.Generator().manual_seed(0)
    worst = 0.0
    for (M, N, K) in [(2048, 768, 768), (2000, 2304, 768), (1111, 768, 3072), (300, 256, 512)]:
        a = ints((M, K), -3, 3, g); w = ints((N, K), -2, 2, g)
        cs = (2.0 ** torch.randint(-3, 2, (N,), generator=g).float())
        bias = ints((N,), -4, 4, g)
        res = ints((M, N), -8, 8, g)
        a8, w8 = a.to(FP8).to(dev), w.to(FP8).to(dev)
        ref = (a.double() @ w.double().T) * cs.double() + bias.double()
        # residual form (fp32 out, exact)
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        ops.gemm_fp8_nt(a8, w8, cs.to(dev), bias=bias.to(dev), residual=res.to(dev), out_f32=out)
        e = (out.cpu().double() - (ref + res.double())).abs().max().item(); worst = max(worst, e)
        print(f"M={M} N={N} K={K} residual form max err {e}")
        # bf16 out with rank update
        u = torch.zeros((M, 8)); u[:, :4] = ints((M, 4), -1, 1, g); u[:, 4:] = ints((M, 4), -1, 1, g)
        v = ints((N, 8), -1, 1, g)
        outb = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ops.gemm_fp8_nt(a8, w8, cs.to(dev), bias=bias.to(dev), rank_u=u.bfloat16().to(dev), rank_v=v.bfloat16().to(dev), out_bf16=outb)
        refb = (ref + u.double() @ v.double().T).float().bfloat16()
        e = (outb.cpu().float() - refb.float()).abs().max().item(); worst = max(worst, e)
        print(f"   bf16 + rank update max err {e}")
        # gelu form
        a2 = a * 0.25
        a8b = a2.to(FP8).to(dev)
        cs2 = cs * 0.125
        x = ((a2.double() @ w.double().T) * cs2.double() + bias.double() * 0.25).float()
        xb = x.bfloat16().float()
        gel = torch.nn.functional.gelu(xb)
        pre = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        go = torch.empty((M, N), dtype=torch.uint8, device=dev).view(FP8)
        ops.gemm_fp8_nt(a8b, w8, cs2.to(dev), bias=(bias * 0.25).to(dev), gelu_out_fp8=go, gelu_out_scale=4.0, out_pre=pre)
        got = go.cpu().float() / 4.0
        want = (gel * 4.0).clamp(-448, 448).to(FP8).float() / 4.0
        e = (got - want).abs().max().item()
        rel = ((got - gel).abs() / (gel.abs() + 0.02)).max().item()
        print(f"   gelu fp8 out: vs torch-quantised {e:.4g} (1 ulp steps allowed), rel to exact {rel:.3g}")
    return worst


def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def bench(batch):
    M = batch * 197
    print(f"timing at M={M}")
    g = torch.Generator(device=dev).manual_seed(1)
    def r8(shape): return torch.randint(0, 0x60, shape, dtype=torch.uint8, device=dev, generator=g).view(FP8)
    def rb(shape): return (torch.randn(shape, device=dev, generator=g) * 0.5).bfloat16()
    for name, N, K, form in [("qkv", 2304, 768, "bf16"), ("proj", 768, 768, "res"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "res")]:
        a8, w8 = r8((M, K)), r8((N, K)); ab, wb = rb((M, K)), rb((N, K))
        cs = torch.full((N,), 1e-3, device=dev); bias = torch.zeros((N,), device=dev)
        if form == "bf16":
            ob = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            u = rb((M, 8)); v = rb((N, 8))
            t8 = timeit(lambda: ops.gemm_fp8_nt(a8, w8, cs, bias=bias, rank_u=u, rank_v=v, out_bf16=ob))
            tb = timeit(lambda: ops.gemm_nt(ab, wb, bias=bias, rank_u=u, rank_v=v, out_bf16=ob))
        elif form == "res":
            x = torch.randn((M, N), device=dev); o = torch.empty_like(x)
            t8 = timeit(lambda: ops.gemm_fp8_nt(a8, w8, cs, bias=bias, residual=x, out_f32=o))
            tb = timeit(lambda: ops.gemm_nt(ab, wb, bias=bias, residual=x, out_f32=o))
        else:
            pre = torch.empty((M, N), dtype=torch.bfloat16, device=dev); ob = torch.empty_like(pre)
            o8 = torch.empty((M, N), dtype=torch.uint8, device=dev).view(FP8)
            t8 = timeit(lambda: ops.gemm_fp8_nt(a8, w8, cs, bias=bias, gelu_out_fp8=o8, gelu_out_scale=1.0, out_pre=pre))
            tb = timeit(lambda: ops.gemm_nt(ab, wb, bias=bias, act=ops.ACT_GELU_SAVE_GRAD, out_pre=pre, out_bf16=ob))
        fl = 2.0 * M * N * K
        print(f"{name:5s} N={N:5d} K={K:5d}  bf16 {tb:8.1f} us ({fl / tb / 1e6:7.1f} TF)   fp8 {t8:8.1f} us ({fl / t8 / 1e6:7.1f} TF)   x{tb / t8:.2f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=2048); a = ap.parse_args()
    w = check()
    print("worst abs err (exact forms):", w)
    bench(a.batch)
