"""The step's GEMM shapes through the product library, one line per shape (µs per launch, TFLOP/s).

    python tools/bench_gemm_shapes.py [M] [reps] [seconds]   # default M = 403456 (ViT token rows at b=2048)

With [seconds] > 0 every shape runs for about that long and prints its wall-clock window (for tools/power_sampler.py).
"""
import sys
import time
import torch

sys.path.insert(0, ".")
from clibd_amd import ops  # noqa: E402

SHAPES = (("fc1_gelu2", 3072, 768), ("qkv_bf16", 2304, 768), ("proj_res", 768, 768), ("fc2dgrad_aux", 3072, 768), ("fc2_res", 768, 3072),
          ("fc1dgrad", 768, 3072), ("qkvdgrad", 768, 2304), ("projdgrad_add", 768, 768))


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 403456
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dev = torch.device("cuda:0"); BF16 = torch.bfloat16
    tot = 0.0
    for name, N, K in SHAPES:
        a = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        bias = torch.randn(N, device=dev)
        if name == "fc1_gelu2":
            kw = dict(bias=bias, act=ops.ACT_GELU_SAVE_GRAD, out_pre=torch.empty(M, N, device=dev, dtype=BF16), out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        elif name in ("proj_res", "fc2_res"):
            kw = dict(bias=bias, residual=torch.randn(M, N, device=dev), out_f32=torch.empty(M, N, device=dev))
        elif name == "fc2dgrad_aux":
            kw = dict(act=ops.ACT_MUL_AUX, aux=torch.randn(M, N, device=dev).to(BF16), out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        elif name == "projdgrad_add":
            kw = dict(act=ops.ACT_ADD_AUX, aux=torch.randn(M, N, device=dev).to(BF16), out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        elif name == "qkv_bf16":
            kw = dict(bias=bias, out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        else:
            kw = dict(out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        for _ in range(3):
            ops.gemm_nt(a, w, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm_nt(a, w, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        tot += us
        line = f"{name:14s} N={N:5d} K={K:5d} {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF"
        if secs > 0:
            n = max(1, int(secs * 1e6 / us))
            t0 = time.time()
            e0.record()
            for _ in range(n):
                ops.gemm_nt(a, w, **kw)
            e1.record(); torch.cuda.synchronize()
            t1 = time.time()
            us2 = e0.elapsed_time(e1) / n * 1e3
            line += f" | sustained {us2:8.1f} us {2.0 * M * N * K / us2 / 1e6:6.0f} TF window {t0:.3f} {t1:.3f}"
        print(line, flush=True)
        del a, w, kw
    print(f"sum {tot:8.1f} us")


if __name__ == "__main__":
    main()
