"""Micro-benchmarks of the individual HIP ops at BASELINE shapes (b=256): prints ms and TFLOP/s or GB/s."""
import sys
import time
import torch

sys.path.insert(0, ".")
from clibd_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_case(name, M, N, K, **kw):
    a = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    args = {}
    if kw.get("bias"):
        args["bias"] = torch.randn(N, device=dev)
    if kw.get("gelu"):
        args["act"] = ops.ACT_GELU
        args["out_pre"] = torch.empty(M, N, device=dev, dtype=BF16)
    if kw.get("gelu_grad"):
        args["act"] = ops.ACT_GELU_GRAD
        args["aux"] = torch.randn(M, N, device=dev).to(BF16)
    if kw.get("res"):
        args["residual"] = torch.randn(M, N, device=dev)
        args["out_f32"] = torch.empty(M, N, device=dev)
    else:
        args["out_bf16"] = torch.empty(M, N, device=dev, dtype=BF16)
    if kw.get("rank"):
        args["rank_u"] = torch.randn(M, 8, device=dev).to(BF16)
        args["rank_v"] = torch.randn(N, 8, device=dev).to(BF16)
    ms = timeit(lambda: ops.gemm_nt(a, w, **args))
    print(f"gemm {name:28s} M={M:6d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)


def main():
    Mv, Md = 256 * 197, 256 * 133
    gemm_case("plain 4096^3", 4096, 4096, 4096)
    gemm_case("plain 8192^3", 8192, 8192, 8192)
    gemm_case("vit qkv (bias+rank8)", Mv, 2304, 768, bias=True, rank=True)
    gemm_case("vit proj (bias+res f32)", Mv, 768, 768, bias=True, res=True)
    gemm_case("vit fc1 (bias+gelu, 2 outs)", Mv, 3072, 768, bias=True, gelu=True)
    gemm_case("vit fc2 (bias+res f32)", Mv, 768, 3072, bias=True, res=True)
    gemm_case("vit dgrad fc2 (gelu')", Mv, 3072, 768, gelu_grad=True)
    gemm_case("vit dgrad fc1", Mv, 768, 3072)
    gemm_case("vit dgrad qkv (rank8)", Mv, 768, 2304, rank=True)
    gemm_case("dna qkv", Md, 2304, 768, bias=True, rank=True)
    gemm_case("dna fc1", Md, 3072, 768, bias=True, gelu=True)
    # LayerNorm
    for M, H in ((Mv, 768), (Md, 768)):
        x = torch.randn(M, H, device=dev)
        g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
        yb = torch.empty(M, H, device=dev, dtype=BF16)
        st = torch.empty(M, 2, device=dev)
        acat = torch.randn(8, H, device=dev).to(BF16)
        t = torch.empty(M, 8, device=dev, dtype=BF16)
        ms = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=yb, stats=st, lora_a=acat, t_out=t))
        print(f"layernorm_fwd+lora M={M} H={H}: {ms:.3f} ms  {(M*H*6)/ms/1e6:.0f} GB/s")
        ms = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=yb, stats=st))
        print(f"layernorm_fwd      M={M} H={H}: {ms:.3f} ms  {(M*H*6)/ms/1e6:.0f} GB/s")
        dy = torch.randn(M, H, device=dev).to(BF16)
        dxf = torch.empty(M, H, device=dev)
        dxb = torch.empty(M, H, device=dev, dtype=BF16)
        ms = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dres=x, dx_f32=dxf, dx_bf16=dxb))
        print(f"layernorm_bwd      M={M} H={H}: {ms:.3f} ms  {(M*H*(2+4+4+4+2))/ms/1e6:.0f} GB/s")
    # attention
    for B, S, nh in ((256, 197, 12), (256, 133, 12), (256, 20, 8)):
        H = nh * 64
        qkv = torch.randn(B * S, 3 * H, device=dev).to(BF16)
        out = torch.empty(B * S, H, device=dev, dtype=BF16)
        ms = timeit(lambda: ops.attention_fwd(qkv, B, S, nh, None, out))
        fl = 4.0 * B * nh * S * S * 64
        print(f"attention_fwd B={B} S={S} h={nh}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s  {(B*S*H*8)/ms/1e6:.0f} GB/s")
        do = torch.randn(B * S, H, device=dev).to(BF16)
        dqkv = torch.empty_like(qkv)
        ms = timeit(lambda: ops.attention_bwd(qkv, do, B, S, nh, None, dqkv))
        print(f"attention_bwd B={B} S={S} h={nh}: {ms:.3f} ms  {2.5*fl/ms/1e9:.1f} TFLOP/s")
    # lora wgrad
    M, H = Mv, 768
    dqkv = torch.randn(M, 3 * H, device=dev).to(BF16)
    x = torch.randn(M, H, device=dev).to(BF16)
    t = torch.randn(M, 8, device=dev).to(BF16)
    dt = torch.randn(M, 16, device=dev).to(BF16)
    dA_q, dA_v = torch.zeros(4, H, device=dev), torch.zeros(4, H, device=dev)
    dB_q, dB_v = torch.zeros(H, 4, device=dev), torch.zeros(H, 4, device=dev)
    ms = timeit(lambda: ops.lora_wgrad(dqkv, x, t, dt, dA_q, dA_v, dB_q, dB_v))
    print(f"lora_wgrad M={M}: {ms:.3f} ms  {(M*H*6)/ms/1e6:.0f} GB/s")
    wdt = torch.randn(16, 3 * H, device=dev).to(BF16)
    ms = timeit(lambda: ops.gemm_nt(dqkv, wdt, out_bf16=dt))
    print(f"lora dt gemm M={M}: {ms:.3f} ms  {(M*H*6)/ms/1e6:.0f} GB/s")
    # softmax-mean
    B, S, C = 256, 133, 768
    lg = torch.randn(B * S, C, device=dev).to(BF16)
    ms = timeit(lambda: ops.softmax_mean_fwd(lg, B, S))
    print(f"softmax_mean_fwd: {ms:.3f} ms {(B*S*C*2)/ms/1e6:.0f} GB/s")
    img = torch.rand(256, 3, 224, 224, device=dev)
    ms = timeit(lambda: ops.patchify(img))
    print(f"patchify b=256: {ms:.3f} ms {(256*3*224*224*6)/ms/1e6:.0f} GB/s")
    # loss
    N, D = 2048, 768
    xx = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=-1)
    yy = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=-1)
    lab = torch.arange(N, device=dev)
    ws = ops.softce_workspace(N, N, D, dev)
    loss = torch.zeros(1, device=dev)
    sc = torch.tensor([14.28], device=dev)
    ms = timeit(lambda: ops.softce_rows_fwd(xx, yy, lab, 0, sc, loss, ws))
    print(f"softce_fwd N={N}: {ms:.3f} ms")
    dx, dy, ds = torch.zeros(N, D, device=dev), torch.zeros(N, D, device=dev), torch.zeros(1, device=dev)
    ms = timeit(lambda: ops.softce_rows_bwd(lab, N, N, D, 0, sc, 1.0 / N, dx, dy, ds, ws))
    print(f"softce_bwd N={N}: {ms:.3f} ms")


if __name__ == "__main__":
    main()
