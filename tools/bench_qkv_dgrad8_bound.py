"""bound of an 8-bit QKV dgrad BEFORE building it.  The QKV dgrad is dX = dQKV . Wqkv (M x 768 x 2304 per layer) with the
adapters' rank-8 update in its epilogue; an 8-bit form would take dqkv as e4m3 with per-(row, head) scales.  Timing-only upper bound of what it
could save: the same shape through the EXISTING 8-bit dgrad kernel (per-row scales, no rank update: less work than the real thing would do)
against the product's bf16 launch (with the rank update), per layer and per step (12 layers per tower).
    python tools/bench_qkv_dgrad8_bound.py [batch, default 2048 and 1024]"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16; FP8 = torch.float8_e4m3fn
for B in [int(a) for a in sys.argv[1:]] or [2048, 1024]:
    tot16 = tot8 = 0.0
    for name, S in (("ViT", 197), ("DNA", 133)):
        M, N, K = B * S, 768, 2304
        dqkv = (torch.randn(M, K, device=dev) * 0.1).to(BF16)
        wt = (torch.randn(N, K, device=dev) * 0.02).to(BF16)
        dt = (torch.randn(M, 16, device=dev) * 0.1).to(BF16)
        vb = (torch.randn(N, 8, device=dev) * 0.1).to(BF16)
        out = torch.empty((M, N), dtype=BF16, device=dev)
        t16 = timeit(lambda: ops.gemm_nt(dqkv, wt, rank_u=dt, rank_v=vb, out_bf16=out))
        a8 = torch.randint(0, 120, (M, K), device=dev, dtype=torch.uint8).view(FP8)
        w8, cs = ops.quantize_rows_fp8_bf16(wt, 1.0)
        rd = torch.ones((M,), device=dev)
        t8 = timeit(lambda: ops.gemm_fp8_dgrad_nt(a8, w8, cs, a_row_dequant=rd, out_bf16=out))
        tot16 += 11 * t16; tot8 += 11 * t8     # the bottom layer of a LoRA tower has no QKV dgrad (nothing trainable below it)
        print(f"B={B} {name}: QKV dgrad M={M}: bf16 (+rank update) {t16*1e3:7.1f} us = {2*M*N*K/t16/1e9:6.0f} TF   e4m3 bound {t8*1e3:7.1f} us = {2*M*N*K/t8/1e9:6.0f} TF   x{t16/t8:.2f}", flush=True)
    print(f"B={B}: per step (11 layers per tower): bf16 {tot16:.2f} ms, e4m3 bound {tot8:.2f} ms -> at most {tot16 - tot8:.2f} ms per step before the cost of producing e4m3 dqkv + scales in the attention backward", flush=True)
