"""Round 5: the LayerNorm -> Linear fold at the bench's ViT shape — projection (plain fp32-residual kind against the producer kind),
LayerNorm forward, finalize, fc1 (GELU-save kind against the consumer kind).   python tools/bench_ln_fold.py [M]"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 403456
H, FF = 768, 3072
g = torch.Generator().manual_seed(1)
a = torch.randn(M, H, device=dev).to(BF16); wo = (torch.randn(H, H, device=dev) * 0.03).to(BF16); bo = torch.randn(H, device=dev)
res = torch.randn(M, H, device=dev); x1 = torch.empty((M, H), device=dev); x1b = torch.empty((M, H), dtype=BF16, device=dev)
sums = torch.empty((H // 128, M, 2), device=dev); stats = torch.empty((M, 2), device=dev)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
w1 = torch.randn(FF, H, device=dev) * 0.03; b1 = torch.randn(FF, device=dev)
wg, s_n, bp = ops.ln_fold_weights(w1, gamma, beta, b1)
w1b = w1.to(BF16)
act, dg = torch.empty((M, FF), dtype=BF16, device=dev), torch.empty((M, FF), dtype=BF16, device=dev)
xn = torch.empty((M, H), dtype=BF16, device=dev)
t = {}
for rep in range(2):
    t["proj plain"] = timeit(lambda: ops.gemm_nt(a, wo, bias=bo, residual=res, out_f32=x1))
    t["proj + bf16 copy + row sums"] = timeit(lambda: ops.gemm_nt(a, wo, bias=bo, residual=res, out_f32=x1, out_bf16=x1b, row_sums=sums))
    t["layernorm_fwd"] = timeit(lambda: ops.layernorm_fwd(x1, gamma, beta, 1e-6, y_bf16=xn, stats=stats))
    t["rowsum_finalize"] = timeit(lambda: ops.rowsum_finalize(sums, 1e-6, stats))
    t["fc1 gelu-save"] = timeit(lambda: ops.gemm_nt(xn, w1b, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, out_pre=dg, out_bf16=act))
    t["fc1 row-norm + gelu-save"] = timeit(lambda: ops.gemm_nt(x1b, wg, bias=bp, act=ops.ACT_GELU_SAVE_GRAD, out_pre=dg, out_bf16=act, row_stats=stats, col_sum_w=s_n))
    print(f"M={M} rep {rep}: " + "  ".join(f"{k} {v * 1e3:7.1f} us" for k, v in t.items()), flush=True)
    std = t["proj plain"] + t["layernorm_fwd"] + t["fc1 gelu-save"]
    fold = t["proj + bf16 copy + row sums"] + t["rowsum_finalize"] + t["fc1 row-norm + gelu-save"]
    print(f"   unfolded {std * 1e3:7.1f} us   folded {fold * 1e3:7.1f} us   saved {(std - fold) * 1e3:6.1f} us per ViT block", flush=True)
