"""Diagnostic: per-tile s_memtime stamps of gemm256 (where a persistent workgroup spends its time).
Needs the diagnostic build (the product library carries no stamp buffer): python -m clibd_amd.build --diag"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops, _lib
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
M, N, K = 50432, int(sys.argv[1]) if len(sys.argv) > 1 else 2304, int(sys.argv[2]) if len(sys.argv) > 2 else 768
a = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
out = torch.empty(M, N, device=dev, dtype=BF16)
kw = dict(out_bf16=out)
if len(sys.argv) > 3 and sys.argv[3] == "gelu":
    kw = dict(bias=torch.randn(N, device=dev), act=ops.ACT_GELU_SAVE_GRAD, out_pre=torch.empty_like(out), out_bf16=out)
if len(sys.argv) > 3 and sys.argv[3] == "res":
    kw = dict(bias=torch.randn(N, device=dev), residual=torch.randn(M, N, device=dev), out_f32=torch.empty(M, N, device=dev))
for _ in range(3): ops.gemm_nt(a, w, **kw)
buf = torch.zeros((256, 16, 2, 8), dtype=torch.int64, device=dev)
lib = _lib.load(); lib.clibd_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]; lib.clibd_debug_set_gemm_stamps(buf.data_ptr())
ops.gemm_nt(a, w, **kw); torch.cuda.synchronize(); lib.clibd_debug_set_gemm_stamps(None)
s = buf.cpu().double()
names = ["head0-3", "head4-7", "steady", "last iter", "epilogue"]
for g in (0, 1):
    print(f"wave group {g}: cycles per segment, median over workgroups, tiles 0..5")
    for ti in range(3):
        x = s[:, ti, g, :]
        ok = x[:, 0] > 0
        d = [(x[ok, k + 1] - x[ok, k]).median().item() for k in range(5)]
        tot = (x[ok, 5] - x[ok, 0]).median().item()
        gap = (s[ok, ti + 1, g, 0] - x[ok, 5]).median().item() if ti < 2 else float("nan")
        print(f"  tile {ti}: " + "  ".join(f"{n}={v:7.0f}" for n, v in zip(names, d)) + f"  total={tot:7.0f}  gap_to_next_start={gap:7.0f}")
