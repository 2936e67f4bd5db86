"""lora_wgrad timing at the bench shapes; M + 8 rows (not a multiple of 32) takes the VALU kernel, M the MFMA form."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from tools.bench_ops import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
for M0 in (50432, 34048, 403456, 272384):
    for M in (M0, M0 + 8):
        H = 768
        dqkv = torch.randn(M, 3 * H, device=dev).to(BF16); x = torch.randn(M, H, device=dev).to(BF16)
        t = torch.randn(M, 8, device=dev).to(BF16); dt = torch.randn(M, 16, device=dev).to(BF16)
        dA_q, dA_v = torch.zeros((4, H), device=dev), torch.zeros((4, H), device=dev)
        dB_q, dB_v = torch.zeros((H, 4), device=dev), torch.zeros((H, 4), device=dev)
        a = timeit(lambda: ops.lora_wgrad(dqkv, x, t, dt, dA_q, dA_v, dB_q, dB_v))
        print(f"M={M}: lora_wgrad {a*1e3:7.1f} us  ({(M*H*2*3)/a/1e9:5.2f} TB/s)  {'MFMA' if M % 32 == 0 else 'VALU'}", flush=True)
        del dqkv, x, t, dt
# the adapters' whole backward: dt GEMM + weight gradients (before) against clibd_lora_backward (dq, dv read once)
for M in (50432, 403456, 272384):
    H = 768
    dqkv = torch.randn(M, 3 * H, device=dev).to(BF16); x = torch.randn(M, H, device=dev).to(BF16)
    t = torch.randn(M, 8, device=dev).to(BF16); dt = torch.empty(M, 16, device=dev, dtype=BF16)
    w_dt = (torch.randn(16, 3 * H, device=dev) * 0.1).to(BF16); w_dt[8:] = 0; w_dt[:, H:2 * H] = 0
    dA_q, dA_v = torch.zeros((4, H), device=dev), torch.zeros((4, H), device=dev)
    dB_q, dB_v = torch.zeros((H, 4), device=dev), torch.zeros((H, 4), device=dev)
    def two_calls():
        ops.gemm_nt(dqkv, w_dt, out_bf16=dt, k_hole=(H, H))
        ops.lora_wgrad(dqkv, x, t, dt, dA_q, dA_v, dB_q, dB_v)
    a = timeit(two_calls)
    b = timeit(lambda: ops.lora_backward(dqkv, x, t, w_dt, dt, dA_q, dA_v, dB_q, dB_v))
    c = timeit(lambda: ops.lora_backward(dqkv, x, t, w_dt, dt, dA_q, dA_v, dB_q, dB_v, workspace=False))
    print(f"M={M}: dt GEMM + lora_wgrad {a*1e3:7.1f} us   lora_backward {b*1e3:7.1f} us (partials workspace)   {c*1e3:7.1f} us (float atomics)", flush=True)
    del dqkv, x, t, dt
