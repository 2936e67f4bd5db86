"""Run one GEMM shape a few times (for rocprofv3 --pmc runs):  python3 tools/one_gemm.py N K [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
dev = torch.device("cuda:0")
M, N, K = 50432, int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm_nt(a, w, out_bf16=out)
torch.cuda.synchronize()
