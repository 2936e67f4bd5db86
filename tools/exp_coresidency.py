"""Does an HBM-bound kernel run UNDER the persistent GEMM when the GEMM leaves registers free?  (build the GEMM with a VGPR cap first)
Streams: A = GEMM x n_g, B = LayerNorm backward x n_l; wall time together against each alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clibd_amd import ops
dev = torch.device("cuda:0")
M = 1024 * 197
g = torch.Generator(device=dev).manual_seed(1)
rb = lambda shape: (torch.randn(shape, device=dev, generator=g) * 0.5).bfloat16()
a, w = rb((M, 3072)), rb((768, 3072)) * 0.1
out = torch.empty((M, 768), dtype=torch.bfloat16, device=dev)
x = torch.randn((M, 768), device=dev); dy = rb((M, 768)); st = torch.zeros((M, 2), device=dev); st[:, 1] = 1.0
gam = torch.ones(768, device=dev); dres = torch.randn((M, 768), device=dev)
dx32 = torch.empty_like(x); dx16 = torch.empty_like(dy)
xa = torch.randn((M, 768), device=dev); y16 = torch.empty_like(dy); st2 = torch.empty((M, 2), device=dev)
la = rb((8, 768)); t8 = torch.empty((M, 8), dtype=torch.bfloat16, device=dev)
def gemm(): ops.gemm_nt(a, w, out_bf16=out)
def lnb(): ops.layernorm_bwd(dy, x, st, gam, dres=dres, dx_f32=dx32, dx_bf16=dx16)
def lnf(): ops.layernorm_fwd(xa, gam, gam, 1e-6, y_bf16=y16, stats=st2, lora_a=la, t_out=t8)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def wall(fa, na, fb, nb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sA):
        for _ in range(na): fa()
    with torch.cuda.stream(sB):
        for _ in range(nb): fb()
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for f in (gemm, lnb, lnf): f()
for name, fb in (("ln_bwd", lnb), ("ln_fwd+lora", lnf)):
    tg = min(wall(gemm, 20, fb, 0) for _ in range(3)); tl = min(wall(gemm, 0, fb, 40) for _ in range(3))
    nb = max(1, int(40 * tg / tl))
    tl = min(wall(gemm, 0, fb, nb) for _ in range(3))
    tb = min(wall(gemm, 20, fb, nb) for _ in range(3))
    print(f"{name}: gemm x20 alone {tg:.2f} ms, {name} x{nb} alone {tl:.2f} ms, together {tb:.2f} ms  (sum {tg + tl:.2f}, max {max(tg, tl):.2f})")
