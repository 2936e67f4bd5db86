"""Experiment (diagnostic build only: python -m clibd_amd.build --diag): gemm256 start-up skew / band-height knobs on the step's
GEMM shapes.  Parent mode spawns one child per (skew, band) setting (the knobs are read once per process).

    python tools/exp_gemm_knobs.py [M]          # parent
"""
import ctypes, os, subprocess, sys
import torch

sys.path.insert(0, ".")


def child(M):
    from clibd_amd import ops, _lib
    dev = torch.device("cuda:0"); BF16 = torch.bfloat16
    lib = _lib.load(); lib.clibd_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]
    stamps = torch.zeros(256 * 16 * 2 * 8, dtype=torch.int64, device=dev)
    lib.clibd_debug_set_gemm_stamps(stamps.data_ptr())
    res = []
    for name, N, K in (("fc1_gelu2", 3072, 768), ("qkv_bf16", 2304, 768), ("proj_res", 768, 768), ("fc2dgrad_aux", 3072, 768), ("fc2_res", 768, 3072), ("fc1dgrad", 768, 3072)):
        a = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        bias = torch.randn(N, device=dev)
        if name == "fc1_gelu2":
            kw = dict(bias=bias, act=ops.ACT_GELU_SAVE_GRAD, out_pre=torch.empty(M, N, device=dev, dtype=BF16), out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        elif name in ("proj_res", "fc2_res"):
            kw = dict(bias=bias, residual=torch.randn(M, N, device=dev), out_f32=torch.empty(M, N, device=dev))
        elif name == "fc2dgrad_aux":
            kw = dict(bias=bias, act=ops.ACT_MUL_AUX, aux=torch.randn(M, N, device=dev).to(BF16), out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        else:
            kw = dict(bias=bias, out_bf16=torch.empty(M, N, device=dev, dtype=BF16))
        for _ in range(2):
            ops.gemm_nt(a, w, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 6
        e0.record()
        for _ in range(reps):
            ops.gemm_nt(a, w, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        res.append(f"{name} {us:8.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF")
        del a, w, kw
    print(f"skew={os.environ.get('CLIBD_GEMM_SKEW', '0'):>5s} band={os.environ.get('CLIBD_GEMM_BAND', '4'):>2s} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        child(int(sys.argv[1]))
    else:
        M = int(sys.argv[1]) if len(sys.argv) > 1 else 403456
        for skew, band in ((0, 4), (0, 8), (0, 2), (0, 16), (300, 4), (600, 4), (900, 4), (1500, 4), (600, 8)):
            env = dict(os.environ, CLIBD_GEMM_SKEW=str(skew), CLIBD_GEMM_BAND=str(band))
            subprocess.run([sys.executable, __file__, str(M), "child"], env=env, check=False)
