"""Diagnostic: does any kernel change its result when another stream's kernels run beside it?
For each kernel X: run it alone (reference), then repeatedly while a noise stream runs GEMM / LayerNorm / attention work on
other buffers; count bitwise mismatches."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops

dev = torch.device("cuda:0"); BF16, F32 = torch.bfloat16, torch.float32
torch.manual_seed(0)
B, S, H, FF, NH = 64, 197, 768, 3072, 12
M = B * S
noise_stream = torch.cuda.Stream()


def mk(*shape, dt=BF16, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dt)


# ---- noise work (own buffers)
nM = 256 * 133
n_a, n_w1, n_w2 = mk(nM, H), mk(FF, H, scale=0.05), mk(H, FF, scale=0.05)
n_h, n_g, n_o = torch.empty(nM, FF, device=dev, dtype=BF16), torch.empty(nM, FF, device=dev, dtype=BF16), torch.empty(nM, H, device=dev, dtype=F32)
n_x, n_res = mk(nM, H, dt=F32), mk(nM, H, dt=F32)
n_gam, n_bet, n_bias1, n_bias2 = torch.ones(H, device=dev), torch.zeros(H, device=dev), torch.zeros(FF, device=dev), torch.zeros(H, device=dev)
n_y, n_st = torch.empty(nM, H, device=dev, dtype=BF16), torch.empty(nM, 2, device=dev)
n_qkv, n_att = mk(nM, 3 * H), torch.empty(nM, H, device=dev, dtype=BF16)


def noise(kind):
    if kind in ("gemm", "mix"):
        ops.gemm_nt(n_a, n_w1, bias=n_bias1, act=ops.ACT_GELU_SAVE_GRAD, out_pre=n_g, out_bf16=n_h)
        ops.gemm_nt(n_h, n_w2, bias=n_bias2, residual=n_res, out_f32=n_o)
    if kind in ("ln", "mix"):
        ops.layernorm_fwd(n_x, n_gam, n_bet, 1e-6, y_bf16=n_y, stats=n_st)
        ops.layernorm_bwd(n_y, n_x, n_st, n_gam, dres=n_res, dx_f32=n_o, dx_bf16=n_y)
    if kind in ("attn", "mix"):
        ops.attention_fwd(n_qkv, 256, 133, NH, None, n_att)


# ---- kernels under test
x_a, x_w1, x_w2, x_wq = mk(M, H), mk(FF, H, scale=0.05), mk(H, FF, scale=0.05), mk(3 * H, H, scale=0.05)
x_bias1, x_bias2, x_biasq = mk(FF, dt=F32), mk(H, dt=F32), mk(3 * H, dt=F32)
x_f32, x_res = mk(M, H, dt=F32), mk(M, H, dt=F32)
x_gam, x_bet = mk(H, dt=F32), mk(H, dt=F32)
x_acat = mk(8, H)
x_t, x_vf = mk(M, 8), mk(3 * H, 8, scale=0.05)
x_qkv = mk(M, 3 * H)
x_do = mk(M, H)
x_aux = mk(M, FF)
x_big = mk(M, FF)


def k_fc1():
    a, g = torch.empty(M, FF, device=dev, dtype=BF16), torch.empty(M, FF, device=dev, dtype=BF16)
    ops.gemm_nt(x_a, x_w1, bias=x_bias1, act=ops.ACT_GELU_SAVE_GRAD, out_pre=g, out_bf16=a)
    return a, g


def k_qkv_lora():
    o = torch.empty(M, 3 * H, device=dev, dtype=BF16)
    ops.gemm_nt(x_a, x_wq, bias=x_biasq, rank_u=x_t, rank_v=x_vf, out_bf16=o)
    return (o,)


def k_fc2_res():
    o = torch.empty(M, H, device=dev, dtype=F32)
    ops.gemm_nt(x_big, x_w2, bias=x_bias2, residual=x_res, out_f32=o)
    return (o,)


def k_mul_aux():
    o = torch.empty(M, FF, device=dev, dtype=BF16)
    ops.gemm_nt(x_a, x_w1, act=ops.ACT_MUL_AUX, aux=x_aux, out_bf16=o)
    return (o,)


def k_gemm128():
    o = torch.empty(512, H, device=dev, dtype=F32)
    ops.gemm_nt(x_a[:512], x_wq[:H], bias=x_bias2, residual=x_res[:512], out_f32=o)
    return (o,)


def k_ln_fwd():
    y, st, t = torch.empty(M, H, device=dev, dtype=BF16), torch.empty(M, 2, device=dev), torch.empty(M, 8, device=dev, dtype=BF16)
    ops.layernorm_fwd(x_f32, x_gam, x_bet, 1e-6, y_bf16=y, stats=st, lora_a=x_acat, t_out=t)
    return y, st, t


def k_ln_bwd():
    st = torch.empty(M, 2, device=dev); y = torch.empty(M, H, device=dev, dtype=BF16)
    ops.layernorm_fwd(x_f32, x_gam, x_bet, 1e-6, y_bf16=y, stats=st)
    dx, dxb = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev, dtype=BF16)
    ops.layernorm_bwd(x_do, x_f32, st, x_gam, dres=x_res, dx_f32=dx, dx_bf16=dxb)
    return dx, dxb


def k_attn_fwd():
    o = torch.empty(M, H, device=dev, dtype=BF16)
    ops.attention_fwd(x_qkv, B, S, NH, None, o)
    return (o,)


def k_attn_bwd():
    d = torch.empty(M, 3 * H, device=dev, dtype=BF16)
    ops.attention_bwd(x_qkv, x_do, B, S, NH, None, d)
    return (d,)


tests = [("gemm256 fc1 gelu2", k_fc1), ("gemm256 qkv+lora", k_qkv_lora), ("gemm256 fc2 res_f32", k_fc2_res), ("gemm256 mul_aux", k_mul_aux),
         ("gemm128 res", k_gemm128), ("layernorm fwd+lora", k_ln_fwd), ("layernorm bwd", k_ln_bwd), ("attention fwd", k_attn_fwd),
         ("attention bwd", k_attn_bwd)]
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for name, fn in tests:
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    solo = sum(any(not torch.equal(a, b) for a, b in zip(fn(), ref)) for _ in range(5))
    res = {}
    for kind in ("gemm", "ln", "attn", "mix"):
        bad = 0
        for _ in range(REPS):
            noise_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(noise_stream):
                for _ in range(3):
                    noise(kind)
            out = fn()
            torch.cuda.current_stream().wait_stream(noise_stream)
            torch.cuda.synchronize()
            if any(not torch.equal(a, b) for a, b in zip(out, ref)):
                bad += 1
        res[kind] = bad
    print(f"{name:24s} solo mismatches {solo}/5   beside noise (of {REPS}): {res}", flush=True)
