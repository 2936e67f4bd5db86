"""Diagnostic: LayerNorm-forward (+LoRA down-projection) beside attention kernels on another stream: which output changes, where."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
dev = torch.device("cuda:0"); BF16, F32 = torch.bfloat16, torch.float32
torch.manual_seed(0)
B, S, H, NH = 64, 197, 768, 12
M = B * S
ns = torch.cuda.Stream()
mk = lambda *sh, dt=BF16, scale=1.0: (torch.randn(*sh, device=dev) * scale).to(dt)
n_qkv = mk(256 * 133, 3 * H); n_att = torch.empty(256 * 133, H, device=dev, dtype=BF16)
n_qkv2 = mk(64 * 197, 3 * H); n_att2 = torch.empty(64 * 197, H, device=dev, dtype=BF16)
x, gam, bet, acat = mk(M, H, dt=F32), mk(H, dt=F32), mk(H, dt=F32), mk(8, H)


def ln(lora=True):
    y, st = torch.empty(M, H, device=dev, dtype=BF16), torch.empty(M, 2, device=dev)
    t = torch.empty(M, 8, device=dev, dtype=BF16) if lora else None
    ops.layernorm_fwd(x, gam, bet, 1e-6, y_bf16=y, stats=st, lora_a=acat if lora else None, t_out=t)
    return (y, st, t) if lora else (y, st)


for lora in (True,):
    ref = [t.clone() for t in ln(lora)]
    torch.cuda.synchronize()
    tref = ref[0].float() @ acat.float().T
    print("t vs torch reference: max abs err", float((ref[2].float() - tref).abs().max()), "max |t|", float(tref.abs().max()), flush=True)
    for noise_name in ("attn133", "attn197", "attn_bwd133"):
        stats = [0, 0, 0]
        worst = None
        for it in range(10):
            ns.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ns):
                for _ in range(3):
                    if noise_name == "attn133":
                        ops.attention_fwd(n_qkv, 256, 133, NH, None, n_att)
                    elif noise_name == "attn197":
                        ops.attention_fwd(n_qkv2, 64, 197, NH, None, n_att2)
                    else:
                        ops.attention_bwd(n_qkv, n_att, 256, 133, NH, None, torch.empty_like(n_qkv))
            out = ln(lora)
            torch.cuda.current_stream().wait_stream(ns)
            torch.cuda.synchronize()
            for k, (a, b) in enumerate(zip(out, ref)):
                if not torch.equal(a, b):
                    stats[k] += 1
                    if worst is None:
                        d = (a.float() - b.float()).abs()
                        rows = (d.amax(dim=1) > 0).nonzero().flatten()
                        worst = (k, int((d > 0).sum()), float(d.max()), rows[:12].tolist(), int(rows.numel()), a.float().flatten()[d.flatten().argmax()].item(), b.float().flatten()[d.flatten().argmax()].item())
        print(f"lora={lora} noise={noise_name}: mismatching calls of 10 per output (y, stats, t) = {stats[:len(ref)]}; first: {worst}", flush=True)
