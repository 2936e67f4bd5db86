"""Diagnostic (needs python -m clibd_amd.build --diag): where the waves of attention_bwd_sp_kernel spend their cycles.
Per-wave stamps: 0 start | 1 images staged + delta + barrier | per 64-query block it: 2+3it sweep done, 3+3it after the barrier,
4+3it dQ done | 14 loop done | 15 dK / dV stored."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops, _lib
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
B, S, nh = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 197, 12
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
H = nh * 64
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(BF16); do = torch.randn(B * S, H, device=dev).to(BF16); dqkv = torch.empty_like(qkv)
out, o_lo, lse = torch.empty(B * S, H, device=dev, dtype=BF16), torch.empty(B * S, H, device=dev, dtype=BF16), torch.empty(B * nh * S, device=dev)
drop = ops.Drop(p, 5) if p > 0 else None
ops.attention_fwd(qkv, B, S, nh, None, out, drop=drop, lse=lse, o_lo=o_lo)
lib = _lib.load(); lib.clibd_debug_set_att_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.attention_bwd_sp(qkv, do, out, o_lo, lse, B, S, nh, dqkv, drop=drop)
torch.cuda.synchronize()
buf = torch.zeros(B * nh * 8 * 16, dtype=torch.int64, device=dev)
lib.clibd_debug_set_att_stamps(buf.data_ptr())
ops.attention_bwd_sp(qkv, do, out, o_lo, lse, B, S, nh, dqkv, drop=drop); torch.cuda.synchronize()
lib.clibd_debug_set_att_stamps(None)
st = buf.view(B * nh, 8, 16).cpu().double()
life = st[:, :, 15] - st[:, :, 0]
print(f"B={B} S={S} p={p}: wave lifetime median {life.median():.0f} cycles; workgroup (max over waves) median {life.max(dim=1).values.median():.0f}")
names = {1: "stage 4 images + delta + barrier"}
nqb = ((S + 15) // 16 + 3) // 4
for it in range(min(nqb, 4)):
    names[2 + 3 * it] = f"block {it}: sweep"
    names[3 + 3 * it] = f"block {it}: wait at the barrier"
    names[4 + 3 * it] = f"block {it}: dQ"
last = 4 + 3 * (min(nqb, 4) - 1)
for w in (0, 4, 5, 7):
    print(f"  wave {w}:")
    prev = 0
    for k in sorted(names):
        d = st[:, w, k] - st[:, w, prev]
        print(f"    {names[k]:34s} median {d.median():8.0f}")
        prev = k
    print(f"    {'last barrier + epilogue':34s} median {(st[:, w, 15] - st[:, w, last]).median():8.0f}")
span = st[:, :, 15].max() - st[:, :, 0].min()
print(f"  kernel span {span:.0f} cycles for {B * nh} workgroups on 256 CUs: {span / (B * nh / 256):.0f} cycles per head and CU")
