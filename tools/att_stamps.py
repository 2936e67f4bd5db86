"""Diagnostic (needs python -m clibd_amd.build --diag): where a workgroup of attention_bwd spends its cycles.
stamps: 0 start | 1 K,V staged (wait + barrier) | 2 wave 0 done with phase 1 | 3 all waves done (barrier) | 4 Q,dO restaged | 5 wave 0 done with phase 2"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops, _lib
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
B, S, nh = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 197, 12
H = nh * 64
qkv = torch.randn(B * S, 3 * H, device=dev).to(BF16); do = torch.randn(B * S, H, device=dev).to(BF16); dqkv = torch.empty_like(qkv)
lib = _lib.load(); lib.clibd_debug_set_att_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.attention_bwd(qkv, do, B, S, nh, None, dqkv)
torch.cuda.synchronize()
buf = torch.zeros(B * nh * 8, dtype=torch.int64, device=dev)
lib.clibd_debug_set_att_stamps(buf.data_ptr())
ops.attention_bwd(qkv, do, B, S, nh, None, dqkv); torch.cuda.synchronize()
lib.clibd_debug_set_att_stamps(None)
st = buf.view(B * nh, 8).cpu().double()
d = st[:, 1:6] - st[:, 0:5]
names = ["stage K,V (DMA wait + barrier)", "phase 1 (wave 0)", "wait for the slowest wave", "restage Q,dO", "phase 2 (wave 0)"]
tot = (st[:, 5] - st[:, 0])
print(f"B={B} S={S}: workgroup lifetime median {tot.median():.0f} cycles (s_memtime = shader clock), mean {tot.mean():.0f}")
for k, n in enumerate(names):
    print(f"  {n:34s} median {d[:, k].median():8.0f}  mean {d[:, k].mean():8.0f}  ({100 * d[:, k].mean() / tot.mean():.1f} %)")
span = st[:, 5].max() - st[:, 0].min()
print(f"  kernel span {span:.0f} cycles; workgroups {B * nh}; sum of lifetimes / span = {tot.sum() / span:.1f} concurrent workgroups (of 512 slots at 2 per CU)")
