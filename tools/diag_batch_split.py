"""Diagnostic: which kernel output differs between a b=2048 forward and the b=256 forward of its first 256 samples?
Wraps every clibd_amd.ops entry point used by the towers' forward and logs a checksum of the rows that belong to the first
256 samples; prints the first call whose checksum differs between the two runs (image tower, then DNA tower)."""
import sys
import torch
sys.path.insert(0, ".")
from clibd_amd import ops
from clibd_amd.data import synthetic_batch
from clibd_amd.model import CLIBDDNAEncoder, CLIBDImageEncoder, SimpleCLIP, create_vit, load_pre_trained_bioscan_bert

dev = torch.device("cuda:0")
NB, CH = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 256
torch.manual_seed(2048)
model = SimpleCLIP(CLIBDImageEncoder(create_vit("vit_base_patch16_224"), r=4, num_classes=768),
                   CLIBDDNAEncoder(load_pre_trained_bioscan_bert(None), r=4, num_classes=768), None)
with torch.no_grad():
    for enc in (model.image_encoder, model.dna_encoder):
        for wb in enc.w_Bs:
            wb.weight.normal_(0, 0.02)
model = model.to(dev).eval()
model.overlap_towers = False
batch = synthetic_batch(NB, dev, seed=42, rank=0, with_text=False)
log = []
cur_B = [NB]


def rows_of_first(t):
    B = cur_B[0]
    if t.shape[0] % B:
        return None
    return t[: t.shape[0] // B * CH]


def wrap(name):
    inner = getattr(ops, name)

    def f(*a, **k):
        r = inner(*a, **k)
        outs = [v for v in list(a) + list(k.values()) if torch.is_tensor(v)]
        if torch.is_tensor(r):
            outs.append(r)
        elif isinstance(r, tuple):
            outs += [v for v in r if torch.is_tensor(v)]
        sig = []
        for v in outs:
            if v.dim() >= 1 and v.shape[0] >= cur_B[0] and v.dtype in (torch.bfloat16, torch.float32):
                s = rows_of_first(v)
                if s is not None:
                    sig.append((tuple(s.shape), float(s.double().abs().sum()), float(s.double().sum())))
        log.append((name, sig))
        return r

    setattr(ops, name, f)


for n in ("gemm_nt", "layernorm_fwd", "attention_fwd", "patchify", "vit_assemble_tokens", "gather_rows", "bert_embed", "softmax_mean_fwd",
          "l2norm_fwd", "lora_pack"):
    wrap(n)


def run(B):
    cur_B[0] = B
    log.clear()
    with torch.no_grad():
        i, d, _, _, _ = model(batch["image"][:B], batch["dna"][:B], None)
    torch.cuda.synchronize()
    return list(log), i[:CH].clone(), d[:CH].clone()


la, ia, da = run(NB)
lb, ib, db = run(CH)
lc, ic, dc = run(NB)
print("full vs full (determinism): image", float((ia - ic).abs().max()), "dna", float((da - dc).abs().max()))
print("full vs chunk0: image", float((ia - ib).abs().max()), "dna", float((da - db).abs().max()))
assert len(la) == len(lb), (len(la), len(lb))
bad = 0
for k, ((na, sa), (nb, sb)) in enumerate(zip(la, lb)):
    if na != nb or sa != sb:
        print(f"call {k}: {na}")
        for x, y in zip(sa, sb):
            print("    ", x, y, "DIFF" if x != y else "")
        bad += 1
        if bad >= 6:
            break
print("calls", len(la), "first-diff printed" if bad else "no checksum differs")
